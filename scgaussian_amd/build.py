"""Build libscg_raster.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m scgaussian_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  geometry.hip is compiled with -ffp-contract=off (bit-exact tile assignment).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libscg_raster.so")
EXPORTS_MAP = os.path.join(CSRC, "exports.map")           # global: scg_*; local: everything else

ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-munsafe-fp-atomics", "-Wall",
          "-Wno-unused-function", "-I", INCLUDE]
SOURCES = {
    "api.hip": [],
    "geometry.hip": ["-ffp-contract=off"],
    "binning.hip": [],
    "binning_tiles.hip": [],
    # measured (tools/probes/valu_rate.hip): v_pk_*_f32 costs 1.67 issue slots, so the SLP vectoriser's packing of the
    # per-pixel scalar math (plus the v_mov shuffles it needs) loses; the explicit f32x2 accumulators still pack
    "blend.hip": ["-fno-slp-vectorize"],
    "knn.hip": [],
    "loss.hip": [],
    "matchloss.hip": [],
}
HEADERS = [os.path.join(CSRC, "scg_common.h"), os.path.join(CSRC, "tile_sort.h"), os.path.join(CSRC, "tile_walk.h"), os.path.join(CSRC, "scg_debug.h"), os.path.join(INCLUDE, "scg_raster.h"), os.path.join(INCLUDE, "scg_knn.h"), os.path.join(INCLUDE, "scg_loss.h"), os.path.join(INCLUDE, "scg_matchloss.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _digest(paths, extra: str) -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


# The one variant that ships next to the product: the blend kernels with the COMPILER-written forward trip and backward walk
# instead of the hand-written ISA (csrc/blend.hip SCG_FWD_TRIP_CXX).  tests/test_gpu_parity.py renders with both and demands
# bit-identical outputs, and differentiates through both (gradients equal up to the order of the float atomics) — the guard of
# the hand-written code's hard-coded registers against a toolchain change.
CXX_TRIP_TAG = "cxx"


def build_cxx_trip_variant(verbose: bool = False) -> str:
    build(verbose=verbose)
    return build(verbose=verbose, tag=CXX_TRIP_TAG, defines=("SCG_FWD_TRIP_CXX",), only=("blend.hip",))


def build(force: bool = False, verbose: bool = False, tag: str = "", defines=(), flags=(), only=None) -> str:
    """tag/defines build a VARIANT libscg_raster_<tag>.so (selected at run time with SCG_LIB_PATH); the default build has
    neither.  `only`: the sources the defines / flags apply to — the others are linked from the default build's objects."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    rebuilt = False
    lib_path = LIB_PATH if not tag else LIB_PATH.replace(".so", f"_{tag}.so")
    for src, extra in SOURCES.items():
        variant = bool(tag) and (only is None or src in only)
        if variant:
            extra = list(extra) + ["-D" + d for d in defines] + list(flags)
        src_path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ_DIR, src.replace(".hip", (f"_{tag}" if variant else "") + ".o"))
        stamp = obj + ".sha"
        dig = _digest([src_path] + HEADERS, " ".join(COMMON + extra))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        cmd = [hipcc] + COMMON + extra + ["-c", src_path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stamp, "w") as fh:
            fh.write(dig)
        rebuilt = True
    if rebuilt or force or not os.path.exists(lib_path):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + EXPORTS_MAP, "-o", lib_path] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib_path


if __name__ == "__main__":
    tag, defines, flags = "", [], []
    for a in sys.argv[1:]:
        if a.startswith("--tag="):
            tag = a.split("=", 1)[1]
        elif a.startswith("-D"):
            defines.append(a[2:])
        elif a.startswith(("-f", "-m")):          # experiment variants: extra compiler flags for every source
            flags.append(a)
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv, tag=tag,
                 defines=defines, flags=flags)
    print(path)
