"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol that
include/scg_raster.h declares, validates arguments without touching a GPU, and the Python operator
mirrors the reference's error behaviour.  No compute calls."""
import ctypes as C
import os
import re

import pytest
import torch

from scgaussian_amd import _lib
from scgaussian_amd._lib import ScgFrame
from scgaussian_amd import rasterizer as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = set()
    for hdr in ("scg_raster.h", "scg_knn.h", "scg_loss.h", "scg_matchloss.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(scg_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"libscg_raster.so does not export {name}"
        assert name in _lib.SYMBOLS, f"ctypes binding lacks {name}"
    assert lib.scg_abi_version() == _lib.ABI_VERSION == 10
    # ... and NOTHING ELSE: the sources are compiled with -fvisibility=hidden (the declarations carry SCG_API) and linked with
    # csrc/exports.map, so no internal scg:: function, kernel handle or __hip_cuid_* symbol leaks into the dynamic table
    import subprocess
    for path in (_lib.LIB_PATH,):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
        assert exported == declared, sorted(set(exported) ^ set(declared))
    # the structs the binding declares have the size the library was compiled with (ScgFrame grew in ABI 5)
    import ctypes as C
    for which, struct in enumerate((_lib.ScgFrame, _lib.ScgWorkspaceLayout, _lib.ScgStageEvents)):
        assert lib.scg_struct_bytes(which) == C.sizeof(struct) > 0
    assert lib.scg_struct_bytes(99) == 0


def test_scratch_size_queries_are_monotone():
    lib = _lib.load()
    assert lib.scg_geometry_scratch_bytes(1) > 0
    assert lib.scg_geometry_scratch_bytes(1_000_000) >= 1_000_000 // 256 * 4
    a = lib.scg_binning_scratch_bytes(500, 1000, 256, 256, 0)
    b = lib.scg_binning_scratch_bytes(500_000, 1_000_000, 1920, 1080, 0)
    c = lib.scg_binning_scratch_bytes(500_000, 1_000_000, 1920, 1080, 1)      # global 64-bit sort: 20 B / instance
    assert 0 < a < b and c >= 1_000_000 * 20
    # the tile-first path never materialises the R 64-bit key/value pairs twice: smaller scratch than the global sort
    assert lib.scg_binning_scratch_bytes(500_000, 4_000_000, 1920, 1080, 0) < lib.scg_binning_scratch_bytes(500_000, 4_000_000, 1920, 1080, 1)
    assert lib.scg_sort_scratch_bytes(10) > 0 and lib.scg_scan_scratch_bytes(10) > 0


def _frame(**kw):
    base = dict(P=4, sh_degree=3, sh_coeffs=16, width=64, height=48, tanfovx=0.5, tanfovy=0.4,
                scale_modifier=1.0, prefiltered=0, debug=0, viewmatrix=0x1000, projmatrix=0x1000, campos=0x1000, bg=0x1000)
    base.update(kw)
    return ScgFrame(**base)


def test_argument_validation_returns_codes_without_a_gpu():
    lib = _lib.load()
    fake = 0x1000  # never dereferenced: validation fails first
    # NULL frame
    rc = lib.scg_geometry_forward(None, *([fake] * 7), *([fake] * 6), fake, 1 << 20, None)
    assert rc == -1 and b"frame" in lib.scg_last_error()
    # degree out of range
    fr = _frame(sh_degree=5)
    rc = lib.scg_geometry_forward(C.byref(fr), *([fake] * 7), *([fake] * 6), fake, 1 << 20, None)
    assert rc == -2
    # both shs and colors_precomp
    fr = _frame()
    rc = lib.scg_geometry_forward(C.byref(fr), fake, fake, fake, fake, fake, fake, None, *([fake] * 6), fake, 1 << 20, None)
    assert rc == -3 and b"exactly one" in lib.scg_last_error()
    # neither scale/rot nor cov3D
    rc = lib.scg_geometry_forward(C.byref(fr), fake, fake, fake, None, None, None, None, *([fake] * 6), fake, 1 << 20, None)
    assert rc == -3
    # scratch too small
    rc = lib.scg_geometry_forward(C.byref(fr), fake, fake, fake, None, fake, fake, None, *([fake] * 6), fake, 0, None)
    assert rc == -4
    # misaligned splats
    rc = lib.scg_geometry_forward(C.byref(fr), fake, fake, fake, None, fake, fake, None, fake + 4, fake, fake, fake, fake, fake, fake, 1 << 20, None)
    assert rc == -5
    # binning: unknown algorithm selector
    assert lib.scg_binning(C.byref(fr), 10, fake, fake, fake, fake, None, 7, fake, 1 << 30, None) == -2
    # sort: bad end_bit
    assert lib.scg_sort_pairs(fake, fake, fake, fake, 10, 0, fake, 1 << 20, None) == -2
    # geometry backward: gradient outputs must match the input path
    rc = lib.scg_geometry_backward(C.byref(fr), fake, fake, fake, None, fake, fake, None, fake, fake, fake,
                                   fake, fake, fake, None, None, fake, fake, None, 0, None)
    assert rc == -3


def test_operator_error_behaviour_matches_reference_api():
    st = R.GaussianRasterizationSettings(48, 64, 0.5, 0.4, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3,
                                         torch.zeros(3), False, False)
    assert st._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                          "projmatrix", "sh_degree", "campos", "prefiltered", "debug")
    rast = R.GaussianRasterizer(raster_settings=st)
    P = 5
    m, m2, op = torch.zeros(P, 3), torch.zeros(P, 3), torch.ones(P, 1)
    sh, col = torch.zeros(P, 16, 3), torch.zeros(P, 3)
    sc, rot, cov = torch.ones(P, 3), torch.zeros(P, 4), torch.zeros(P, 6)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=m, means2D=m2, opacities=op, shs=sh, colors_precomp=col, scales=sc, rotations=rot)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=m, means2D=m2, opacities=op, scales=sc, rotations=rot)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=m, means2D=m2, opacities=op, shs=sh, scales=sc, rotations=rot, cov3D_precomp=cov)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=m, means2D=m2, opacities=op, shs=sh, scales=sc)
    # no silent CPU fallback: CPU tensors are refused loudly
    with pytest.raises(_lib.ScgError, match="no CPU path"):
        rast(means3D=m, means2D=m2, opacities=op, shs=sh, scales=sc, rotations=rot)


def test_drop_in_module_name():
    import diff_gaussian_rasterization as dgr
    assert dgr.GaussianRasterizer is R.GaussianRasterizer
    assert dgr.GaussianRasterizationSettings is R.GaussianRasterizationSettings


def test_one_call_entry_points_layout_and_validation_without_a_gpu():
    """scg_workspace_layout: aligned, non-overlapping, monotone in the capacity; scg_forward / scg_backward /
    scg_wait_num_rendered reject bad arguments with codes, before anything touches a device."""
    lib = _lib.load()
    L = _lib.ScgWorkspaceLayout()
    assert lib.scg_workspace_layout(200_000, 1_500_000, 1008, 756, C.byref(L)) == 0
    names = ("splats", "rects", "depth_keys", "clamped", "point_list", "ranges", "final_T", "n_contrib", "bin_scratch")
    offs = [int(getattr(L, n)) for n in names] + [int(L.total)]
    assert all(o % 256 == 0 for o in offs) and offs == sorted(offs) and offs[0] == 0
    assert offs[1] - offs[0] >= 200_000 * 48 and offs[5] - offs[4] >= 1_500_000 * 4
    assert offs[7] - offs[6] >= 1008 * 756 * 4 and int(L.partial_words) * 4 == lib.scg_geometry_scratch_bytes(200_000)
    L2 = _lib.ScgWorkspaceLayout()
    assert lib.scg_workspace_layout(200_000, 3_000_000, 1008, 756, C.byref(L2)) == 0 and L2.total > L.total
    assert lib.scg_workspace_layout(200_000, -1, 1008, 756, C.byref(L2)) == -2                  # SCG_E_RANGE
    assert lib.scg_workspace_layout(200_000, 10, 1008, 756, None) == -1                         # SCG_E_NULL
    fr = _frame()
    fake = 0x1000
    args = [C.byref(fr), fake, fake, fake, None, fake, fake, None]
    # NULL workspace
    assert lib.scg_forward(*args, 100, None, 0, fake, fake, fake, fake, fake, None, None, 0, None, None) == -1
    # workspace too small
    rc = lib.scg_forward(*args, 100, fake, 16, fake, fake, fake, fake, fake, None, None, 0, None, None)
    assert rc == -4 and b"workspace" in lib.scg_last_error()
    # both colour inputs: SCG_E_EXCLUSIVE, with the reference's wording
    bad = [C.byref(fr), fake, fake, fake, fake, fake, fake, None]
    assert lib.scg_forward(*bad, 100, fake, 1 << 30, fake, fake, fake, fake, fake, None, None, 0, None, None) == -3
    assert b"exactly one of either SHs or precomputed colors" in lib.scg_last_error()
    assert lib.scg_backward(None, *([fake] * 7), fake, 100, fake, fake, None, None, fake, 0, *([fake] * 8), 0, None, None) == -1
    assert lib.scg_wait_num_rendered(None, None, 10) < 0
    # the host-side sum of the per-workgroup partial sums (no event: nothing to wait for)
    part = (C.c_uint32 * 8)(5, 7, 11, 0, 0, 0, 0, 0)
    assert lib.scg_wait_num_rendered(None, C.cast(part, C.c_void_p), 600) == 23
    assert lib.scg_event_elapsed_ms(None, None, None) == -1
    # ABI 9: armed words (scg_forward's SCG_FORWARD_ARM_PARTIAL_SUMS) are WATCHED until their workgroup has written them — here a
    # thread plays the geometry kernel, writing the three words of P = 600 one by one
    import threading
    import time
    armed = (C.c_uint32 * 3)(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)

    def kernel():
        for i, v in ((1, 40), (0, 2), (2, 100)):
            time.sleep(0.02)
            armed[i] = v
    th = threading.Thread(target=kernel)
    t0 = time.perf_counter()
    th.start()
    assert lib.scg_wait_num_rendered(None, C.cast(armed, C.c_void_p), 600) == 142
    assert time.perf_counter() - t0 >= 0.05
    th.join()


def test_debug_flag_dumps_the_arguments_of_a_failing_call(tmp_path, monkeypatch):
    """arguments/__init__.py:68 `--debug` -> gaussian_renderer/__init__.py:50: with debug=True a failing rasterizer call leaves
    a snapshot of its arguments behind (upstream: snapshot_fw.dump) and re-raises; without it nothing is written."""
    monkeypatch.chdir(tmp_path)
    P = 5
    base = dict(image_height=48, image_width=64, tanfovx=0.5, tanfovy=0.4, bg=torch.zeros(3), scale_modifier=1.0,
                viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=3, campos=torch.zeros(3), prefiltered=False)
    kw = dict(means3D=torch.rand(P, 3), means2D=torch.zeros(P, 3), opacities=torch.rand(P, 1), shs=torch.rand(P, 16, 3),
              scales=torch.rand(P, 3), rotations=torch.rand(P, 4))
    with pytest.raises(_lib.ScgError):                       # CPU tensors: the product has no CPU path
        R.GaussianRasterizer(R.GaussianRasterizationSettings(debug=False, **base))(**kw)
    assert not os.path.exists(R.DEBUG_SNAPSHOT_FW)
    with pytest.raises(_lib.ScgError):
        R.GaussianRasterizer(R.GaussianRasterizationSettings(debug=True, **base))(**kw)
    snap = torch.load(R.DEBUG_SNAPSHOT_FW, weights_only=False)
    assert torch.equal(snap["means3D"], kw["means3D"]) and torch.equal(snap["sh"], kw["shs"]) and snap["cov3Ds_precomp"] is None
    assert snap["raster_settings"][0] == 48 and snap["raster_settings"][-1] is True
