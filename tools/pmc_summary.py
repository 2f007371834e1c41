#!/usr/bin/env python3
"""Turn the counter CSVs of tools/pmc.sh (one rocprofv3 --pmc pass per counter group) into profiles/pmc_summary.json,
the file bench.py reads for `roofline.traffic` and `roofline.valu`.

    python tools/pmc_summary.py gpurun_out/<pmc dir>[,<another pmc dir>] [--out profiles/pmc_summary.json]

Per workload (recognised by the blend kernels' grid size) and stage:
  hbm_bytes   (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch — FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
              reports half of the bytes of wide (16 B / lane) reads (MI355X_MICROARCH.md, "HBM"), hence the factor 2;
              WRITE_SIZE is uncalibrated (float atomics count as writes).  `binning` sums its kernels.
  insts_valu / insts_salu / insts_lds     dynamic instruction counts per launch (SQ_INSTS_*)
  shader_clock_ghz                        GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of the same pass
  issue_frac_all_plain_2cyc               insts_valu * 2 cycles / (1024 SIMDs * kernel cycles): every vector instruction
                                          priced at the plain wave64 rate of the SIMD-32 — a LOWER bound of how full the
                                          vector pipe is
  cycles_per_inst_mix, issue_frac_mix_weighted   the same with the kernel's STATIC instruction mix (llvm-objdump of
                                          libscg_raster.so) weighted by the issue costs tools/probes/clock_probe.hip
                                          measured on the MI355X: plain 2, DPP / v_cndmask / v_readlane / v_cmp-to-SGPR 4,
                                          transcendental 8, packed fp32 4, v_permlane*_swap 7 cycles per wave64 instruction
  mean_waves_per_simd                     SQ_WAVE_CYCLES * 4 / (1024 * kernel cycles): waves resident on average
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE_OF = {"geometry_forward_kernel": "geometry_forward", "geometry_hist_kernel": "geometry_forward",
            "geometry_backward_kernel": "geometry_backward",
            # (the model path's kernels — scg_forward_model / scg_backward_model — only run in bench legs the counter passes
            # switch off: no entry here, so that they can never be averaged into the operator's stages)
            "blend_forward_kernel": "blend_forward", "tile_blend_forward_kernel": "blend_forward",
            "blend_backward_kernel": "blend_backward"}
BINNING = ("tile_hist_kernel", "table_colscan_kernel", "tile_scatter_kernel", "tile_sort_kernel", "tile_sort_rare_kernel",
           "tile_split_long_kernel")
# Kernels launched with one workgroup per slice of the Gaussians (128 or 256 slices whatever the image): the grid does not
# say which workload a dispatch belongs to — tools/_workload_tag.py: the forward-blend dispatch that follows does.
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _workload_tag as wt                                                                   # noqa: E402
BLEND_GRID = {774144: "S2", 2088960: "S3", 522240: "S4", 65536: "S1", 196608: "S2r8"}
P_OF = {"S2": 200000, "S3": 500000, "S4": 1000000, "S1": 10000, "S2r8": 200000}          # Gaussians of the named workloads


COST = {"plain": 2.0, "dpp": 4.0, "trans": 8.0, "packed": 4.0, "swap": 7.0}
DEFAULT_GHZ = 2.25


def short(name):
    return name.split("(")[0].replace("scg::", "").replace("void ", "").split("<")[0].strip()


def static_mix():
    """{kernel: mean cycles per vector instruction} from the ISA hipcc emits for csrc/blend.hip and csrc/geometry.hip
    with the build's own flags (static mix of the whole kernel body: a proxy for the dynamic mix of its hot loop)."""
    sys.path.insert(0, ROOT)
    out = {}
    try:
        from scgaussian_amd import build as B
        hipcc = B._hipcc()
    except Exception:
        return out
    for src, extra in (("blend.hip", B.SOURCES["blend.hip"]), ("geometry.hip", B.SOURCES["geometry.hip"])):
        cmd = [hipcc] + [f for f in B.COMMON if f != "-fPIC"] + list(extra) + ["-S", "--cuda-device-only",
                                                                           os.path.join(B.CSRC, src), "-o", "-"]
        try:
            asm = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
        except (OSError, subprocess.CalledProcessError):
            continue
        cur = None
        counts = collections.defaultdict(collections.Counter)
        for line in asm.splitlines():
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1)
                continue
            if line.strip().startswith("s_endpgm"):
                cur = None
            m = re.match(r"^\s+(v_\w+)", line)
            if not m or cur is None:
                continue
            op = m.group(1)
            if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
                k = "trans"
            elif "permlane" in op and "swap" in op:
                k = "swap"
            elif op.startswith("v_pk_") and op.endswith("f32"):
                k = "packed"
            elif "_dpp" in op or op.startswith(("v_cndmask", "v_readlane", "v_writelane", "v_readfirstlane")) or \
                    (op.startswith("v_cmp") and op.endswith("_e64")):
                k = "dpp"
            elif op.startswith(("v_mfma", "v_accvgpr")):
                continue
            else:
                k = "plain"
            counts[cur][k] += 1
        for sym, c in counts.items():
            n = sum(c.values())
            for key in STAGE_OF:
                if (str(len(key)) + key) in sym and n:          # itanium-mangled name component
                    out[key] = round(sum(COST[k] * v for k, v in c.items()) / n, 3)
    return out


def lib_sha256(path=None):
    """sha256 of the library the counters were collected on (the in-tree libscg_raster.so unless given)."""
    import hashlib
    path = path or os.path.join(ROOT, "scgaussian_amd", "libscg_raster.so")
    try:
        with open(path, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()
    except OSError:
        return None


def kernel_source_sha256():
    """sha256 over the sources + headers of the rasterizer path's kernels (the ones the counters are of): survives a rebuild on
    another box (the .so itself is not bit-reproducible across toolchain installs), changes with any change to those kernels.
    The kernels of the rows next to the path (image loss, match loss, 3-NN) are in the same library and have no counters here."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "scgaussian_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")) and f not in ("loss.hip", "matchloss.hip", "knn.hip"):
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    with open(os.path.join(ROOT, "include", "scg_raster.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def stamp():
    """What the counters belong to: bench.py refuses to print them next to a different library (roofline.traffic_source)."""
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    return {"lib_sha256": lib_sha256(), "kernel_source_sha256": kernel_source_sha256(), "commit": commit}


def summarise(src, mix):
    """{workload: {stage: counters}} of the pass directories `src` (comma separated)."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in sorted(p for one in src.split(",") for p in glob.glob(os.path.join(one, "p*"))):    # (several pass directories: a,b)
        cc = os.path.join(d, "c_counter_collection.csv")
        kt = os.path.join(d, "c_kernel_trace.csv")
        if not os.path.exists(cc):
            continue
        has_grbm = False
        rows = list(csv.DictReader(open(cc)))
        gk = "Grid_Size" if rows and "Grid_Size" in rows[0] else "Grid_Size_X"
        tag = wt.tags(rows, gk)
        for r in rows:
            key = (short(r["Kernel_Name"]), wt.norm_grid(r["Kernel_Name"], int(r[gk])))
            if key[0] in ("geometry_hist_kernel", "tile_hist_kernel", "tile_scatter_kernel"):
                key = (key[0], "lds:%s" % BLEND_GRID.get(tag.get(int(r["Dispatch_Id"]))))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            has_grbm |= r["Counter_Name"] == "GRBM_GUI_ACTIVE"
        if has_grbm and os.path.exists(kt):
            krows = list(csv.DictReader(open(kt)))
            ktag = wt.tags(krows, "Grid_Size_X")
            for r in krows:
                key = (short(r["Kernel_Name"]), wt.norm_grid(r["Kernel_Name"], int(r["Grid_Size_X"])))
                if key[0] in ("geometry_hist_kernel", "tile_hist_kernel", "tile_scatter_kernel"):
                    key = (key[0], "lds:%s" % BLEND_GRID.get(ktag.get(int(r["Dispatch_Id"]))))
                dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    mean = lambda v: sum(v) / len(v) if v else None                  # noqa: E731
    # workloads by the blend grid (tiles * 4 * 64 lanes)
    grids = {774144: "S2", 2088960: "S3", 522240: "S4", 65536: "S1", 196608: "S2r8"}
    out = {}
    blend_keys = [k for k in agg if k[0] in ("blend_forward_kernel", "tile_blend_forward_kernel", "blend_backward_kernel")
                  and k[1] in grids]
    for kname, grid in blend_keys:
        wl = grids[grid]
        out.setdefault(wl, {})
    # geometry kernels: P-sized grids -> attribute by nearest workload P
    p_of = {"S2": 200192, "S3": 500224, "S4": 1000192, "S1": 10240, "S2r8": 200192}
    for (kname, grid), ctr in agg.items():
        stage = STAGE_OF.get(kname)
        # the one-wave forward blend only runs in the first (staged) call of a shape: when the sorting forward blend of the
        # same grid was captured too, IT is the stage's kernel
        if kname == "blend_forward_kernel" and ("tile_blend_forward_kernel", grid) in agg:
            continue
        wls = []
        if (kname.startswith("blend_") or kname == "tile_blend_forward_kernel") and grid in grids:
            wls = [grids[grid]]
        elif kname == "geometry_hist_kernel":
            wls = [w for w in out if not w.startswith("_") and grid == "lds:%s" % w]
        elif kname.startswith("geometry_"):
            if kname == "geometry_forward_kernel" and any(k[0] == "geometry_hist_kernel" for k in agg):
                continue                 # (first, staged call of a shape only: the histogramming kernel is the stage's kernel)
            # grids: ceil(P / 256) * 256 (geometry forward, un-staged backward) or ceil(P / 60) * 64 (the backward's single-wave
            # workgroups of sixty Gaussians)
            wls = [w for w in out if not w.startswith("_") and
                   grid in (p_of.get(w), -(-P_OF.get(w, 0) // 60) * 64)]
        for wl in wls:
            e = {}
            f, w_ = mean(ctr.get("FETCH_SIZE", [])), mean(ctr.get("WRITE_SIZE", []))
            if f is not None and w_ is not None:
                e["hbm_bytes"] = int((2 * f + w_) * 1024)
            for c, name in (("SQ_INSTS_VALU", "insts_valu"), ("SQ_INSTS_SALU", "insts_salu"), ("SQ_INSTS_LDS", "insts_lds")):
                if mean(ctr.get(c, [])) is not None:
                    e[name] = int(mean(ctr[c]))
            g, t = mean(ctr.get("GRBM_GUI_ACTIVE", [])), mean(dur.get((kname, grid), []))
            if g and t:
                cycles = g / 8.0
                if t < 50_000:           # GRBM_GUI_ACTIVE brackets more than a short kernel: price those at the clock the
                    cycles = t * DEFAULT_GHZ     # long kernels of the same pass sustained
                e["kernel_cycles"] = int(cycles)
                e["shader_clock_ghz"] = round(cycles / t, 3)
                if "insts_valu" in e:
                    e["issue_frac_all_plain_2cyc"] = round(e["insts_valu"] * 2.0 / (1024 * cycles), 3)
                    if kname in mix:
                        e["cycles_per_inst_mix"] = mix[kname]
                        e["issue_frac_mix_weighted"] = round(min(1.0, e["insts_valu"] * mix[kname] / (1024 * cycles)), 3)
                wc = mean(ctr.get("SQ_WAVE_CYCLES", []))
                if wc:
                    e["mean_waves_per_simd"] = round(wc * 4.0 / (1024 * cycles), 2)
            out[wl][stage] = e
    # binning = the sum of its kernels.  Their grids identify the workload: sort = tiles * 256 threads, column scan =
    # ceil(tiles / 64) * 1024; histogram / scatter grids follow the slice count (128 slices up to ~2 M instances: the S2
    # class, more above: the S3 class); the rare-size launch is idle in these scenes (no traffic).
    tiles = {"S2": 3024, "S3": 8160, "S4": 2040, "S1": 256, "S2r8": 768}
    for wl in [w for w in out if not w.startswith("_")]:
        tn = tiles[wl]
        tot, parts = 0.0, {}
        for (kname, grid), ctr in agg.items():
            if kname not in BINNING:
                continue
            # kernels of the staged first call of a shape only, when the one-call path's kernels that took their work over
            # were captured: not part of the stage that runs
            if (kname == "tile_hist_kernel" and any(k[0] == "geometry_hist_kernel" for k in agg)) or \
                    (kname == "tile_sort_kernel" and any(k[0] == "tile_blend_forward_kernel" for k in agg)):
                continue
            f, w_ = mean(ctr.get("FETCH_SIZE", [])), mean(ctr.get("WRITE_SIZE", []))
            if f is None or w_ is None:
                continue
            if kname == "tile_sort_kernel":
                mine = grid == tn * 256
            elif kname == "table_colscan_kernel":
                mine = grid == (tn + 63) // 64 * 1024
            elif kname in ("tile_hist_kernel", "tile_scatter_kernel"):
                mine = grid == "lds:%s" % wl
            else:
                mine = False
            if mine:
                b_ = (2 * f + w_) * 1024
                parts[kname] = parts.get(kname, []) + [b_]
        for kname, v in parts.items():
            tot += sum(v) / len(v)
        if parts:
            out[wl]["binning"] = {"hbm_bytes": int(tot), "per_kernel": {k: int(sum(v) / len(v)) for k, v in parts.items()}}
    return out


def main():
    """pmc_summary.py <dirs>[,<dirs>...] [<dirs>@<suffix> ...] [--out file]: the first argument's workloads keep their names
    (S2, S3, S4); every further `dirs@suffix` argument is a collection of the same bench at another setting — the SH degrees
    the reference trains at: `gpurun_out/r06z/pmc_deg0@deg0` — and its workloads are stored as `<workload>_<suffix>`."""
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--out" in sys.argv:
        args.remove(sys.argv[sys.argv.index("--out") + 1])
    dst = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "profiles", "pmc_summary.json")
    mix = static_mix()
    src = args[0]
    out = {"_note": __doc__.split("Per workload")[1].strip().splitlines()[0:1],
           "_source": "+".join(os.path.basename(one.rstrip("/")) for a in args for one in a.split("@")[0].split(",")),
           "_mix_cycles_per_inst": mix, "_stamp": stamp()}
    out.update(summarise(src, mix))
    for extra in args[1:]:
        dirs, _, suffix = extra.partition("@")
        for wl, v in summarise(dirs, mix).items():
            out[f"{wl}_{suffix}" if suffix else wl] = v
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
