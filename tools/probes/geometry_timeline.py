#!/usr/bin/env python3
"""Per-wave timeline of geometry_hist_kernel (probe build: python -m scgaussian_amd.build --tag=tl -DSCG_PROBE_TIMELINE): per wave
the 100 MHz wall clock summed over its chunks — inputs + cull / covariance math | flush of the previous chunk + histogram walk
(the SH loads in flight) | remaining wait for the SH record | SH evaluation + parking the outputs — logged behind the n_tiles words
of ScgFrame.tile_cost_out.        tools/probes/geometry_timeline.py [S2|S3|S4]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scgaussian_amd import _lib, rasterizer as R, synthetic as syn          # noqa: E402
import bench                                                                   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "S3"
_lib._lib = _lib.open_library(_lib.LIB_PATH.replace(".so", "_tl.so"))
dev = torch.device("cuda", 0)
w = syn.WORKLOADS[name]
P, W, H = w["P"], w["width"], w["height"]
sc = syn.make_scene(P, W, H, seed=0).to(dev)
params = [sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations]
sett = bench.settings_for(syn.default_camera(W, H), 3, torch.zeros(3, device=dev), dev)
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
LOG = 65792 + 32768 + (n_tiles + 8) * 40 + (n_tiles + 8) * 16 + 64   # every probe of the tl build logs: all regions must exist
orig = R._hints_for


def big_hints(*a):
    h = orig(*a)
    if h.cost is not None and h.cost[0].numel() < LOG:
        n_tiles = h.cost[0].numel()
        h.cost = [torch.zeros(n_tiles + LOG, dtype=torch.int32, device=dev) for _ in range(2)]
    return h


R._hints_for = big_hints
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
with torch.no_grad():
    for _ in range(5):
        bench.render_once(sett, params)
    torch.cuda.synchronize()
    h = next(iter(R._CAM_HINTS.values()))
    for c in h.cost:
        c[n_tiles:].zero_()
    bench.render_once(sett, params)
    torch.cuda.synchronize()
log = h.cost[h.cur].cpu().numpy().astype(np.uint32)[n_tiles + 65792:]
rec = log[: (log.size // 8) * 8].reshape(-1, 8)
rec = rec[(rec[:, 7] >> 8) == 0xC0FFEE]
it = (rec[:, 7] & 0xFF).astype(np.int64)
t0 = rec[:, 0].astype(np.int64)
start = t0.min()
us = lambda x: x / 100.0                                                      # noqa: E731
print(f"{name}: {len(rec)} waves logged, chunks per wave mean {it.mean():.2f} (max {it.max()}); "
      f"kernel span first entry -> last exit {us(rec[:, 6].astype(np.int64).max() - start):.1f} us")
for label, v in (("loop start after first", t0 - start), ("inputs + cull/cov math", rec[:, 1]), ("flush + histogram walk", rec[:, 2]),
                 ("rest of the SH wait", rec[:, 3]), ("SH eval + park", rec[:, 4]),
                 ("loop end after first", rec[:, 5].astype(np.int64) - start), ("exit after first", rec[:, 6].astype(np.int64) - start)):
    v = us(np.asarray(v, dtype=np.int64))
    print(f"  {label:26s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  "
          f"p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f} us")
# per workgroup (= per compute unit: one 16-wave workgroup each) and per XCD (workgroup b runs on XCD b % 8)
nw = 16
wg = np.arange(len(rec)) // nw if len(rec) == 4096 else None
if wg is not None:
    ends = us(rec[:, 5].astype(np.int64) - start).reshape(-1, nw)
    exits = us(rec[:, 6].astype(np.int64) - start).reshape(-1, nw)
    slow, mean_ = ends.max(1), ends.mean(1)
    print(f"  per workgroup: slowest wave's loop end mean {slow.mean():.1f} p90 {np.percentile(slow, 90):.1f} max {slow.max():.1f} us; "
          f"mean wave's loop end {mean_.mean():.1f}; exit mean {exits.max(1).mean():.1f} max {exits.max(1).max():.1f}")
    for x in range(8):
        m = (np.arange(256) % 8) == x
        print(f"    XCD {x}: slowest-wave loop end mean {slow[m].mean():6.1f} max {slow[m].max():6.1f}   chunks per workgroup {it.reshape(-1, nw)[m].sum(1).mean():.1f}")
