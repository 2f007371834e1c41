#!/usr/bin/env python3
"""Per-wave timeline of tile_scatter_kernel (probe build: python -m scgaussian_amd.build --tag=tl -DSCG_PROBE_TIMELINE).
The kernel logs, per wave, the 100 MHz wall clock at entry / after its prologue / at its end, the time inside its walks, the
number of walks and the sum of their trip counts — behind the n_tiles words of ScgFrame.tile_cost_out (this script hands the
library a larger buffer there).      tools/probes/scatter_timeline.py [S2|S3|S4]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scgaussian_amd import _lib, rasterizer as R, synthetic as syn          # noqa: E402
sys.path.insert(0, ROOT)
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
with torch.no_grad():
    for _ in range(5):
        bench.render_once(sett, params)
    torch.cuda.synchronize()
h = next(iter(R._CAM_HINTS.values()))
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
log = h.cost[h.cur].cpu().numpy().astype(np.uint32)[n_tiles: n_tiles + 65792]
rec = log[: (log.size // 8) * 8].reshape(-1, 8)
rec = rec[rec[:, 7] == 0xC0FFEE]
t0, t1, t2 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64), rec[:, 2].astype(np.int64)
start = t0.min()
us = lambda x: x / 100.0                                                      # noqa: E731  (100 MHz ticks -> us)
print(f"{name}: {len(rec)} scattering waves logged; kernel span (first entry -> last exit) {us(t2.max() - start):.1f} us")
for label, v in (("entry after first entry", t0 - start), ("prologue", t1 - t0), ("loop + walks", t2 - t1),
                 ("inside walks", rec[:, 3].astype(np.int64)), ("exit after first entry", t2 - start)):
    v = us(v)
    print(f"  {label:26s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  "
          f"p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f} us")
print(f"  walks per wave mean {rec[:, 4].mean():.2f}, trip-count sum per wave mean {rec[:, 5].mean():.1f}, "
      f"Gaussians per wave mean {rec[:, 6].mean():.0f}")
