#!/usr/bin/env python3
"""Per-wave timeline of tile_blend_forward_kernel (probe build --tag=tl -DSCG_PROBE_TIMELINE): entry | end of the tile's sort |
behind the barrier | end of the quadrant's walk, per quadrant wave, 100 MHz wall clock.   tools/probes/blend_timeline.py [S2|S3|S4]"""
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
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
sc = syn.make_scene(P, W, H, seed=0).to(dev)
params = [sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations]
sett = bench.settings_for(syn.default_camera(W, H), 3, torch.zeros(3, device=dev), dev)
OFF = 65792 + 32768
LOG = 65792 + 32768 + (n_tiles + 8) * 40 + (n_tiles + 8) * 16 + 64   # every probe of the tl build logs: all regions must exist
orig = R._hints_for


def big_hints(*a):
    h = orig(*a)
    if h.cost is not None and h.cost[0].numel() < LOG:
        h.cost = [torch.zeros(n_tiles + LOG, dtype=torch.int32, device=dev) for _ in range(2)]
    return h


R._hints_for = big_hints
with torch.no_grad():
    for _ in range(6):
        bench.render_once(sett, params)
    torch.cuda.synchronize()
h = next(iter(R._CAM_HINTS.values()))
log = h.cost[h.cur].cpu().numpy().astype(np.uint32)[n_tiles + OFF:]
rec = log[: (log.size // 8) * 8].reshape(-1, 8)
rec = rec[rec[:, 7] == 0xB1E9D000]
t0, t1, t2, t3 = (rec[:, i].astype(np.int64) for i in range(4))
start = t0.min()
us = lambda x: x / 100.0                                                      # noqa: E731
print(f"{name}: {len(rec)} quadrant waves logged; list length mean {rec[:, 4].mean():.0f} max {rec[:, 4].max()}; "
      f"kernel span {us(t3.max() - start):.1f} us")
for label, v in (("entry after first entry", t0 - start), ("sort", t1 - t0), ("barrier", t2 - t1), ("walk", t3 - t2),
                 ("life of a wave", t3 - t0), ("exit after first entry", t3 - start)):
    v = us(v)
    print(f"  {label:26s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  "
          f"p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f} us")
# list-scheduler replay at WORKGROUP granularity (a tile's four quadrant waves start together and hold their slots — LDS — until
# the last one ends): 8 workgroups per compute unit = 2 048 slots
import heapq                                                                  # noqa: E402
tiles = rec[:, 5].astype(np.int64)
wg_start = {}
wg_end = {}
for t, a, b in zip(tiles, t0, t3):
    wg_start[t] = min(wg_start.get(t, a), a)
    wg_end[t] = max(wg_end.get(t, b), b)
ts = np.array(sorted(wg_start))
st = np.array([wg_start[t] for t in ts])
life = us(np.array([wg_end[t] - wg_start[t] for t in ts]))
print(f"  workgroups: {len(ts)}, life mean {life.mean():.1f} p90 {np.percentile(life, 90):.1f} max {life.max():.1f} us; "
      f"sum / 2048 slots = {life.sum() / 2048:.1f} us")
for label, order in (("as launched", np.argsort(st, kind="stable")), ("longest first (oracle)", np.argsort(-life, kind="stable"))):
    heap = [0.0] * 2048
    heapq.heapify(heap)
    end = 0.0
    for i in order:
        t = heapq.heappop(heap) + life[i]
        end = max(end, t)
        heapq.heappush(heap, t)
    print(f"  greedy replay on 2 048 workgroup slots, {label}: makespan {end:.1f} us")
srt = log[: (log.size // 8) * 8].reshape(-1, 8)
srt = srt[srt[:, 7] == 0x50B70000]
if len(srt):
    c = [srt[:, k].astype(np.int64) for k in range(7)]
    e0 = c[6]
    print(f"  inside the sort ({len(srt)} tiles; thread 0's clock):")
    for label, v in (("ids + keys fetched, min/max", c[0] - e0), ("barrier 1", c[1] - c[0]), ("bucket atomics", c[2] - c[1]),
                     ("barrier 2", c[3] - c[2]), ("scan of the counts (2 barriers)", c[4] - c[3]), ("scatter into LDS + barrier", c[5] - c[4])):
        v = us(v)
        print(f"    {label:32s} mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f} us")
