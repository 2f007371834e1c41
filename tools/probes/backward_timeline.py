#!/usr/bin/env python3
"""Per-wave timeline of blend_backward_kernel (probe build --tag=tl -DSCG_PROBE_TIMELINE): entry / exit of every quadrant wave
(100 MHz wall clock) and the list length it walked — and what a list scheduler would make of the same waves: longest-first
greedy on 1 024 SIMDs x 6 slots (what the kernel has; x 7 beside it) with the measured durations (the lower bound of what a better LAUNCH ORDER could buy).
    tools/probes/backward_timeline.py [S2|S4]"""
import heapq
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scgaussian_amd import _lib, rasterizer as R, synthetic as syn          # noqa: E402
import bench                                                                   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "S2"
_lib._lib = _lib.open_library(_lib.LIB_PATH.replace(".so", "_tl.so"))
dev = torch.device("cuda", 0)
w = syn.WORKLOADS[name]
P, W, H = w["P"], w["width"], w["height"]
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
sc = syn.make_scene(P, W, H, seed=0).to(dev)
params = [sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations]
for p in params:
    p.requires_grad_(True)
means, shs, opac, scales, rots = params
sett = bench.settings_for(syn.default_camera(W, H), 3, torch.zeros(3, device=dev), dev)
rast = R.GaussianRasterizer(sett)
ups = tuple(t.to(dev) for t in syn.make_upstream_grads(W, H, seed=10))
OFF = 65792 + 32768 + (n_tiles + 8) * 40
LOG = 65792 + 32768 + (n_tiles + 8) * 40 + (n_tiles + 8) * 16 + 64   # every probe of the tl build logs: all regions must exist
orig = R._hints_for


def big_hints(*a):
    h = orig(*a)
    if h.cost is not None and h.cost[0].numel() < LOG:
        h.cost = [torch.zeros(n_tiles + LOG, dtype=torch.int32, device=dev) for _ in range(2)]
    return h


R._hints_for = big_hints
for _ in range(6):
    for p in params:
        p.grad = None
    c, radii, d, a = rast(means3D=means, means2D=torch.zeros_like(means, requires_grad=True), opacities=opac, shs=shs,
                          scales=scales, rotations=rots)
    torch.autograd.backward([c, d, a], list(ups))
torch.cuda.synchronize()
h = next(iter(R._CAM_HINTS.values()))
log = h.cost[h.cur].cpu().numpy().astype(np.uint32)[n_tiles + OFF:]
rec = log[: (log.size // 4) * 4].reshape(-1, 4)
rec = rec[(rec[:, 3] >> 20) == 0xBAC]
t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
start = np.percentile(t0, 0.5)          # (a quadrant that returned at once this time keeps an older record)
us = lambda x: x / 100.0                                                      # noqa: E731
dur = us(t1 - t0)
trips, walked = (rec[:, 2] >> 16).astype(np.float64), (rec[:, 2] & 0xFFFF).astype(np.float64)
rec = rec.copy()
rec[:, 2] = walked.astype(np.uint32)
print(f"{name}: {len(rec)} quadrant waves walked a list (mean {rec[:, 2].mean():.0f} entries, {trips.mean():.0f} trips); kernel span first entry -> last exit "
      f"{us(t1.max() - start):.1f} us; sum of wave lives {dur.sum() / 1e3:.2f} ms = {dur.sum() / 6144:.1f} us on 6 144 slots (six waves per SIMD)")
for label, v in (("entry after first entry", us(t0 - start)), ("life of a wave", dur), ("exit after first entry", us(t1 - start))):
    print(f"  {label:26s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  "
          f"p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f} us")
tq = (rec[:, 3] & 0xFFFFF).astype(np.int64)
per = (n_tiles + 7) // 8
band = (tq >> 2) // per
order_rank = np.argsort(np.argsort(t0, kind="stable"))
print("  per XCD band (contiguous eighth of the tiles): sum of wave lives / 768 slots, last exit, correlation(entry rank, life):")
for b in range(8):
    m = band == b
    if m.any():
        print(f"    band {b}: {m.sum():5d} waves  {dur[m].sum() / 768:6.1f} us of work  last exit {us(t1[m].max() - start):6.1f} us  "
              f"corr {np.corrcoef(order_rank[m], dur[m])[0, 1]:+.2f}")
print(f"  correlation of a wave's life with the list length it walked: {np.corrcoef(dur, rec[:, 2])[0, 1]:.3f}")
A = np.stack([np.ones_like(trips), trips, walked], 1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
fit = A @ coef
print(f"  life ~ {coef[0]:.1f} us + {coef[1] * 1e3:.1f} ns x trips + {coef[2] * 1e3:.1f} ns x entries walked   "
      f"(correlation of the fit with the life {np.corrcoef(fit, dur)[0, 1]:.3f}; trips alone {np.corrcoef(trips, dur)[0, 1]:.3f})")
# how many waves were alive at once (sweep over the entries and exits): the slot count the kernel really has
ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
live = np.cumsum(ev[:, 1])
mid = (ev[:, 0] >= start + 500) & (ev[:, 0] <= np.percentile(t0, 90))          # from 5 us in until the last tenth starts
print(f"  waves alive at once: max {int(live.max())}, median while waves still queue {int(np.median(live[mid])) if mid.any() else 0} "
      f"(7 per SIMD = 7 168, 6 per SIMD = 6 144)")
for slots_per_simd in (6, 7):
    slots = 1024 * slots_per_simd
    for label, order in (("as launched", np.argsort(t0, kind="stable")), ("longest first (oracle)", np.argsort(-dur, kind="stable"))):
        heap = [0.0] * slots
        heapq.heapify(heap)
        end = 0.0
        for i in order:
            t = heapq.heappop(heap) + dur[i]
            end = max(end, t)
            heapq.heappush(heap, t)
        print(f"  greedy replay on {slots} slots, {label}: makespan {end:.1f} us  (waves keep their measured lives: an upper bound of "
              f"what the order alone can change)")
