"""Per-wave timeline of the blend backward (library built with -DSCG_ABL_BWD_TIMING: every wave leaves start / end stamps,
its trips, the time it spent in flushes and in chunk staging where the forward's n_contrib was).  Profiling aid only.

    python -m scgaussian_amd.build --tag=btime -DSCG_ABL_BWD_TIMING
    SCG_LIB_PATH=$PWD/scgaussian_amd/libscg_raster_btime.so python tools/probes/bwd_wave_timeline.py S2
"""
import heapq
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from scgaussian_amd import rasterizer as R, synthetic as syn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "S2"
w = syn.WORKLOADS[name]
P, W, H = w["P"], w["width"], w["height"]
dev = torch.device("cuda:0")
sc = syn.make_scene(P, W, H, seed=0).to(dev)
setts = bench.settings_for(bench.make_views(W, H)[0], 3, torch.zeros(3, device=dev), dev)
ups = tuple(t.to(dev) for t in syn.make_upstream_grads(W, H, seed=10))
inputs = (sc.means3D, sc.opacities, sc.shs, None, sc.scales, sc.rotations, None)
for _ in range(3):
    fs = R.forward_stages(setts, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations,
                          prepare_backward=True)
    R.backward_stages(setts, inputs, fs, ups[0], ups[1], ups[2])
torch.cuda.synchronize()
st = fs["n_contrib"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
rows = st[::8]                                            # first pixel row of every quadrant row
nq_x = (W + 7) // 8
v = np.stack([rows[:, k::8][:, :nq_x] for k in range(8)], -1).reshape(-1, 8)     # (quadrants, 8 words)
if W % 8:
    v = v[np.arange(v.shape[0]) % nq_x != nq_x - 1]       # a clipped last column of quadrants has no complete record
ok = v[:, 7] == 0xB00B
v = v[ok]
start, end, trips, t_flush, flushes, t_stage, chunks = [v[:, k].astype(np.float64) for k in range(7)]
origin = start.min()
s_us, e_us = (start - origin) / 100.0, ((end - origin) % 2 ** 32) / 100.0
dur = e_us - s_us
pct = lambda x: {k: round(float(np.percentile(x, p)), 3) for k, p in (("p5", 5), ("p50", 50), ("p95", 95), ("max", 100))}
edges = np.arange(0, e_us.max() + 5, 5.0)
conc = [int(((s_us <= t) & (e_us > t)).sum()) for t in edges]


def makespan(durs, slots):
    h = [0.0] * min(slots, len(durs))
    heapq.heapify(h)
    end_ = 0.0
    for d in durs:
        t = heapq.heappop(h) + d
        end_ = max(end_, t)
        heapq.heappush(h, t)
    return end_


slots = max(conc)
order = np.argsort(s_us, kind="stable")
walk = dur - t_flush / 100.0 - t_stage / 100.0
out = {
    "workload": name, "waves_with_work": int(ok.sum()), "kernel_span_us": float(e_us.max()), "slots": int(slots),
    "wave_dur_us": pct(dur), "trips_per_wave": pct(trips), "chunks_per_wave": pct(chunks),
    "us_per_flush": pct(t_flush / 100.0 / np.maximum(flushes, 1)), "flush_share_of_wave": pct(t_flush / 100.0 / dur),
    "staging_us_per_chunk": pct(t_stage / 100.0 / np.maximum(chunks, 1)), "staging_share_of_wave": pct(t_stage / 100.0 / dur),
    "per_trip_us_outside_flush_and_staging": pct(walk / np.maximum(trips, 1)),
    "last_start_us": float(s_us.max()),
    "dur_first_round_vs_rest_us": [float(np.median(dur[order[:slots]])), float(np.median(dur[order[slots:]])) if len(dur) > slots else None],
    "sched_sim_us": {"as_launched": round(makespan(dur[order], slots), 2), "longest_first": round(makespan(np.sort(dur)[::-1], slots), 2),
                     "mean_bound": round(float(dur.sum()) / slots, 2)},
    "corr_trips_dur": round(float(np.corrcoef(trips, dur)[0, 1]), 3),
    "resident_waves_every_5us": conc,
}
# modelled work (a trips + b chunks + c, fitted on the waves of the full phase) replayed in other orders: whole tiles keyed by
# their busiest quadrant, and every wave on its own
A = np.stack([trips, chunks, np.ones_like(trips)], 1)
full = s_us < np.percentile(s_us, 55)
coef, *_ = np.linalg.lstsq(A[full], dur[full], rcond=None)
work = A @ coef
qidx = np.flatnonzero(ok)
nq_x_all = (W + 7) // 8
if W % 8:
    raise SystemExit("quadrant -> tile map below assumes W % 8 == 0")
qy, qx = np.divmod(qidx, nq_x_all)
tile_q = (qy // 2) * ((W + 15) // 16) + (qx // 2)
n_t = int(tile_q.max()) + 1
key = np.zeros(n_t); np.maximum.at(key, tile_q, work)
torder = np.argsort(-key, kind="stable"); rank = np.empty(n_t, dtype=np.int64); rank[torder] = np.arange(n_t)
out["dur_fit_us_per_trip_chunk_const"] = [round(float(x), 4) for x in coef]
out["sched_sim_modelled_work_us"] = {
    "as_launched": round(makespan(work[order], slots), 2),
    "tile_max_of_backward_work": round(makespan(work[np.argsort(rank[tile_q], kind="stable")], slots), 2),
    "longest_first": round(makespan(np.sort(work)[::-1], slots), 2), "mean_bound": round(float(work.sum()) / slots, 2)}
# what would two waves per quadrant (front / back half of the walk, 3 us of fixed cost each) be worth?
half = np.repeat((dur[order] - 3.0) / 2 + 3.0, 2)
out["sched_sim_split_in_two_us"] = {"as_launched": round(makespan(half, slots), 2),
                                    "longest_first": round(makespan(np.sort(half)[::-1], slots), 2),
                                    "mean_bound": round(float(half.sum()) / slots, 2)}
print(json.dumps(out))
