"""Per-wave start / end times of the forward blend (library built with -DSCG_ABL_FWD_TIMING: final_T / n_contrib carry
wall_clock64 stamps, out_alpha the last blended index).  Profiling aid only.

    python -m scgaussian_amd.build --tag=ftime -DSCG_ABL_FWD_TIMING
    SCG_LIB_PATH=$PWD/scgaussian_amd/libscg_raster_ftime.so python tools/probes/fwd_wave_timeline.py S2
"""
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
bg = torch.zeros(3, device=dev)
setts = bench.settings_for(bench.make_views(W, H)[0], 3, bg, dev)
for _ in range(3):
    fs = R.forward_stages(setts, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
torch.cuda.synchronize()
prev_t0 = fs["final_T"].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
prev_t1 = fs["n_contrib"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
prev_dur = ((prev_t1 - prev_t0)[::8, ::8].ravel()) / 100.0
fs = R.forward_stages(setts, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
torch.cuda.synchronize()
t0 = fs["final_T"].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
t1 = fs["n_contrib"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
last = fs["alpha"][0].cpu().numpy()
ranges = fs["ranges"].cpu().numpy().astype(np.int64)
n_list = (ranges[:, 1] - ranges[:, 0])
# one sample per 8x8 quadrant
q0 = t0[::8, ::8].ravel()
q1 = t1[::8, ::8].ravel()
lastq = last.reshape(H // 8 if H % 8 == 0 else -1, 8, -1, 8).max(axis=(1, 3)).ravel() if H % 8 == 0 and W % 8 == 0 else None
origin = q0.min()
start = (q0 - origin) / 100.0          # us (100 MHz)
end = (q1 - origin) / 100.0
dur = end - start
edges = np.arange(0, end.max() + 5, 5.0)
conc = [(int(((start <= t) & (end > t)).sum())) for t in edges]
out = {
    "workload": name, "waves": int(q0.size), "kernel_span_us": float(end.max()),
    "num_rendered": int(fs["num_rendered"]), "list_len_mean": float(n_list.mean()), "list_len_max": int(n_list.max()),
    "wave_dur_us": {k: float(np.percentile(dur, p)) for k, p in [("p5", 5), ("p50", 50), ("p95", 95), ("max", 100)]},
    "start_us": {k: float(np.percentile(start, p)) for k, p in [("p50", 50), ("p68", 68), ("p90", 90), ("max", 100)]},
    "dur_of_first_8192_us": float(np.median(dur[np.argsort(start)[:8192]])),
    "dur_of_rest_us": float(np.median(dur[np.argsort(start)[8192:]])) if q0.size > 8192 else None,
    "resident_waves_every_5us": conc,
}
trips = fs["depth"][0].cpu().numpy()[::8, ::8].ravel()
chunks = fs["color"][0].cpu().numpy()[::8, ::8].ravel()
packed = fs["color"][1].cpu().numpy()[::8, ::8].ravel().astype(np.float64)
t_first = np.floor(packed / 65536.0); t_range = packed - 65536.0 * t_first
packed2 = fs["color"][2].cpu().numpy()[::8, ::8].ravel().astype(np.float64)
t_staging = np.floor(packed2 / 65536.0); t_walk = packed2 - 65536.0 * t_staging
pct = lambda v: {k: round(float(np.percentile(v, p)) / 100.0, 2) for k, p in (("p5", 5), ("p50", 50), ("p95", 95))}
out["wave_phases_us"] = {"start_to_range": pct(t_range), "range_to_first_records": pct(t_first - t_range),
                         "first_records_to_end_of_walk": pct(t_walk - t_first), "end_of_walk_to_last_store_issued": pct(dur * 100.0 - t_walk),
                         "chunk_staging_total (top of the chunk loop to the ballot)": pct(t_staging),
                         "staging_per_chunk": pct(t_staging / np.maximum(chunks, 1)),
                         "trips_per_trip": pct((t_walk - t_first - t_staging) / np.maximum(trips, 1))}
A = np.stack([trips, chunks, np.ones_like(trips)], 1)
# duration = a trips + b chunks + c, fitted separately on the waves of the full phase and of the drain
full = start < np.percentile(start, 60)
coef_full, *_ = np.linalg.lstsq(A[full], dur[full], rcond=None)
coef_drain, *_ = np.linalg.lstsq(A[~full], dur[~full], rcond=None)
out["trips_per_wave"] = {"mean": float(trips.mean()), "p5": float(np.percentile(trips, 5)), "p95": float(np.percentile(trips, 95)), "max": float(trips.max())}
out["chunks_per_wave_mean"] = float(chunks.mean())
out["dur_fit_us_per_trip_chunk_const"] = {"full_phase": [round(float(x), 4) for x in coef_full], "drain_phase": [round(float(x), 4) for x in coef_drain]}
out["corr_trips_dur"] = round(float(np.corrcoef(trips, dur)[0, 1]), 3)
work = A @ coef_full                                   # contention-free cost model of a wave
# what would the launch order be worth?  Greedy list scheduling of the measured durations on the slots the launch had
# (waves start in launch order as slots free up), in the order they ran, in the ideal order (longest first) and shortest first.
import heapq


def makespan(durs, slots):
    h = [0.0] * min(slots, len(durs))
    heapq.heapify(h)
    end = 0.0
    for d in durs:
        t = heapq.heappop(h) + d
        end = max(end, t)
        heapq.heappush(h, t)
    return end


order = np.argsort(start, kind="stable")
slots = max(conc)
dl = dur[order]
out["sched_sim_us"] = {"slots": int(slots), "as_launched": round(makespan(dl, slots), 2),
                       "longest_first": round(makespan(np.sort(dl)[::-1], slots), 2),
                       "shortest_first": round(makespan(np.sort(dl), slots), 2),
                       "mean_bound": round(float(dl.sum()) / slots, 2)}
# the order the kernels can actually use: whole tiles (their four quadrant waves start together), keyed by what a PREVIOUS
# run of the same frame measured (per-tile maximum / sum of its quadrants' durations), waves of a tile adjacent
tiles_x = (W + 15) // 16
qy_, qx_ = np.divmod(np.arange(q0.size), (W + 7) // 8)
tile_q = (qy_ // 2) * tiles_x + (qx_ // 2)
n_t = int(tile_q.max()) + 1
for keyname, red in (("tile_max_prev_run", np.maximum), ("tile_sum_prev_run", np.add)):
    key = np.zeros(n_t)
    red.at(key, tile_q, prev_dur)
    torder = np.argsort(-key, kind="stable")
    rank = np.empty(n_t, dtype=np.int64); rank[torder] = np.arange(n_t)
    worder = np.argsort(rank[tile_q], kind="stable")
    out["sched_sim_us"][keyname] = round(makespan(dur[worder], slots), 2)
out["corr_prev_run_dur"] = round(float(np.corrcoef(prev_dur, dur)[0, 1]), 3)
# the same with the modelled work (trips, chunks) instead of the measured durations, which embed where a wave ran
wl_ = work[order]
sim = {"as_launched": round(makespan(wl_, slots), 2), "longest_first": round(makespan(np.sort(wl_)[::-1], slots), 2)}
for keyname, red in (("tile_max", np.maximum), ("tile_sum", np.add)):
    key = np.zeros(n_t)
    red.at(key, tile_q, work)
    torder = np.argsort(-key, kind="stable")
    rank = np.empty(n_t, dtype=np.int64); rank[torder] = np.arange(n_t)
    sim[keyname] = round(makespan(work[np.argsort(rank[tile_q], kind="stable")], slots), 2)
key = np.zeros(n_t); np.add.at(key, tile_q, 1.0); key = n_list.astype(float)
torder = np.argsort(-key, kind="stable"); rank = np.empty(n_t, dtype=np.int64); rank[torder] = np.arange(n_t)
sim["tile_by_list_length"] = round(makespan(work[np.argsort(rank[tile_q], kind="stable")], slots), 2)
out["sched_sim_modelled_work_us"] = sim
# the eight XCD bands (tile t runs on XCD t / ceil(Tn / 8)): when does each finish, how much work did it get?
per_band = (n_t + 7) // 8
band = tile_q // per_band
out["xcd_bands"] = {"end_us": [round(float(end[band == b].max()), 1) for b in range(8)],
                    "wave_seconds_share": [round(float(dur[band == b].sum() / dur.sum()), 4) for b in range(8)]}
# does the list length predict the duration?  (the launch order sorts by it)
tile_of_q = tile_q
out["corr_listlen_dur"] = round(float(np.corrcoef(n_list[tile_of_q], dur)[0, 1]), 3)
if lastq is not None:
    out["corr_lastblended_dur"] = round(float(np.corrcoef(lastq, dur)[0, 1]), 3)
if lastq is not None:
    out["last_blended_mean"] = float(lastq.mean())
    out["last_blended_max"] = float(lastq.max())
print(json.dumps(out))
