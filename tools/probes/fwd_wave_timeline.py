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
if lastq is not None:
    out["last_blended_mean"] = float(lastq.mean())
    out["last_blended_max"] = float(lastq.max())
print(json.dumps(out))
