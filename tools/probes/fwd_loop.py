"""Run the forward of a named workload N times (target for rocprofv3 PC sampling).  Profiling aid only."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from scgaussian_amd import rasterizer as R, synthetic as syn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "S3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = syn.WORKLOADS[name]
P, W, H = w["P"], w["width"], w["height"]
dev = torch.device("cuda:0")
sc = syn.make_scene(P, W, H, seed=0).to(dev)
bg = torch.zeros(3, device=dev)
setts = bench.settings_for(bench.make_views(W, H)[0], 3, bg, dev)
for _ in range(n):
    fs = R.forward_stages(setts, sc.means3D, sc.opacities, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
torch.cuda.synchronize()
print("done", fs["num_rendered"])
