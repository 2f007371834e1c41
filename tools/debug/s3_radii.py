"""Which Gaussians of the S3 scene get a different radius / rectangle from the HIP geometry stage and the CPU oracle."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import parity_utils as pu
from oracle import torch_rasterizer as orc
from scgaussian_amd import synthetic as syn, rasterizer as R

name = sys.argv[1] if len(sys.argv) > 1 else "S3"
w = syn.WORKLOADS[name]
P, W, H, deg = w["P"], w["width"], w["height"], 3
sc = syn.make_scene(P, W, H, seed=0)
cam = syn.default_camera(W, H)
bg = (0.2, 0.1, 0.3)
st = pu.oracle_settings(cam, deg, bg)
with torch.no_grad():
    pre = orc.preprocess(sc.means3D, torch.zeros(P, 3), sc.opacities, st, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
dev = torch.device("cuda")
hs = pu.hip_settings(cam, deg, bg)
fs = R.forward_stages(hs, sc.means3D.to(dev), sc.opacities.to(dev), shs=sc.shs.to(dev), scales=sc.scales.to(dev), rotations=sc.rotations.to(dev))
radii = fs["radii"].cpu()
rects = fs["rects"].cpu().numpy().view(np.uint32).reshape(P, 2)
splats = fs["splats"].cpu()
bad = (radii != pre["radii"]).nonzero().flatten()
print(name, "mismatching radii:", bad.numel(), "of", P)
orect = pre["rect"].numpy()
for i in bad[:20].tolist():
    rx = rects[i]
    print(i, "hip radius", int(radii[i]), "oracle", int(pre["radii"][i]), "hip rect", (rx[0] & 0xffff, rx[0] >> 16, rx[1] & 0xffff, rx[1] >> 16),
          "oracle rect", orect[i], "vis", bool(pre["visible"][i]), "oracle xy", pre["xy"][i].tolist(), "hip xy", splats[i, :2].tolist(),
          "depth", float(pre["depth"][i]), "conic o", pre["conic"][i].tolist(), "conic h", [float(splats[i, 2]), float(splats[i, 3]), float(splats[i, 4])])
