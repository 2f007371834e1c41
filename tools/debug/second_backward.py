import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import parity_utils as pu
from scgaussian_amd import synthetic as syn, rasterizer as R
P, W, H = 6000, 200, 120
sc = syn.make_scene(P, W, H, seed=21, log_scale_mean=-2.6)
cam = syn.orbit_camera(W, H, -7.0, 4.0, 7.0)
grads = syn.make_upstream_grads(W, H, seed=4)
dev = torch.device("cuda", 0)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    lv = {k: v.detach().to(dev).requires_grad_(True) for k, v in pu.run_oracle_inputs(sc, cam, 3, 1.0, "sh_sr").items()}
    kw = {k: v for k, v in lv.items() if k not in ("means3D", "means2D", "opacities")}
    c, r, d, a = R.GaussianRasterizer(pu.hip_settings(cam, 3, (0.3, 0.1, 0.2)))(means3D=lv["means3D"], means2D=lv["means2D"],
                                                                                  opacities=lv["opacities"], **kw)
    loss = (c * grads[0].to(dev)).sum() + (d * grads[1].to(dev)).sum() + (a * grads[2].to(dev)).sum()
    g1 = torch.autograd.grad(loss, list(lv.values()), retain_graph=True)
    g2 = torch.autograd.grad(loss, list(lv.values()))
    for x, y, k in zip(g1, g2, lv):
        diff = (x - y).abs()
        tol = 1e-4 * x.abs() + 1e-6 * x.abs().max()
        n = int((diff > tol).sum())
        if n:
            bad += 1
            idx = torch.nonzero(diff > tol)[:5].tolist()
            print(it, k, "elements outside", n, "max diff", float(diff.max()), "max", float(x.abs().max()), idx,
                  [(float(x[tuple(i)]), float(y[tuple(i)])) for i in idx[:3]], flush=True)
print("bad", bad)
