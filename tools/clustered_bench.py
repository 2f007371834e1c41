#!/usr/bin/env python3
"""Training-step time on NON-uniform scenes: a share of the Gaussians is concentrated in a small screen region
(long per-tile lists next to nearly empty tiles), the situation real captures produce.  Prints step time, stage times
and the list-length distribution."""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scgaussian_amd import synthetic as syn, rasterizer as R

dev = torch.device("cuda", 0)
W, H, P = 1008, 756, 200_000


def scene(frac, spread, seed=0):
    return syn.make_clustered_scene(P, W, H, frac, spread, seed=seed)


for frac, spread in ((0.0, 0.0), (0.3, 0.08), (0.6, 0.05), (0.9, 0.03)):
    sc = scene(frac, spread)
    cam = syn.default_camera(W, H)
    st = R.GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev),
                                         1.0, cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3,
                                         cam.camera_center.to(dev), False, False)
    rast = R.GaussianRasterizer(st)
    params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
    means, opac, shs, scales, rots = params
    ups = [u.to(dev) for u in syn.make_upstream_grads(W, H)]

    def step():
        for p in params:
            p.grad = None
        c, radii, d, a = rast(means3D=means, means2D=torch.zeros_like(means), opacities=opac, shs=shs, scales=scales,
                              rotations=rots)
        torch.autograd.backward([c, d, a], ups)

    for _ in range(5):
        step()
    timer = R.StageTimer()
    R.set_stage_timer(timer)
    for _ in range(20):
        step()
    stages = {k: round(v[0] * 1e3) for k, v in timer.summary().items()}
    R.set_stage_timer(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    fs = R.forward_stages(st, means.detach(), opac.detach(), shs=shs.detach(), scales=scales.detach(), rotations=rots.detach())
    rng = fs["ranges"].cpu().numpy().astype("int64")
    ln = rng[:, 1] - rng[:, 0]
    print(f"cluster {frac:.0%} (sigma {spread}): step {dt*1e3:.3f} ms  R {fs['num_rendered']}  list len mean {ln.mean():.0f} "
          f"p99 {int(sorted(ln)[int(len(ln)*0.99)])} max {ln.max()}  stages(us) {stages}")
