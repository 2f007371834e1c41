#!/usr/bin/env python3
"""Soak: many training steps in one process on scenes whose Gaussian count changes (as densification / pruning does),
alternating views and image sizes — watches for errors, non-finite outputs and growth of device or host memory.
Usage: tools/soak.py [steps]"""
import math
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scgaussian_amd import synthetic as syn, rasterizer as R

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda", 0)
sizes = [(320, 240), (504, 378), (256, 256)]
scenes = {}
t0 = time.time()
peak0 = None
steady, steady_n, ts = 0.0, 0, 0.0
for it in range(steps):
    W, H = sizes[(it // 50) % len(sizes)]
    P = 8000 + 1500 * ((it // 200) % 9)                    # the count moves like densify / prune
    key = (P, W, H)
    if key not in scenes:
        sc = syn.make_scene(P, W, H, seed=P % 97).to(dev)
        params = [t.clone().requires_grad_(True) for t in (sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations)]
        cams = [syn.orbit_camera(W, H, yaw, 2.0, 7.0) for yaw in (-8.0, 0.0, 9.0)]
        sts = [R.GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), torch.zeros(3, device=dev),
                                               1.0, c.world_view_transform.to(dev), c.full_proj_transform.to(dev), 3,
                                               c.camera_center.to(dev), False, False) for c in cams]
        if any(k[0] != P for k in scenes):                 # a new Gaussian count: drop the old scenes' buffers
            scenes = {}
        scenes[key] = (params, [R.GaussianRasterizer(s) for s in sts])
    params, rasts = scenes[key]
    if it % 50 == 10:                                       # steady state inside a block of 50 steps on one size
        torch.cuda.synchronize()
        ts = time.perf_counter()
    if it % 50 == 49:
        torch.cuda.synchronize()
        steady += time.perf_counter() - ts
        steady_n += 39
    means, shs, opac, scales, rots = params
    for p in params:
        p.grad = None
    c, radii, d, a = rasts[it % 3](means3D=means, means2D=torch.zeros_like(means), shs=shs, opacities=opac, scales=scales,
                                   rotations=rots)
    (c.mean() + 0.1 * d.mean() + 0.1 * a.mean()).backward()
    if it % 1000 == 999:
        torch.cuda.synchronize()
        ok = all(bool(torch.isfinite(p.grad).all()) for p in params) and bool(torch.isfinite(c).all())
        mem = torch.cuda.memory_allocated() / 2**20
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
        if peak0 is None:
            peak0 = (mem, rss)
        print(f"step {it + 1}: finite={ok} device {mem:.0f} MiB  host maxrss {rss:.0f} MiB  "
              f"{(time.time() - t0) / (it + 1) * 1e3:.3f} ms/step incl. scene builds, {steady / max(steady_n, 1) * 1e3:.3f} ms/step steady",
              flush=True)
        assert ok
torch.cuda.synchronize()
mem = torch.cuda.memory_allocated() / 2**20
rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
print(f"done: {steps} steps; device {peak0[0]:.0f} -> {mem:.0f} MiB, host maxrss {peak0[1]:.0f} -> {rss:.0f} MiB")
