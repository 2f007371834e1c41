#!/usr/bin/env python3
"""Same-PROCESS A/B: one warm device, the variants alternate in blocks (A B A B ...), so clock state, box and allocator are
shared — differences of 1 % show.  A variant is a library build (tag of scgaussian_amd/libscg_raster_<tag>.so; '' = the
product build) and / or a module switch of scgaussian_amd.rasterizer.

    tools/ab_inproc.py [--workload S2|S3|S1|S4|clustered30|clustered60] [--mode train|render] [--libs ,probe1] \\
                       [--switch SKIP_IDLE_RARE_SORT=True,False] [--reps 4] [--steps 150] [--warm 600]
"""
import argparse
import math
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scgaussian_amd                                                         # noqa: E402
from scgaussian_amd import _lib, rasterizer as R, synthetic as syn           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="S2")
ap.add_argument("--mode", default="train", choices=["train", "render"])
ap.add_argument("--libs", default="")
ap.add_argument("--switch", default="")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--warm", type=int, default=600)
args = ap.parse_args()
scgaussian_amd.single_gpu_host_setup()
dev = torch.device("cuda", 0)

tags = args.libs.split(",") if args.libs else [""]
libs = {t: _lib.open_library(_lib.LIB_PATH if not t else _lib.LIB_PATH.replace(".so", f"_{t}.so")) for t in tags}
sw_name, sw_vals = None, [None]
if args.switch:
    sw_name, vals = args.switch.split("=")
    sw_vals = [eval(v) for v in vals.split(",")]                             # noqa: S307 - a developer tool
variants = [(t, v) for t in tags for v in sw_vals]

if args.workload in syn.CLUSTERED:
    w = syn.WORKLOADS["S2"]
    sc = syn.make_clustered_scene(w["P"], w["width"], w["height"], *syn.CLUSTERED[args.workload], seed=0).to(dev)
else:
    w = syn.WORKLOADS[args.workload]
    sc = syn.make_scene(w["P"], w["width"], w["height"], seed=0).to(dev)
P, W, H = w["P"], w["width"], w["height"]
params = [sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations]
for p in params:
    p.requires_grad_(True)
means, shs, opac, scales, rots = params
bg = torch.zeros(3, device=dev)
views = [syn.default_camera(W, H), syn.orbit_camera(W, H, 6.0, 0.0, 7.0), syn.orbit_camera(W, H, -6.0, 2.0, 7.0)]


def sett(cam):
    c = cam.to(dev)
    return R.GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                           bg, 1.0, c.world_view_transform, c.full_proj_transform, 3, c.camera_center, False,
                                           False)


setts = [sett(v) for v in views]
rasts = [R.GaussianRasterizer(s) for s in setts]
ups = [tuple(t.to(dev) for t in syn.make_upstream_grads(W, H, seed=10 + i)) for i in range(3)]


def step(i):
    if args.mode == "render":
        with torch.no_grad():
            out = R.forward_fused(setts[i % 3], means, opac, shs, None, scales, rots, None, False)
            if out is None:
                R.forward_stages(setts[i % 3], means, opac, shs=shs, scales=scales, rotations=rots)
        return
    for p in params:
        p.grad = None
    m2 = torch.zeros_like(means, requires_grad=True)
    c, radii, d, a = rasts[i % 3](means3D=means, means2D=m2, opacities=opac, shs=shs, scales=scales, rotations=rots)
    torch.autograd.backward([c, d, a], list(ups[i % 3]))


def select(variant):
    tag, val = variant
    _lib._lib = libs[tag]
    if sw_name:
        setattr(R, sw_name, val)


for v in variants:                       # every variant's kernels loaded, capacities known, device warm
    select(v)
    for i in range(30):
        step(i)
for i in range(args.warm):
    step(i)
torch.cuda.synchronize()
wall = {v: [] for v in variants}
stages = {v: {} for v in variants}
for rep in range(args.reps):
    for v in (variants if rep % 2 == 0 else variants[::-1]):
        select(v)
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        wall[v].append((time.perf_counter() - t0) / args.steps * 1e3)
        tm = R.StageTimer()
        R.set_stage_timer(tm)
        for i in range(30):
            step(i)
        for k, (ms, _) in tm.summary().items():
            stages[v].setdefault(k, []).append(ms)
        R.set_stage_timer(None)
print(f"{args.workload} {args.mode}: {args.reps} blocks of {args.steps} steps per variant, alternating, one process")
for v in variants:
    name = f"lib={v[0] or 'product'}" + (f" {sw_name}={v[1]}" if sw_name else "")
    ws = wall[v]
    print(f"  {name:40s} ms/step median {statistics.median(ws):.4f}  min {min(ws):.4f}  all {[round(x, 4) for x in ws]}")
    print("      stages (us, median over blocks):", {k: round(statistics.median(x) * 1e3, 1) for k, x in stages[v].items()})
