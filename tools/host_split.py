#!/usr/bin/env python3
"""Where the HOST time of a small training step goes: per library call (scg_forward, scg_wait_num_rendered, scg_backward — the
C side: kernel launches, event record / wait) and per binding function (forward_fused, backward_fused), next to the whole step.
Usage: python tools/host_split.py [S1]      (run with SCG_AUTOGRAD_SINGLE_THREAD=1 for the bench's host setup)"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scgaussian_amd import synthetic as syn, rasterizer as R, _lib      # noqa: E402

if os.environ.get("SCG_AUTOGRAD_SINGLE_THREAD", "1") == "1":
    torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda", 0)
wl = syn.WORKLOADS["S1"] if len(sys.argv) > 1 and sys.argv[1] == "S1" else dict(P=2000, width=128, height=96)
sc = syn.make_scene(wl["P"], wl["width"], wl["height"])
cam = syn.default_camera(wl["width"], wl["height"])
st = R.GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                     torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                     cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
rast = R.GaussianRasterizer(st)
params = [t.to(dev).requires_grad_(True) for t in (sc.means3D, sc.opacities, sc.shs, sc.scales, sc.rotations)]
means, opac, shs, scales, rots = params
ups = [u.to(dev) for u in syn.make_upstream_grads(cam.image_width, cam.image_height)]
lib = _lib.load()
acc = {}


class Wrap:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        f = getattr(self._lib, name)

        def g(*a):
            t0 = time.perf_counter()
            r = f(*a)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        return g


w = Wrap(lib)
_lib.load = lambda: w
fn = {}
for name in ("forward_fused", "backward_fused", "_grad_outputs", "_frame_for"):
    orig = getattr(R, name)

    def timed(*a, _o=orig, _n=name, **k):
        t0 = time.perf_counter()
        r = _o(*a, **k)
        fn[_n] = fn.get(_n, 0.0) + time.perf_counter() - t0
        return r
    setattr(R, name, timed)


def step():
    for p in params:
        p.grad = None
    m2 = torch.zeros_like(means, requires_grad=True)
    c, radii, d, a = rast(means3D=means, means2D=m2, opacities=opac, shs=shs, scales=scales, rotations=rots)
    torch.autograd.Variable._execution_engine.run_backward((c, d, a), tuple(ups), False, False, (), allow_unreachable=True,
                                                           accumulate_grad=True)       # as bench.py hands them over


for _ in range(50):
    step()
torch.cuda.synchronize()
acc.clear()
fn.clear()
N = 1000
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("per step us: host %.1f  wall %.1f" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
for k, v in sorted(fn.items(), key=lambda kv: -kv[1]):
    print("  binding %-24s %.1f us (inclusive)" % (k, v / N * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  library %-24s %.1f us" % (k, v / N * 1e6))
