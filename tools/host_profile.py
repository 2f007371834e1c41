"""Host time of a training step through the drop-in rasterizer on a scene small enough for the GPU to be idle:
forward vs backward, grad vs no-grad, and a cProfile of the calling thread.  Usage: python tools/host_profile.py [S1]"""
import cProfile
import math
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scgaussian_amd import rasterizer as R, synthetic as syn      # noqa: E402

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


DIRECT = os.environ.get("SCG_BACKWARD_DIRECT", "1") == "1"     # hand the fixed upstream gradients to the engine (bench.py)


def step():
    for p in params:
        p.grad = None
    m2 = torch.zeros_like(means, requires_grad=True)
    c, radii, d, a = rast(means3D=means, means2D=m2, opacities=opac, shs=shs, scales=scales, rotations=rots)
    if DIRECT:
        torch.autograd.Variable._execution_engine.run_backward((c, d, a), tuple(ups), False, False, (),
                                                               allow_unreachable=True, accumulate_grad=True)
    else:
        torch.autograd.backward([c, d, a], ups)


if os.environ.get("SCG_AUTOGRAD_SINGLE_THREAD") == "1":
    # the autograd engine runs CUDA backward nodes on a per-device worker thread: every backward() pays two thread
    # hand-offs; with one GPU and one Python thread (the reference's setup) the calling thread can run them itself
    torch.autograd.set_multithreading_enabled(False)
    print("autograd multithreading disabled")
import gc                                                   # noqa: E402
gc.collect()
gc.disable()                    # as bench.py's legs: a 0.12 ms step creates no cycles worth a collector pause
for _ in range(50):
    step()
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("train step (%s): host %.1f us, wall %.1f us" % ("run_backward direct" if DIRECT else "torch.autograd.backward", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
t0 = time.perf_counter()
with torch.no_grad():
    for _ in range(N):
        rast(means3D=means, means2D=means, opacities=opac, shs=shs, scales=scales, rotations=rots)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("no_grad forward: host %.1f us per call" % ((t1 - t0) / N * 1e6))
t0 = time.perf_counter()
for _ in range(N):
    m2 = torch.zeros_like(means, requires_grad=True)
    rast(means3D=means, means2D=m2, opacities=opac, shs=shs, scales=scales, rotations=rots)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("grad-mode forward: host %.1f us per call" % ((t1 - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    step()
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr)
ps.sort_stats("tottime").print_stats(24)
