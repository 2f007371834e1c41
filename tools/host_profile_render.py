"""Host time of a training step THROUGH render() on the reference's raw model (the model path) at the reference's own scene size:
cProfile of the calling thread, next to the bare operator's step.  Usage: python tools/host_profile_render.py [S1]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scgaussian_amd                                               # noqa: E402
from scgaussian_amd import render as rmod, synthetic as syn         # noqa: E402

scgaussian_amd.single_gpu_host_setup()
dev = torch.device("cuda", 0)
wl = syn.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "S1"]
W, H = wl["width"], wl["height"]
sc = syn.make_scene(wl["P"], W, H)
model = syn.make_raw_model(sc).to(dev).requires_grad_()
model.active_sh_degree = int(os.environ.get("DEG", "1"))
cam = syn.default_camera(W, H).to(dev)
pipe = rmod.PipelineParams()
bg = torch.zeros(3, device=dev)
ups = [u.to(dev) for u in syn.make_upstream_grads(W, H)]
params = model.parameters()


def step():
    for p in params:
        p.grad = None
    o = rmod.render(cam, model, pipe, bg)
    torch.autograd.backward([o["render"], o["rendered_depth"], o["rendered_alpha"]], ups)


import gc                                                            # noqa: E402
gc.collect()
gc.disable()
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
print(f"render() on the raw model, {wl['P']} Gaussians {W}x{H}: enqueue {1e3 * (t1 - t0) / N:.4f} ms per step, with the device {1e3 * (t2 - t0) / N:.4f}")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
