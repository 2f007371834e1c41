#!/usr/bin/env python3
"""Where the time of a training ITERATION goes at the reference's own scene size (10 k Gaussians @ 256x256: render() on the raw model
+ fused image loss + backward + Adam over the reference's twelve parameter groups): captured step vs eager step, fused vs
multi-tensor Adam.  Usage: python tools/captured_loop_timing.py   (profiles/r06r_captured_loop_timing.txt)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scgaussian_amd                                                                       # noqa: E402
from scgaussian_amd import losses, synthetic as syn                                         # noqa: E402
from scgaussian_amd.graph_step import CapturedStep                                          # noqa: E402
from scgaussian_amd.render import PipelineParams, render                                    # noqa: E402

scgaussian_amd.single_gpu_host_setup()
dev = torch.device("cuda")
P, W, H = 10000, 256, 256
gt = syn.make_scene(P, W, H, seed=0, log_scale_mean=-3.3)
model = syn.make_raw_model(gt).to(dev).requires_grad_()
cams = [c.to(dev) for c in (syn.default_camera(W, H), syn.orbit_camera(W, H, 8.0, 0.0, 7.0), syn.orbit_camera(W, H, -8.0, 3.0, 7.0))]
pipe, bg = PipelineParams(), torch.zeros(3, device=dev)
with torch.no_grad():
    targets = [render(c, model, pipe, bg)["render"].clone() for c in cams]
params = model.parameters()


def step_of(v):
    def fn():
        pkg = render(cams[v], model, pipe, bg)
        loss = losses.image_loss(pkg["render"], targets[v], 0.2)
        loss.backward()
        return loss
    return fn


fns = [step_of(v) for v in range(3)]
steps = [CapturedStep(f, params=params) for f in fns]


def timeit(name, body, n=600):
    for i in range(30):
        body(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        body(i)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms")


for fused in (True, False):
    opt = torch.optim.Adam([{"params": [p], "lr": 1e-4} for p in params], eps=1e-15, fused=fused)

    def eager(i):
        opt.zero_grad(set_to_none=True)
        fns[i % 3]()
        opt.step()

    def eager_no_opt(i):
        for p in params:
            p.grad = None
        fns[i % 3]()
    timeit("replay only", lambda i: steps[i % 3].replay())
    timeit(f"replay + Adam(fused={fused})", lambda i: (steps[i % 3].replay(), opt.step()))
    timeit(f"eager + Adam(fused={fused})", eager)
    timeit("eager step only", eager_no_opt)
    timeit(f"Adam(fused={fused}) only", lambda i: opt.step())
