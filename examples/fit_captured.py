#!/usr/bin/env python3
"""The reference's training loop at the reference's own scene size, as this repository would run it: the MODEL in the reference's
raw parameterisation (ray-bound + background sets, logits, log-scales, un-normalised quaternions, features_dc + features_rest:
scene/gaussian_model.py:452-509) rendered by `scgaussian_amd.render.render` (the model path: no getters, raw-parameter gradients in
one arena), loss 0.8 L1 + 0.2 (1 - SSIM) (train.py:160-161, the fused kernels), and one CAPTURED step per training view
(`graph_step.CapturedStep`: forward + loss + backward in a hipGraph; the optimizer steps outside it, in place).

    python examples/fit_captured.py [--iters 600] [--gaussians 10000] [--eager | --optimizer-in-graph]

Prints loss / PSNR every 100 iterations and the time per iteration.  tests/test_gpu_graph_step.py runs a short form of both modes
and holds their trajectories against each other."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scgaussian_amd                                                                       # noqa: E402
from scgaussian_amd import losses, synthetic as syn                                         # noqa: E402
from scgaussian_amd.graph_step import CapturedStep                                          # noqa: E402
from scgaussian_amd.render import PipelineParams, render                                    # noqa: E402


def psnr(a, b):
    return float(-10.0 * torch.log10(((a - b) ** 2).mean()))


def fit(iters=600, P=10_000, W=256, H=256, seed=0, captured=True, verbose=True, dev="cuda", optimizer_in_graph=False):
    scgaussian_amd.single_gpu_host_setup()
    dev = torch.device(dev)
    gt = syn.make_scene(P, W, H, seed=seed, log_scale_mean=-3.3)
    truth = syn.make_raw_model(gt).to(dev)
    cams = [c.to(dev) for c in (syn.default_camera(W, H), syn.orbit_camera(W, H, 8.0, 0.0, 7.0), syn.orbit_camera(W, H, -8.0, 3.0, 7.0))]
    pipe, bg = PipelineParams(), torch.zeros(3, device=dev)
    with torch.no_grad():
        targets = [render(c, truth, pipe, bg)["render"].clone() for c in cams]
    # start from a perturbed copy of the ground truth: depths along the rays off by a few per cent (what the reference optimises:
    # zval), colours and opacities disturbed
    g = torch.Generator().manual_seed(seed + 1)
    model = syn.make_raw_model(gt).to(dev)
    with torch.no_grad():
        model.zval.mul_(1.0 + 0.03 * torch.randn(model.zval.shape, generator=g).to(dev))
        model.bg_xyz.add_(0.03 * torch.randn(model.bg_xyz.shape, generator=g).to(dev))
        model.features_dc.add_(0.3 * torch.randn(model.features_dc.shape, generator=g).to(dev))
        model.opacity.add_(0.5 * torch.randn(model.opacity.shape, generator=g).to(dev))
    model.requires_grad_()
    model.active_sh_degree = 3
    params = model.parameters()
    lr = {1: 2e-3, 3: 1e-2, 4: 5e-4}                              # by tensor rank / role below
    groups = []
    for p in params:
        if p.dim() == 3:
            groups.append({"params": [p], "lr": 1e-2 if p.shape[1] == 1 else 5e-4})
        elif p.shape[1] == 1:
            groups.append({"params": [p], "lr": 3e-2 if p is not model.zval else 2e-3})
        elif p.shape[1] == 4:
            groups.append({"params": [p], "lr": 1e-3})
        else:
            groups.append({"params": [p], "lr": 2e-3 if p is model.bg_xyz else 5e-3})
    try:
        opt = torch.optim.Adam(groups, eps=1e-15, fused=True, capturable=optimizer_in_graph)
    except (RuntimeError, TypeError):
        opt = torch.optim.Adam(groups, eps=1e-15, capturable=optimizer_in_graph)
    if optimizer_in_graph:
        # The optimizer steps INSIDE the captured graph (torch's capturable Adam: its step counters live on the device).  Its state
        # must exist before the capture (created inside it, the state would be re-zeroed by every replay): one step on zero
        # gradients creates it without moving a parameter (0 / (0 + eps) = 0), then the step counters go back to zero.
        for p in params:
            p.grad = torch.zeros_like(p)
        opt.step()
        for st in opt.state.values():
            st["step"].zero_()
        for p in params:
            p.grad = None

    def step_fn(v):
        def fn():
            pkg = render(cams[v], model, pipe, bg)
            loss = losses.image_loss(pkg["render"], targets[v], 0.2)
            loss.backward()
            return loss, pkg["radii"]
        return fn
    fns = [step_fn(v) for v in range(len(cams))]
    if captured and optimizer_in_graph:
        def with_opt(f):
            def fn():
                out = f()
                opt.step()
                return out
            return fn
        for f in fns:                                            # the warm-up (capacities, camera keys, hints) WITHOUT the optimizer
            for _ in range(3):
                for p in params:
                    p.grad = None
                f()
        steps = [CapturedStep(with_opt(f), params=params, warmup=0) for f in fns]
    else:
        steps = [CapturedStep(f, params=params) for f in fns] if captured else None
    history = []
    t0, timed_from = None, min(30, iters - 1)                      # (the first launch of a graph uploads it: not a step's time)
    for it in range(iters):
        if it == timed_from:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        v = it % len(cams)
        if captured:
            loss, radii = steps[v].replay()                      # gradients land in the (static) .grad tensors of the capture
        else:
            opt.zero_grad(set_to_none=True)
            loss, radii = fns[v]()
        if not (captured and optimizer_in_graph):
            opt.step()
        if it % 100 == 0 or it == iters - 1:
            with torch.no_grad():
                p = sum(psnr(render(c, model, pipe, bg)["render"], t) for c, t in zip(cams, targets)) / len(cams)
            history.append((it, float(loss.detach()), p))
            if verbose:
                print(f"iter {it:4d}  loss {float(loss):.5f}  PSNR {p:.2f} dB  visible {int((radii > 0).sum())}")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / max(iters - timed_from, 1)
    if verbose:
        extra = "" if not captured else f"  ({sum(s.overflows for s in steps)} overflows, {sum(s.recaptures for s in steps)} recaptures)"
        mode = ("captured, optimizer inside the graph" if optimizer_in_graph else "captured") if captured else "eager"
        print(f"{mode}: {dt * 1e3:.4f} ms per iteration incl. the optimizer and the PSNR probes{extra}")
    if captured:
        for s in steps:
            s.close()
    return history, dt


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=600)
    ap.add_argument("--gaussians", type=int, default=10_000)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--optimizer-in-graph", action="store_true", help="capturable Adam stepping inside the captured graph")
    a = ap.parse_args()
    fit(a.iters, a.gaussians, captured=not a.eager, optimizer_in_graph=a.optimizer_in_graph)
