#!/usr/bin/env python3
"""Fit Gaussians to images of a synthetic ground-truth scene with the drop-in rasterizer — the reference's training
loop shape (train.py:108-208: render -> L1 (+ depth term) -> backward -> Adam), reduced to the hot path.

    python examples/fit_synthetic.py [--iters 300] [--gaussians 5000]

Prints loss / PSNR every 50 iterations.  Used by tests/test_gpu_train_loop.py as an end-to-end check that the
gradients the HIP kernels produce actually optimise a scene."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer   # noqa: E402
from scgaussian_amd import synthetic as syn                                                    # noqa: E402


def settings(cam, deg, bg, dev):
    camd = cam.to(dev)
    return GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2),
                                         math.tan(cam.FoVy / 2), bg, 1.0, camd.world_view_transform,
                                         camd.full_proj_transform, deg, camd.camera_center, False, False)


def render(params, st):
    xyz, f_dc, f_rest, opac_raw, scale_raw, rot_raw = params
    means2D = torch.zeros_like(xyz, requires_grad=True)
    shs = torch.cat([f_dc, f_rest], dim=1)
    return GaussianRasterizer(st)(means3D=xyz, means2D=means2D, opacities=torch.sigmoid(opac_raw), shs=shs,
                                  scales=torch.exp(scale_raw), rotations=torch.nn.functional.normalize(rot_raw))


def psnr(a, b):
    return float(-10.0 * torch.log10(((a - b) ** 2).mean()))


def fit(iters=300, P=5000, W=256, H=192, seed=0, verbose=True, dev="cuda"):
    dev = torch.device(dev)
    gt = syn.make_scene(P, W, H, seed=seed, log_scale_mean=-3.3).to(dev)
    cams = [syn.default_camera(W, H), syn.orbit_camera(W, H, 8.0, 0.0, 7.0), syn.orbit_camera(W, H, -8.0, 3.0, 7.0)]
    bg = torch.zeros(3, device=dev)
    sts = [settings(c, 3, bg, dev) for c in cams]
    with torch.no_grad():
        targets = [GaussianRasterizer(s)(means3D=gt.means3D, means2D=torch.zeros_like(gt.means3D), opacities=gt.opacities,
                                         shs=gt.shs, scales=gt.scales, rotations=gt.rotations) for s in sts]
    # start from a perturbed copy of the ground truth (the reference starts from matched ray depths)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    noise = lambda t, s: (t + s * torch.randn(t.shape, generator=g).to(dev))                     # noqa: E731
    params = [noise(gt.means3D, 0.03), noise(gt.shs[:, :1], 0.3), torch.zeros_like(gt.shs[:, 1:]),
              torch.logit(gt.opacities.clamp(0.02, 0.98)) + 0.5 * torch.randn(P, 1, generator=g).to(dev),
              torch.log(gt.scales) + 0.2 * torch.randn(P, 3, generator=g).to(dev), noise(gt.rotations, 0.1)]
    params = [p.detach().clone().requires_grad_(True) for p in params]
    opt = torch.optim.Adam([{"params": [params[0]], "lr": 2e-3}, {"params": [params[1]], "lr": 1e-2},
                            {"params": [params[2]], "lr": 5e-4}, {"params": [params[3]], "lr": 3e-2},
                            {"params": [params[4]], "lr": 5e-3}, {"params": [params[5]], "lr": 1e-3}], eps=1e-15)
    history = []
    for it in range(iters):
        v = it % len(sts)
        color, radii, depth, alpha = render(params, sts[v])
        t_color, _, t_depth, t_alpha = targets[v]
        loss = (color - t_color).abs().mean() + 0.05 * (depth - t_depth).abs().mean() + 0.1 * (alpha - t_alpha).abs().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if it % 50 == 0 or it == iters - 1:
            with torch.no_grad():
                p = sum(psnr(render(params, s)[0], t[0]) for s, t in zip(sts, targets)) / len(sts)
            history.append((it, float(loss.detach()), p))
            if verbose:
                print(f"iter {it:4d}  loss {float(loss):.5f}  PSNR {p:.2f} dB  visible {int((radii > 0).sum())}")
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--gaussians", type=int, default=5000)
    a = ap.parse_args()
    fit(a.iters, a.gaussians)
