#!/usr/bin/env python3
"""Training-trajectory equivalence: a proxy for the north star's "LLFF-fern 3-view PSNR matching the reference to
+-0.05 dB" on a box that has neither the dataset nor the reference's CUDA rasterizer.

The SAME seeded 3-view scene is trained twice with the reference's loss (train.py:143-165):

    render -> 0.8 L1 + 0.2 (1 - SSIM) + 0.3 match loss on the rendered depth -> backward -> Adam

  run A  CPU: the oracle rasterizer (oracle/torch_rasterizer.py, autograd gradients), the reference's SSIM formulation
         in plain torch (five grouped 11x11 conv2d, utils/loss_utils.py:56-94), the line-by-line match-loss restatement
  run B  MI355X: the HIP rasterizer, the fused L1+SSIM kernel, the fused match-loss kernel — the product path

Same initial parameters (raw: log-scales, logit opacities, un-normalised quaternions, as scene/gaussian_model.py keeps
them), same targets, same Adam hyper-parameters.  Reported: max |loss_A - loss_B| over the iterations, final PSNR of
both runs on the training views and on a held-out view.  Checker use of oracle/: this is test infrastructure
(tests/test_gpu_quality_proxy.py runs a short version; the long form is run by hand and kept under profiles/).

    python tools/quality_proxy.py [--iters 500] [--gaussians 5000] [--width 504 --height 378] [--json out.json]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scgaussian_amd import synthetic as syn          # noqa: E402

LAMBDA_DSSIM = 0.2       # arguments/__init__.py: lambda_dssim
MATCH_WEIGHT = 0.3       # train.py:165


def _window(dev):
    g = torch.tensor([math.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)], dtype=torch.float32)
    g = g / g.sum()
    return (g[:, None] @ g[None, :]).expand(3, 1, 11, 11).contiguous().to(dev)


def torch_image_loss(x, y, win):
    """utils/loss_utils.py l1_loss + ssim (window 11, sigma 1.5, zero padding 5, C1 = 0.01^2, C2 = 0.03^2)."""
    conv = lambda t: F.conv2d(t[None], win, padding=5, groups=3)[0]      # noqa: E731
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    smap = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return (1 - LAMBDA_DSSIM) * (x - y).abs().mean() + LAMBDA_DSSIM * (1 - smap.mean())


def psnr(a, b):
    return float(-10.0 * torch.log10(((a - b) ** 2).mean()))


def _intr(cam):
    W, H = cam.image_width, cam.image_height
    fx, fy = W / (2 * math.tan(cam.FoVx / 2)), H / (2 * math.tan(cam.FoVy / 2))
    return torch.tensor([[fx, 0, W / 2.0], [0, fy, H / 2.0], [0, 0, 1]], dtype=torch.float32)


def make_problem(P, W, H, seed, n_match):
    """Ground-truth scene, 3 training views + 1 held-out view, perturbed initial parameters (raw form), and match data
    between view 0 and views 1, 2 derived from the ground-truth geometry."""
    gt = syn.make_scene(P, W, H, seed=seed, log_scale_mean=-3.2)
    cams = [syn.default_camera(W, H), syn.orbit_camera(W, H, 8.0, 0.0, 7.0), syn.orbit_camera(W, H, -8.0, 3.0, 7.0)]
    held_out = syn.orbit_camera(W, H, 3.0, -2.0, 7.0)
    g = torch.Generator().manual_seed(seed + 1)
    init = dict(
        xyz=gt.means3D + 0.03 * torch.randn(P, 3, generator=g),
        f_dc=gt.shs[:, :1] + 0.3 * torch.randn(P, 1, 3, generator=g),
        f_rest=torch.zeros(P, 15, 3),
        opacity=torch.logit(gt.opacities.clamp(0.02, 0.98)) + 0.5 * torch.randn(P, 1, generator=g),
        scaling=torch.log(gt.scales) + 0.2 * torch.randn(P, 3, generator=g),
        rotation=gt.rotations + 0.1 * torch.randn(P, 4, generator=g))
    return gt, cams, held_out, init, g


def match_pairs(cams, depth0, n_match, g):
    """uv / rays of view 0 and their (noisy) matches in views 1 and 2, from a depth map of view 0."""
    cam0 = cams[0]
    W, H = cam0.image_width, cam0.image_height
    K0 = _intr(cam0)
    c2w0 = torch.linalg.inv(cam0.world_view_transform.t())
    pairs = []
    for cam1 in cams[1:]:
        uv0 = torch.stack([torch.rand(n_match, generator=g) * W, torch.rand(n_match, generator=g) * H], 1)
        cr = (torch.linalg.inv(K0) @ torch.cat([uv0, torch.ones(n_match, 1)], 1).t()).t()
        cr = cr / cr.norm(dim=1, keepdim=True)
        rd = (c2w0[:3, :3] @ cr.t()).t().contiguous()
        ro = c2w0[:3, 3][None].repeat(n_match, 1).contiguous()
        grid = torch.stack([uv0[:, 0] / W * 2 - 1, uv0[:, 1] / H * 2 - 1], -1)[None, None]
        d = F.grid_sample(depth0[None, None], grid, align_corners=False).reshape(-1).clamp_min(0.5)
        w2c1 = cam1.world_view_transform.t().contiguous()
        world = ro + rd * (d / cr[:, 2])[:, None]
        xyz = _intr(cam1) @ (w2c1 @ torch.cat([world, torch.ones(n_match, 1)], 1).t())[:3]
        uv1 = (xyz[:2] / (xyz[2:] + 1e-8)).t().contiguous() + 0.5 * torch.randn(n_match, 2, generator=g)
        pairs.append(dict(uv0=uv0.contiguous(), rays_o=ro, rays_d=rd, cam_rays_d=cr.contiguous(),
                          mask0=torch.ones(n_match), mask1=torch.ones(n_match), intr1=_intr(cam1), w2c1=w2c1, uv1=uv1))
    return pairs


class Backend:
    """What differs between the two runs: the rasterizer, the image loss, the match loss, the device."""

    def __init__(self, kind, W, H):
        self.kind = kind
        self.dev = torch.device("cuda") if kind == "hip" else torch.device("cpu")
        self.W, self.H = W, H
        if kind == "hip":
            from scgaussian_amd import losses
            from scgaussian_amd.match_loss import match_loss_from_depth
            from scgaussian_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
            self.Settings, self.Rasterizer = GaussianRasterizationSettings, GaussianRasterizer
            self.image_loss = lambda x, y: losses.image_loss(x, y, LAMBDA_DSSIM)
            self.match_loss = lambda d, pairs: match_loss_from_depth(d, pairs, float(W), float(H))
        else:
            from oracle import match_loss_oracle as mlo            # checker side
            from oracle import torch_rasterizer as orc
            win = _window(self.dev)
            self.Settings = orc.Settings

            class OracleRasterizer:
                def __init__(self, s):
                    self.s = s

                def __call__(self, means3D, means2D, opacities, shs, scales, rotations):
                    return orc.rasterize(means3D, means2D, opacities, self.s, shs=shs, scales=scales, rotations=rotations)
            self.Rasterizer = OracleRasterizer
            self.image_loss = lambda x, y: torch_image_loss(x, y, win)
            self.match_loss = lambda d, pairs: sum(
                mlo.match_loss_pair(d[0], p["uv0"], p["rays_o"], p["rays_d"], p["cam_rays_d"], p["mask0"], p["mask1"],
                                    p["intr1"], p["w2c1"], p["uv1"], float(W), float(H)) for p in pairs)

    def settings(self, cam, deg, bg):
        c = cam.to(self.dev)
        return self.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                             bg.to(self.dev), 1.0, c.world_view_transform, c.full_proj_transform, deg, c.camera_center,
                             False, False)

    def render(self, params, st):
        xyz, f_dc, f_rest, opac, scaling, rot = params
        means2D = torch.zeros_like(xyz, requires_grad=True)
        return self.Rasterizer(st)(means3D=xyz, means2D=means2D, opacities=torch.sigmoid(opac),
                                   shs=torch.cat([f_dc, f_rest], dim=1), scales=torch.exp(scaling),
                                   rotations=F.normalize(rot))


def train(backend, init, cams, held_out, targets, held_target, pairs, iters, deg=3, log_every=0):
    dev = backend.dev
    bg = torch.zeros(3)
    sts = [backend.settings(c, deg, bg) for c in cams]
    st_held = backend.settings(held_out, deg, bg)
    params = [init[k].detach().clone().to(dev).requires_grad_(True)
              for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]
    # learning rates of arguments/__init__.py (position 1.6e-4 x spatial scale ~ 10, features 2.5e-3 and /20, opacity
    # 5e-2, scaling 5e-3, rotation 1e-3)
    opt = torch.optim.Adam([{"params": [params[0]], "lr": 1.6e-3}, {"params": [params[1]], "lr": 2.5e-3},
                            {"params": [params[2]], "lr": 2.5e-3 / 20}, {"params": [params[3]], "lr": 5e-2},
                            {"params": [params[4]], "lr": 5e-3}, {"params": [params[5]], "lr": 1e-3}], eps=1e-15)
    tg = [t.to(dev) for t in targets]
    pr = [{k: v.to(dev) for k, v in p.items()} for p in pairs]
    losses = []
    t0 = time.perf_counter()
    for it in range(iters):
        v = it % len(cams)                                 # fixed view order: both runs see the same sequence
        c, radii, d, a = backend.render(params, sts[v])
        loss = backend.image_loss(c, tg[v])
        if v == 0:
            loss = loss + MATCH_WEIGHT * backend.match_loss(d, pr)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if log_every and (it + 1) % log_every == 0:
            print(f"  [{backend.kind}] it {it + 1:5d} loss {losses[-1]:.6f}  ({time.perf_counter() - t0:.1f} s)", flush=True)
    with torch.no_grad():
        train_psnr = [psnr(backend.render(params, s)[0], t) for s, t in zip(sts, tg)]
        held_psnr = psnr(backend.render(params, st_held)[0], held_target.to(dev))
    return dict(losses=losses, train_psnr=train_psnr, held_out_psnr=held_psnr, seconds=time.perf_counter() - t0,
                params=[p.detach().cpu() for p in params])


def run(iters=500, P=5000, W=504, H=378, seed=0, n_match=2000, log_every=0):
    gt, cams, held_out, init, g = make_problem(P, W, H, seed, n_match)
    hip = Backend("hip", W, H)
    bg = torch.zeros(3)
    with torch.no_grad():            # targets: the ground-truth scene rendered once (shared by both runs)
        gtd = gt.to(hip.dev)

        def gt_render(cam):
            return hip.Rasterizer(hip.settings(cam, 3, bg))(means3D=gtd.means3D, means2D=torch.zeros_like(gtd.means3D),
                                                            opacities=gtd.opacities, shs=gtd.shs, scales=gtd.scales,
                                                            rotations=gtd.rotations)
        outs = [gt_render(c) for c in cams]
        targets = [o[0].cpu() for o in outs]
        depth0 = (outs[0][2][0] / outs[0][3][0].clamp_min(1e-3)).cpu()
        held_target = gt_render(held_out)[0].cpu()
    pairs = match_pairs(cams, depth0, n_match, g)
    res_hip = train(hip, init, cams, held_out, targets, held_target, pairs, iters, log_every=log_every)
    # the HIP run once more: its gradients are accumulated with float atomics (as the reference's CUDA rasterizer's are), so
    # two runs of the SAME implementation already follow slightly different trajectories — the noise floor of the comparison
    res_hip2 = train(hip, init, cams, held_out, targets, held_target, pairs, iters)
    res_cpu = train(Backend("oracle", W, H), init, cams, held_out, targets, held_target, pairs, iters, log_every=log_every)
    la, lb = np.array(res_cpu["losses"]), np.array(res_hip["losses"])
    mean = lambda v: float(sum(v) / len(v))                  # noqa: E731
    out = {
        "config": dict(iters=iters, gaussians=P, width=W, height=H, views=3, matches_per_pair=n_match, seed=seed,
                       loss="0.8 L1 + 0.2 (1-SSIM) + 0.3 match loss (view 0)"),
        "loss_first": [float(la[0]), float(lb[0])], "loss_last": [float(la[-1]), float(lb[-1])],
        "max_abs_loss_diff": float(np.abs(la - lb).max()), "max_rel_loss_diff": float((np.abs(la - lb) / np.abs(la)).max()),
        "train_psnr_oracle": res_cpu["train_psnr"], "train_psnr_hip": res_hip["train_psnr"],
        "held_out_psnr_oracle": res_cpu["held_out_psnr"], "held_out_psnr_hip": res_hip["held_out_psnr"],
        "mean_train_psnr_diff_db": abs(mean(res_cpu["train_psnr"]) - mean(res_hip["train_psnr"])),
        "held_out_psnr_diff_db": abs(res_cpu["held_out_psnr"] - res_hip["held_out_psnr"]),
        "max_abs_psnr_diff_db": float(max(abs(a - b) for a, b in zip(res_cpu["train_psnr"] + [res_cpu["held_out_psnr"]],
                                                                      res_hip["train_psnr"] + [res_hip["held_out_psnr"]]))),
        "hip_vs_hip_max_abs_psnr_diff_db": float(max(abs(a - b) for a, b in zip(res_hip2["train_psnr"] + [res_hip2["held_out_psnr"]],
                                                                                 res_hip["train_psnr"] + [res_hip["held_out_psnr"]]))),
        "hip_vs_hip_max_abs_loss_diff": float(np.abs(np.array(res_hip2["losses"]) - lb).max()),
        "max_param_diff": [float((a - b).abs().max()) for a, b in zip(res_cpu["params"], res_hip["params"])],
        "seconds_oracle": round(res_cpu["seconds"], 1), "seconds_hip": round(res_hip["seconds"], 2),
    }
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--gaussians", type=int, default=5000)
    ap.add_argument("--width", type=int, default=504)
    ap.add_argument("--height", type=int, default=378)
    ap.add_argument("--matches", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    res = run(args.iters, args.gaussians, args.width, args.height, args.seed, args.matches, log_every=50)
    print(json.dumps(res, indent=1))
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(res, fh, indent=1)
    # the metric the reference reports is the MEAN PSNR over views; single views at ~40 dB move by several hundredths of
    # a dB between two runs of one implementation (hip_vs_hip_*: float-atomic order), which bounds what any comparison of
    # two fp32 trajectories can resolve
    ok = res["mean_train_psnr_diff_db"] <= 0.05 and res["held_out_psnr_diff_db"] <= 0.05 and res["max_abs_loss_diff"] <= 1e-3
    print("PASS" if ok else "FAIL", "(mean training-view PSNR and held-out PSNR within 0.05 dB, per-iteration loss within 1e-3; "
          f"single-view maximum {res['max_abs_psnr_diff_db']:.3f} dB vs {res['hip_vs_hip_max_abs_psnr_diff_db']:.3f} dB between two HIP runs)")
    sys.exit(0 if ok else 1)
