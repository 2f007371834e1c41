"""Fused match loss from the rendered depth against the line-by-line CPU restatement of
scene/gaussian_model.py:241-282 (oracle/match_loss_oracle.py), value and gradient w.r.t. the depth image."""
import math

import numpy as np
import pytest
import torch

from oracle import match_loss_oracle as mlo
from scgaussian_amd import synthetic as syn


def _intr(cam):
    W, H = cam.image_width, cam.image_height
    fx, fy = W / (2 * math.tan(cam.FoVx / 2)), H / (2 * math.tan(cam.FoVy / 2))
    return torch.tensor([[fx, 0, W / 2.0], [0, fy, H / 2.0], [0, 0, 1]], dtype=torch.float32)


def _pair(cam0, cam1, depth0, M, seed, with_masks=True):
    """Matches of view 0 -> view 1 built from a depth map of view 0 (plus pixel noise on the view-1 side)."""
    g = torch.Generator().manual_seed(seed)
    W, H = cam0.image_width, cam0.image_height
    K0, K1 = _intr(cam0), _intr(cam1)
    w2c0, w2c1 = cam0.world_view_transform.t().contiguous(), cam1.world_view_transform.t().contiguous()
    c2w0 = torch.linalg.inv(w2c0)
    uv0 = torch.stack([torch.rand(M, generator=g) * (W + 6) - 3, torch.rand(M, generator=g) * (H + 6) - 3], 1)
    cam_rays = (torch.linalg.inv(K0) @ torch.cat([uv0, torch.ones(M, 1)], 1).t()).t()
    cam_rays = cam_rays / cam_rays.norm(dim=1, keepdim=True)
    rays_d = (c2w0[:3, :3] @ cam_rays.t()).t().contiguous()
    rays_o = c2w0[:3, 3][None].repeat(M, 1).contiguous()
    # "true" matches: sample the depth, lift, project into view 1, add noise
    with torch.no_grad():
        gx = (uv0[:, 0] / W) * 2 - 1
        gy = (uv0[:, 1] / H) * 2 - 1
        d = torch.nn.functional.grid_sample(depth0[None, None], torch.stack([gx, gy], -1)[None, None], align_corners=False).reshape(-1)
        z = d / cam_rays[:, 2]
        world = rays_o + rays_d * z[:, None]
        cam = (w2c1 @ torch.cat([world, torch.ones(M, 1)], 1).t())[:3]
        xyz = K1 @ cam
        uv1 = (xyz[:2] / (xyz[2:] + 1e-8)).t() + torch.randn(M, 2, generator=g) * 3.0
    mask0 = (torch.rand(M, generator=g) > 0.2).float() if with_masks else None
    mask1 = (torch.rand(M, generator=g) > 0.2).float() if with_masks else None
    return dict(uv0=uv0.contiguous(), rays_o=rays_o, rays_d=rays_d, cam_rays_d=cam_rays.contiguous(), mask0=mask0,
                mask1=mask1, intr1=K1, w2c1=w2c1, uv1=uv1.contiguous())


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 37, 1999, 5000])
def test_match_loss_matches_oracle(M):
    from scgaussian_amd.match_loss import match_loss_from_depth
    W, H = 252, 189
    cam0 = syn.orbit_camera(W, H, 0.0, 0.0, 7.0)
    cams = [syn.orbit_camera(W, H, 9.0, 2.0, 7.0), syn.orbit_camera(W, H, -7.0, -3.0, 7.5)]
    g = torch.Generator().manual_seed(M)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    depth = 6.0 + 1.5 * torch.sin(xx / 31.0) * torch.cos(yy / 23.0) + 0.05 * torch.randn(H, W, generator=g)
    pairs = [_pair(cam0, c, depth, M, 10 * M + k) for k, c in enumerate(cams)]

    d_cpu = depth.clone().requires_grad_(True)
    ones = torch.ones(M)
    ref = sum(mlo.match_loss_pair(d_cpu, p["uv0"], p["rays_o"], p["rays_d"], p["cam_rays_d"],
                                  p["mask0"] if p["mask0"] is not None else ones,
                                  p["mask1"] if p["mask1"] is not None else ones, p["intr1"], p["w2c1"], p["uv1"],
                                  float(W), float(H)) for p in pairs)
    (ref * 0.3).backward()                      # train.py:165 weights it by 0.3

    d_gpu = depth.clone().cuda()[None].requires_grad_(True)           # (1,H,W) like rendered_depth
    out = match_loss_from_depth(d_gpu, [{k: (v.cuda() if v is not None else None) for k, v in p.items()} for p in pairs],
                                float(W), float(H))
    (out * 0.3).backward()
    assert abs(float(out.detach()) - float(ref.detach())) <= 2e-5 * max(1.0, abs(float(ref.detach())))
    gref = d_cpu.grad.numpy()
    ggot = d_gpu.grad[0].cpu().numpy()
    assert np.abs(gref).max() > 0
    assert np.abs(ggot - gref).max() <= 1e-4 * np.abs(gref).max()
    assert (ggot != 0).sum() <= 4 * 2 * M


@pytest.mark.gpu
def test_match_loss_without_masks_and_no_grad():
    from scgaussian_amd.match_loss import match_loss_from_depth
    W, H = 100, 80
    cam0, cam1 = syn.orbit_camera(W, H, 0.0, 0.0, 7.0), syn.orbit_camera(W, H, 6.0, 0.0, 7.0)
    depth = torch.full((H, W), 7.0)
    p = _pair(cam0, cam1, depth, 300, 5, with_masks=False)
    ones = torch.ones(300)
    ref = mlo.match_loss_pair(depth, p["uv0"], p["rays_o"], p["rays_d"], p["cam_rays_d"], ones, ones, p["intr1"],
                              p["w2c1"], p["uv1"], float(W), float(H))
    with torch.no_grad():
        out = match_loss_from_depth(depth.cuda(), [{k: (v.cuda() if v is not None else None) for k, v in p.items()}],
                                    float(W), float(H))
    assert abs(float(out) - float(ref)) < 2e-5
