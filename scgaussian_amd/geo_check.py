"""Cross-view depth consistency check (reference utils/geo_check.py:25-128, numpy + cv2).

`reproject_with_depth` projects every pixel of a reference depth map into a source view, samples the source depth
there (bilinear, zeros outside the image — cv2.remap INTER_LINEAR / BORDER_CONSTANT), lifts the sample back to 3-D and
re-projects it into the reference view; `geocheck` keeps a pixel when, for more than `view_thresh` of its `num_src`
nearest source views, the round trip lands within `dist_thresh` pixels and `depth_thresh` relative depth, and averages
the consistent depths (utils/geo_check.py:33-88).

This is the same computation as torch tensor code on whatever device the depth maps live on (the rendered depth is
already in HBM: no host round trip, no cv2).  Differences from the reference, by construction: cv2.remap interpolates
with 5-bit fixed-point weights, this uses exact fp weights; pixels whose projection is not finite sample 0.
The function is dead code in the reference (never imported) — it is provided for BASELINE config 3.
"""
from __future__ import annotations

from typing import Tuple

import torch


def get_pairs(c2ws: torch.Tensor, num_select: int = 10) -> torch.Tensor:
    """Indices of the `num_select` nearest other cameras per camera (utils/geo_check.py:25-31)."""
    pos = c2ws[:, :3, 3]
    dists = torch.linalg.norm(pos[:, None] - pos[None], dim=-1).clone()
    dists.fill_diagonal_(1e3)
    return torch.argsort(dists, dim=1, stable=True)[:, :num_select]


def _bilinear_zeros(img: torch.Tensor, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """img (H,W) sampled at pixel coordinates (x,y) (H,W): bilinear, zero outside, integer coords = pixel centres."""
    H, W = img.shape
    ok = torch.isfinite(x) & torch.isfinite(y)
    x = torch.where(ok, x, torch.full_like(x, -2.0))
    y = torch.where(ok, y, torch.full_like(y, -2.0))
    x0, y0 = torch.floor(x), torch.floor(y)
    fx, fy = x - x0, y - y0
    x0, y0 = x0.long(), y0.long()

    def tap(yy, xx):
        inside = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = img[yy.clamp(0, H - 1), xx.clamp(0, W - 1)]
        return torch.where(inside, v, torch.zeros_like(v))

    top = tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx
    bot = tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx
    return top * (1 - fy) + bot * fy


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src
                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """utils/geo_check.py:91-128.  All inputs tensors on one device; matrices (3,3) / (4,4)."""
    H, W = depth_ref.shape
    dt = intrinsics_ref.dtype
    dev = depth_ref.device
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=dt), torch.arange(W, device=dev, dtype=dt), indexing="ij")
    x_ref, y_ref = xs.reshape(-1), ys.reshape(-1)
    ones = torch.ones_like(x_ref)
    # reference pixels -> reference 3-D -> source 3-D -> source pixels
    xyz_ref = torch.linalg.inv(intrinsics_ref) @ (torch.stack((x_ref, y_ref, ones)) * depth_ref.reshape(-1).to(dt))
    xyz_src = ((extrinsics_src @ torch.linalg.inv(extrinsics_ref)) @ torch.cat((xyz_ref, ones[None])))[:3]
    k_src = intrinsics_src @ xyz_src
    xy_src = k_src[:2] / k_src[2:3]
    x_src = xy_src[0].reshape(H, W).float()
    y_src = xy_src[1].reshape(H, W).float()
    sampled = _bilinear_zeros(depth_src.float(), x_src, y_src)
    # source pixels + sampled source depth -> source 3-D -> reference 3-D -> reference pixels
    xyz_src2 = torch.linalg.inv(intrinsics_src) @ (torch.cat((xy_src, ones[None])) * sampled.reshape(-1).to(dt))
    xyz_rep = ((extrinsics_ref @ torch.linalg.inv(extrinsics_src)) @ torch.cat((xyz_src2, ones[None])))[:3]
    depth_rep = xyz_rep[2].reshape(H, W).float()
    k_rep = intrinsics_ref @ xyz_rep
    xy_rep = k_rep[:2] / k_rep[2:3]
    return depth_rep, xy_rep[0].reshape(H, W).float(), xy_rep[1].reshape(H, W).float(), x_src, y_src


def _bilinear_zeros_batch(imgs: torch.Tensor, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """imgs (J,H,W) sampled at pixel coordinates (x,y) (J,H,W), image j at its own coordinates: `_bilinear_zeros` for a stack."""
    J, H, W = imgs.shape
    ok = torch.isfinite(x) & torch.isfinite(y)
    x = torch.where(ok, x, torch.full_like(x, -2.0))
    y = torch.where(ok, y, torch.full_like(y, -2.0))
    x0, y0 = torch.floor(x), torch.floor(y)
    fx, fy = x - x0, y - y0
    x0, y0 = x0.long(), y0.long()
    flat = imgs.reshape(J, H * W)

    def tap(yy, xx):
        inside = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        v = torch.gather(flat, 1, (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(J, H * W)).reshape(J, H, W)
        return torch.where(inside, v, torch.zeros_like(v))

    top = tap(y0, x0) * (1 - fx) + tap(y0, x0 + 1) * fx
    bot = tap(y0 + 1, x0) * (1 - fx) + tap(y0 + 1, x0 + 1) * fx
    return top * (1 - fy) + bot * fy


def _reproject_sources(depth_ref, intrinsics_ref, extrinsics_ref, depths_src, intrinsics_src, extrinsics_src):
    """reproject_with_depth of ONE reference view against J source views at once (the same operations on stacked
    operands: one batched launch per step instead of one per source view).  Returns (depth_rep, x_rep, y_rep), each (J,H,W)."""
    H, W = depth_ref.shape
    J = depths_src.shape[0]
    dt = intrinsics_ref.dtype
    dev = depth_ref.device
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=dt), torch.arange(W, device=dev, dtype=dt), indexing="ij")
    x_ref, y_ref = xs.reshape(-1), ys.reshape(-1)
    ones = torch.ones_like(x_ref)
    xyz_ref = torch.linalg.inv(intrinsics_ref) @ (torch.stack((x_ref, y_ref, ones)) * depth_ref.reshape(-1).to(dt))
    hom_ref = torch.cat((xyz_ref, ones[None]))                                             # (4, HW)
    inv_ref, inv_src = torch.linalg.inv(extrinsics_ref), torch.linalg.inv(extrinsics_src)  # (4,4), (J,4,4)
    xyz_src = ((extrinsics_src @ inv_ref) @ hom_ref)[:, :3]                                # (J,3,HW)
    k_src = intrinsics_src @ xyz_src
    xy_src = k_src[:, :2] / k_src[:, 2:3]
    x_src = xy_src[:, 0].reshape(J, H, W).float()
    y_src = xy_src[:, 1].reshape(J, H, W).float()
    sampled = _bilinear_zeros_batch(depths_src.float(), x_src, y_src)
    ones_j = ones[None, None].expand(J, 1, -1)
    xyz_src2 = torch.linalg.inv(intrinsics_src) @ (torch.cat((xy_src, ones_j), 1) * sampled.reshape(J, 1, -1).to(dt))
    xyz_rep = ((extrinsics_ref @ inv_src) @ torch.cat((xyz_src2, ones_j), 1))[:, :3]
    depth_rep = xyz_rep[:, 2].reshape(J, H, W).float()
    k_rep = intrinsics_ref @ xyz_rep
    xy_rep = k_rep[:, :2] / k_rep[:, 2:3]
    return depth_rep, xy_rep[:, 0].reshape(J, H, W).float(), xy_rep[:, 1].reshape(J, H, W).float()


def geocheck(intrs: torch.Tensor, c2ws: torch.Tensor, depths: torch.Tensor, dist_thresh: float = 1.0,
             depth_thresh: float = 0.01, view_thresh: int = 5, num_src: int = 15
             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """utils/geo_check.py:33-88.  intrs (N,3,3), c2ws (N,4,4) [used as the reference uses them: as the view
    transform of each camera], depths (N,H,W).  Returns (filtered depths (N,H,W), masks (N,H,W) float).
    A reference view's source views are processed as ONE stack (round 5: the double loop of round 4 issued ~40 small launches
    per (view, source) pair)."""
    n = intrs.shape[0]
    pairs = get_pairs(c2ws, num_src)
    H, W = depths.shape[1:]
    dev = depths.device
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32),
                            torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    out_d, out_m = [], []
    for i in range(n):
        depth_ref = depths[i].float()
        js = pairs[i]
        d_rep, x_rep, y_rep = _reproject_sources(depths[i], intrs[i], c2ws[i], depths[js], intrs[js], c2ws[js])
        dist = torch.sqrt((x_rep - xs) ** 2 + (y_rep - ys) ** 2)
        rel = (d_rep - depth_ref).abs() / depth_ref
        mask = (dist < dist_thresh) & (rel < depth_thresh)
        mask_sum = mask.sum(0, dtype=torch.int32)
        depth_sum = torch.where(mask, d_rep, torch.zeros_like(d_rep)).sum(0)
        averaged = (depth_sum + depth_ref) / (mask_sum + 1).float()
        final = mask_sum > view_thresh
        out_d.append(averaged * final.float())
        out_m.append(final.float())
    return torch.stack(out_d), torch.stack(out_m)
