"""Host side of the fused depth-consumer step: the reference's
GaussianModel.get_matchloss_from_renderdepth (scene/gaussian_model.py:241-282) as one autograd op."""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from . import _lib
from ._lib import check, ptr


class _MatchLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, pairs, width, height):
        lib = _lib.load()
        if not depth.is_cuda:
            raise _lib.ScgError("match loss needs tensors on the ROCm GPU ('cuda'); there is no CPU path")
        d = depth.detach().float().reshape(depth.shape[-2], depth.shape[-1]).contiguous()
        H, W = d.shape
        dev = d.device
        need_grad = ctx.needs_input_grad[0]
        # (the loss word and the gradient image the pairs' kernels add into: ONE zeroed buffer, one fill)
        buf = torch.zeros((H * W + 1 if need_grad else 1,), dtype=torch.float32, device=dev)
        loss = buf[-1:]
        grad = buf[: H * W].view(H, W) if need_grad else None
        keep = []
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            for p in pairs:
                t = {k: (None if p.get(k) is None else p[k].detach().to(dev).float().contiguous())
                     for k in ("uv0", "rays_o", "rays_d", "cam_rays_d", "mask0", "mask1", "intr1", "w2c1", "uv1")}
                keep.append(t)
                M = t["uv0"].shape[0]
                check(lib.scg_match_loss_pair(ptr(d), H, W, ptr(t["uv0"]), ptr(t["rays_o"]), ptr(t["rays_d"]),
                                              ptr(t["cam_rays_d"]), ptr(t["mask0"]), ptr(t["mask1"]), ptr(t["intr1"]),
                                              ptr(t["w2c1"]), ptr(t["uv1"]), M, float(width), float(height), ptr(loss),
                                              ptr(grad), stream), "scg_match_loss_pair")
        ctx.grad = grad
        ctx.shape = depth.shape
        ctx.in_dtype = depth.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g).reshape(ctx.shape).to(ctx.in_dtype), None, None, None


def match_loss_from_depth(depth: torch.Tensor, pairs: Sequence[Dict[str, torch.Tensor]], width: float, height: float):
    """depth: the rasterizer's rendered_depth (1,H,W) or (H,W).  pairs: one dict per matched view with the tensors
    the reference keeps in view_gs[...]["match_infos"][...]: uv0 (M,2), rays_o (M,3), rays_d (M,3), cam_rays_d (M,3),
    mask0 (M), mask1 (M) (or both None), intr1 (3,3), w2c1 (4,4), uv1 (M,2).  Returns the scalar match loss
    (sum over pairs of the masked means), differentiable w.r.t. depth."""
    return _MatchLoss.apply(depth, list(pairs), width, height)
