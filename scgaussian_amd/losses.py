"""Fused image loss of the training step, host side.

Mirrors the names of the reference's utils/loss_utils.py (`l1_loss` :40, `ssim` :56-94) and adds the fused form of
train.py:160-161:   loss = (1 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1 - ssim(image, gt)).
All three go through ONE forward kernel and ONE backward kernel of libscg_raster.so (include/scg_loss.h)."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check


class _ImageLoss(torch.autograd.Function):
    """(img, gt) -> (l1, ssim) scalars; gradient w.r.t. img only (gt is data)."""

    @staticmethod
    def forward(ctx, img, gt):
        lib = _lib.load()
        if not img.is_cuda:
            raise _lib.ScgError("image loss needs tensors on the ROCm GPU ('cuda'); there is no CPU path")
        shape = img.shape
        if img.dim() == 4:
            C, H, W = shape[0] * shape[1], shape[2], shape[3]
        elif img.dim() == 3:
            C, H, W = shape
        else:
            raise ValueError("img must be (C,H,W) or (B,C,H,W)")
        x = img.detach().float().contiguous()
        y = gt.detach().to(x.device).float().contiguous()
        if y.shape != x.shape:
            raise ValueError("img and gt shapes differ")
        dev = x.device
        need_grad = ctx.needs_input_grad[0]
        with torch.cuda.device(dev):
            sums = torch.empty((2,), dtype=torch.float32, device=dev)
            dmaps = torch.empty((3, C, H, W), dtype=torch.float32, device=dev) if need_grad else None
            nbytes = lib.scg_image_loss_scratch_bytes(C, H, W)
            scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            check(lib.scg_image_loss_forward(x.data_ptr(), y.data_ptr(), C, H, W, sums.data_ptr(),
                                             None if dmaps is None else dmaps.data_ptr(), scratch.data_ptr(), nbytes,
                                             torch.cuda.current_stream(dev).cuda_stream), "scg_image_loss_forward")
        ctx.dims = (C, H, W)
        ctx.shape = shape
        ctx.in_dtype = img.dtype                       # the gradient goes back in the dtype the image came in
        if need_grad:
            ctx.save_for_backward(x, y, dmaps)
        out = sums / float(C * H * W)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_l1, g_ssim):
        lib = _lib.load()
        x, y, dmaps = ctx.saved_tensors
        C, H, W = ctx.dims
        n = float(C * H * W)
        # upstream gradients are device scalars: the weights stay on the device (no host read)
        w = torch.stack([g_l1.reshape(()), g_ssim.reshape(())]).float() * (1.0 / n)
        d_img = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib.scg_image_loss_backward(x.data_ptr(), y.data_ptr(), dmaps.data_ptr(), C, H, W, w.data_ptr(),
                                              d_img.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream),
                  "scg_image_loss_backward")
        return d_img.reshape(ctx.shape).to(ctx.in_dtype), None


class _ImageLossCombined(torch.autograd.Function):
    """(img, gt, lambda_dssim) -> the training loss (1 - l) * L1 + l * (1 - SSIM) as ONE scalar, formed by the forward's reduction
    kernel; the backward kernel takes the scalar's upstream gradient from device memory.  Two library calls per iteration and no
    tensor arithmetic around them: the separate form (_ImageLoss + the expression in torch) launches about a dozen one-element
    kernels per iteration, forward and backward — as much host and queue time as the loss kernels themselves (round 5)."""

    @staticmethod
    def forward(ctx, img, gt, lam):
        lib = _lib.load()
        if not img.is_cuda:
            raise _lib.ScgError("image loss needs tensors on the ROCm GPU ('cuda'); there is no CPU path")
        shape = img.shape
        if img.dim() == 4:
            C, H, W = shape[0] * shape[1], shape[2], shape[3]
        elif img.dim() == 3:
            C, H, W = shape
        else:
            raise ValueError("img must be (C,H,W) or (B,C,H,W)")
        x = img.detach().float().contiguous()
        y = gt.detach().to(x.device).float().contiguous()
        if y.shape != x.shape:
            raise ValueError("img and gt shapes differ")
        dev = x.device
        need_grad = ctx.needs_input_grad[0]
        with torch.cuda.device(dev):
            sums = torch.empty((3,), dtype=torch.float32, device=dev)
            dmaps = torch.empty((3, C, H, W), dtype=torch.float32, device=dev) if need_grad else None
            nbytes = lib.scg_image_loss_scratch_bytes(C, H, W)
            scratch = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            check(lib.scg_image_loss_forward_combined(x.data_ptr(), y.data_ptr(), C, H, W, float(lam), sums.data_ptr(),
                                                      None if dmaps is None else dmaps.data_ptr(), scratch.data_ptr(), nbytes,
                                                      torch.cuda.current_stream(dev).cuda_stream),
                  "scg_image_loss_forward_combined")
        ctx.dims, ctx.shape, ctx.in_dtype, ctx.lam = (C, H, W), shape, img.dtype, float(lam)
        if need_grad:
            ctx.save_for_backward(x, y, dmaps)
        return sums[2]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, y, dmaps = ctx.saved_tensors
        C, H, W = ctx.dims
        g = g.detach()
        if g.dtype != torch.float32 or g.device != x.device:
            g = g.to(x.device, torch.float32)
        g = g.contiguous()
        d_img = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(lib.scg_image_loss_backward_combined(x.data_ptr(), y.data_ptr(), dmaps.data_ptr(), C, H, W, g.data_ptr(),
                                                       ctx.lam, d_img.data_ptr(),
                                                       torch.cuda.current_stream(x.device).cuda_stream),
                  "scg_image_loss_backward_combined")
        return d_img.reshape(ctx.shape).to(ctx.in_dtype), None, None


def l1_and_ssim(img: torch.Tensor, gt: torch.Tensor):
    """(l1_loss(img, gt), ssim(img, gt)) from one fused kernel."""
    return _ImageLoss.apply(img, gt)


def l1_loss(network_output: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    return _ImageLoss.apply(network_output, gt)[0]


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True) -> torch.Tensor:
    if window_size != 11 or not size_average:
        raise NotImplementedError("the fused kernel implements the reference's default: 11x11 window, mean over all")
    return _ImageLoss.apply(img1, img2)[1]


def image_loss(image: torch.Tensor, gt_image: torch.Tensor, lambda_dssim=0.2) -> torch.Tensor:
    """train.py:160-161 in one call: loss = (1 - lambda_dssim) * l1_loss + lambda_dssim * (1 - ssim).
    The combined kernel takes a plain number in [0, 1] (the reference's 0.2, arguments/__init__.py:86); a tensor lambda (which
    may carry a gradient) or a value outside that range is combined in torch from the fused L1 / SSIM pair, as the reference's
    expression accepts either."""
    if isinstance(lambda_dssim, torch.Tensor) or not (0.0 <= float(lambda_dssim) <= 1.0):
        l1, s = _ImageLoss.apply(image, gt_image)
        return (1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - s)
    return _ImageLossCombined.apply(image, gt_image, lambda_dssim)
