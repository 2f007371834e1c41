#!/usr/bin/env python3
"""Time the fused L1+SSIM loss (fwd+bwd) against the plain-torch formulation the reference uses (5 grouped
conv2d + autograd) on the same GPU, at the BASELINE cfg-2 image size."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scgaussian_amd import losses   # noqa: E402


def torch_loss(x, y, win):
    conv = lambda t: F.conv2d(t[None], win, padding=5, groups=3)[0]      # noqa: E731
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    smap = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return 0.8 * (x - y).abs().mean() + 0.2 * (1 - smap.mean())


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (H, W) in ((378, 504), (756, 1008), (1080, 1920)):
    x = torch.rand(3, H, W, device="cuda", requires_grad=True)
    y = torch.rand(3, H, W, device="cuda")
    gauss = torch.tensor([np.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)], dtype=torch.float32)
    gauss = gauss / gauss.sum()
    win = (gauss[:, None] @ gauss[None, :]).expand(3, 1, 11, 11).contiguous().cuda()

    def fused():
        x.grad = None
        losses.image_loss(x, y, 0.2).backward()

    def plain():
        x.grad = None
        torch_loss(x, y, win).backward()

    tf, tp = timeit(fused), timeit(plain)
    px = 3 * H * W
    print(f"{W}x{H}: fused {tf*1e3:8.1f} us  torch {tp*1e3:8.1f} us  speed-up {tp/tf:5.1f}x   "
          f"fused traffic ~{px*4*13/1e6:.0f} MB -> {px*4*13/(tf*1e-3)/1e9:.0f} GB/s")
