#!/usr/bin/env python3
"""Generate tests/golden/ref_pieces.npz by IMPORTING the reference's Python (read-only, from
/root/reference) in the build container.  Only the resulting input/output vectors are
committed; no reference source travels.

Pins:
  * SH -> RGB:  utils/sh_utils.py:57-112 eval_sh + the +0.5 / clamp_min(0) of
                gaussian_renderer/__init__.py:82-83, degrees 0..3
  * cov3D:      utils/general_utils.py:70-116 build_scaling_rotation + strip_symmetric, as
                composed by scene/gaussian_model.py:37-41
  * cameras:    utils/graphics_utils.py:38-71 getWorld2View2 / getProjectionMatrix composed as
                scene/cameras.py:60-63
  * losses:     utils/loss_utils.py l1_loss / ssim, utils/image_utils.py psnr (for the
                training-step twin)

Run:  python tests/golden/make_golden.py      (needs /root/reference; CPU only)
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_pieces.npz")


def _cpu_device_shim():
    """The reference hard-codes device='cuda' (utils/general_utils.py:71,89,108); there is no
    GPU in the build container, so redirect those factory calls to the CPU."""
    for name in ("zeros", "ones", "tensor", "empty"):
        orig = getattr(torch, name)

        def wrapped(*a, __orig=orig, **kw):
            if "device" in kw and "cuda" in str(kw["device"]):
                kw["device"] = "cpu"
            return __orig(*a, **kw)
        setattr(torch, name, wrapped)


def main():
    sys.path.insert(0, REF)
    _cpu_device_shim()
    from utils.sh_utils import eval_sh                       # noqa: E402
    from utils.general_utils import build_scaling_rotation, strip_symmetric   # noqa: E402
    from utils.graphics_utils import getWorld2View2, getProjectionMatrix      # noqa: E402
    from utils.loss_utils import l1_loss, ssim               # noqa: E402
    from utils.image_utils import psnr                       # noqa: E402

    g = torch.Generator().manual_seed(1234)
    out = {}

    # ---- SH -> RGB ----------------------------------------------------------------------
    P = 257
    shs = torch.cat([torch.rand(P, 1, 3, generator=g) * 3 - 1.5, torch.randn(P, 15, 3, generator=g) * 0.3], 1)
    xyz = torch.randn(P, 3, generator=g) * 4
    campos = torch.tensor([0.3, -0.2, 0.1])
    out["sh_shs"], out["sh_xyz"], out["sh_campos"] = shs.numpy(), xyz.numpy(), campos.numpy()
    for deg in range(4):
        # exactly the python twin at gaussian_renderer/__init__.py:79-83
        shs_view = shs.transpose(1, 2).view(-1, 3, 16)
        dir_pp = xyz - campos.repeat(P, 1)
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        sh2rgb = eval_sh(deg, shs_view, dir_pp_normalized)
        out[f"sh_rgb_deg{deg}"] = torch.clamp_min(sh2rgb + 0.5, 0.0).numpy()
        out[f"sh_raw_deg{deg}"] = (sh2rgb + 0.5).numpy()

    # ---- cov3D ---------------------------------------------------------------------------
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.7 - 3.0)
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)     # scene/gaussian_model.py:118 normalises first
    out["cov_scales"], out["cov_rot"] = scales.numpy(), rot.numpy()
    for mod in (1.0, 0.37):
        L = build_scaling_rotation(mod * scales, rot)          # scene/gaussian_model.py:38
        cov = L @ L.transpose(1, 2)
        out[f"cov_sym_mod{mod}"] = strip_symmetric(cov).numpy()

    # ---- cameras -------------------------------------------------------------------------
    cams = []
    rng = np.random.default_rng(7)
    for i in range(6):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        r, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                      [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                      [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        if i == 0:
            R = np.eye(3)
        T = rng.normal(size=3) * (0.0 if i == 0 else 2.0)
        fovx, fovy = float(rng.uniform(0.5, 1.4)), float(rng.uniform(0.4, 1.1))
        wv = torch.tensor(getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wv.inverse()[3, :3]
        cams.append(dict(R=R, T=T, fovx=fovx, fovy=fovy, wv=wv.numpy(), full=full.numpy(), center=center.numpy()))
    for k in cams[0]:
        out[f"cam_{k}"] = np.stack([np.asarray(c[k]) for c in cams])

    # ---- losses around the path (training-step twin) ---------------------------------------
    img1 = torch.rand(3, 40, 52, generator=g)
    img2 = (img1 + 0.1 * torch.randn(3, 40, 52, generator=g)).clamp(0, 1)
    out["loss_img1"], out["loss_img2"] = img1.numpy(), img2.numpy()
    out["loss_l1"] = np.float32(l1_loss(img1, img2).item())
    out["loss_ssim"] = np.float32(ssim(img1, img2).item())
    out["loss_psnr"] = np.float32(psnr(img1[None], img2[None]).mean().item())

    # gradient of the training image loss (train.py:160-161) w.r.t. the rendered image, by the reference's own code
    for tag, (hh, ww) in {"a": (40, 52), "b": (37, 70)}.items():
        x = torch.rand(3, hh, ww, generator=g).requires_grad_(True)
        y = (x.detach() + 0.15 * torch.randn(3, hh, ww, generator=g)).clamp(0, 1)
        ll1 = l1_loss(x, y)
        s_ = ssim(x, y)
        loss = (1.0 - 0.2) * ll1 + 0.2 * (1.0 - s_)
        loss.backward()
        out[f"iloss_{tag}_img"], out[f"iloss_{tag}_gt"] = x.detach().numpy(), y.numpy()
        out[f"iloss_{tag}_l1"], out[f"iloss_{tag}_ssim"] = np.float32(ll1.item()), np.float32(s_.item())
        out[f"iloss_{tag}_loss"], out[f"iloss_{tag}_grad"] = np.float32(loss.item()), x.grad.numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
