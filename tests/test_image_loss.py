"""Fused L1 + SSIM image loss against vectors produced by the REFERENCE's own
utils/loss_utils.py (tests/golden/make_golden.py): values and the gradient w.r.t. the rendered image."""
import numpy as np
import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_image_loss_matches_reference_golden(ref_pieces, tag):
    from scgaussian_amd import losses
    x = torch.from_numpy(ref_pieces[f"iloss_{tag}_img"]).cuda().requires_grad_(True)
    y = torch.from_numpy(ref_pieces[f"iloss_{tag}_gt"]).cuda()
    l1, s = losses.l1_and_ssim(x, y)
    assert abs(float(l1) - float(ref_pieces[f"iloss_{tag}_l1"])) < 1e-6
    assert abs(float(s) - float(ref_pieces[f"iloss_{tag}_ssim"])) < 2e-6
    loss = losses.image_loss(x, y, 0.2)
    assert abs(float(loss) - float(ref_pieces[f"iloss_{tag}_loss"])) < 2e-6
    loss.backward()
    ref = ref_pieces[f"iloss_{tag}_grad"]
    err = np.abs(x.grad.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 1e-4, err


@pytest.mark.gpu
def test_image_loss_matches_small_golden_and_separate_functions(ref_pieces):
    from scgaussian_amd import losses
    a = torch.from_numpy(ref_pieces["loss_img1"]).cuda()
    b = torch.from_numpy(ref_pieces["loss_img2"]).cuda()
    assert abs(float(losses.l1_loss(a, b)) - float(ref_pieces["loss_l1"])) < 1e-6
    assert abs(float(losses.ssim(a, b)) - float(ref_pieces["loss_ssim"])) < 2e-6
    # batched input and identical images
    assert abs(float(losses.ssim(a[None], a[None])) - 1.0) < 1e-6
    assert float(losses.l1_loss(a, a)) == 0.0


@pytest.mark.gpu
def test_image_loss_gradient_against_torch_autograd_at_training_size():
    """1008x756 (BASELINE cfg 2): gradient vs a plain-torch fp32 SSIM (conv2d) on the same GPU."""
    import torch.nn.functional as F
    from scgaussian_amd import losses
    g = torch.Generator().manual_seed(0)
    H, W = 756, 1008
    x = torch.rand(3, H, W, generator=g).cuda().requires_grad_(True)
    y = (x.detach().cpu() + 0.1 * torch.randn(3, H, W, generator=g)).clamp(0, 1).cuda()
    loss = losses.image_loss(x, y, 0.2)
    loss.backward()
    got = x.grad.clone()
    x.grad = None
    gauss = torch.tensor([np.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)], dtype=torch.float32)
    gauss = gauss / gauss.sum()
    win = (gauss[:, None] @ gauss[None, :]).expand(3, 1, 11, 11).contiguous().cuda()
    conv = lambda t: F.conv2d(t[None], win, padding=5, groups=3)[0]      # noqa: E731
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    smap = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    ref_loss = 0.8 * (x - y).abs().mean() + 0.2 * (1 - smap.mean())
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-5
    err = float((got - x.grad).abs().max() / x.grad.abs().max())
    assert err < 1e-4, err


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float64])
def test_image_and_match_loss_gradients_come_back_in_the_input_dtype(dtype):
    """Non-fp32 inputs are converted for the kernels; autograd must get the gradient back in the input's dtype."""
    from scgaussian_amd import losses
    from scgaussian_amd.match_loss import match_loss_from_depth
    g = torch.Generator().manual_seed(1)
    x = torch.rand(3, 40, 56, generator=g).cuda().to(dtype).requires_grad_(True)
    y = torch.rand(3, 40, 56, generator=g).cuda()
    losses.image_loss(x, y, 0.2).backward()
    assert x.grad is not None and x.grad.dtype == dtype and torch.isfinite(x.grad.float()).all()
    d = (torch.rand(1, 40, 56, generator=g) * 3 + 4).cuda().to(dtype).requires_grad_(True)
    M = 50
    pair = dict(uv0=torch.rand(M, 2, generator=g).cuda() * 40, rays_o=torch.zeros(M, 3).cuda(),
                rays_d=torch.nn.functional.normalize(torch.rand(M, 3, generator=g) + torch.tensor([0.0, 0.0, 2.0]), dim=1).cuda(),
                cam_rays_d=torch.nn.functional.normalize(torch.rand(M, 3, generator=g) + torch.tensor([0.0, 0.0, 2.0]), dim=1).cuda(),
                mask0=None, mask1=None, intr1=torch.tensor([[50.0, 0, 28], [0, 50.0, 20], [0, 0, 1]]).cuda(),
                w2c1=torch.eye(4).cuda(), uv1=torch.rand(M, 2, generator=g).cuda() * 40)
    match_loss_from_depth(d, [pair], 56.0, 40.0).backward()
    assert d.grad is not None and d.grad.dtype == dtype


@pytest.mark.gpu
@pytest.mark.parametrize("lam,scale", [(0.2, 1.0), (0.0, 1.0), (1.0, 1.0), (0.35, -2.5)])
def test_combined_image_loss_equals_the_separate_calls_and_the_expression_around_them(lam, scale):
    """losses.image_loss (one scalar from the reduction kernel, the upstream gradient read by the backward kernel) against
    l1_and_ssim + the reference's expression in torch (train.py:160-161): the value to the last bit or two (each operation
    rounded on its own, the reference's order), the image gradient to 1e-6 of its maximum; with an upstream gradient that is
    not 1 (the loss scaled and summed with another term, as the match loss is)."""
    from scgaussian_amd import losses
    g = torch.Generator().manual_seed(3)
    H, W = 97, 130
    x0 = torch.rand(3, H, W, generator=g).cuda()
    y = (x0.cpu() + 0.1 * torch.randn(3, H, W, generator=g)).clamp(0, 1).cuda()
    xa = x0.clone().requires_grad_(True)
    xb = x0.clone().requires_grad_(True)
    la = losses.image_loss(xa, y, lam)
    l1, s = losses.l1_and_ssim(xb, y)
    lb = (1.0 - lam) * l1 + lam * (1.0 - s)
    assert la.shape == lb.shape == ()
    assert abs(float(la) - float(lb)) <= 2.5e-7 * max(1.0, abs(float(lb))), (float(la), float(lb))
    (scale * la + 1.0).backward()
    (scale * lb + 1.0).backward()
    err = float((xa.grad - xb.grad).abs().max() / xb.grad.abs().max())
    assert err < 1e-6, err
    with torch.no_grad():                                   # no gradient wanted: no derivative maps are written
        assert abs(float(losses.image_loss(x0, y, lam)) - float(la)) == 0.0


@pytest.mark.gpu
def test_image_loss_takes_any_lambda_the_reference_expression_takes():
    """a lambda outside [0, 1] or a tensor lambda (with a gradient of its own) is combined in torch from the fused
    L1 / SSIM pair — the combined kernel only takes a plain number in [0, 1]."""
    from scgaussian_amd import losses
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 48, 64, generator=g).cuda().requires_grad_(True)
    y = torch.rand(3, 48, 64, generator=g).cuda()
    l1, s = losses.l1_and_ssim(x.detach(), y)
    for lam in (1.5, -0.25):
        got = losses.image_loss(x, y, lam)
        assert abs(float(got) - float((1.0 - lam) * l1 + lam * (1.0 - s))) <= 1e-6
    lam_t = torch.tensor(0.3, device="cuda", requires_grad=True)
    losses.image_loss(x, y, lam_t).backward()
    assert x.grad is not None and lam_t.grad is not None
    assert abs(float(lam_t.grad) - float((1.0 - s) - l1)) <= 1e-6       # d/d lambda of (1 - lambda) L1 + lambda (1 - SSIM)
