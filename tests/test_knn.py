"""distCUDA2 replacement: oracle self-consistency on CPU, HIP kernel vs oracle on GPU."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle as ko


def _cloud(n, seed, dup=False):
    rng = np.random.default_rng(seed)
    p = rng.normal(size=(n, 3)).astype(np.float32) * np.array([3.0, 1.0, 0.3], dtype=np.float32)
    if dup and n > 10:
        p[n // 2: n // 2 + n // 10] = p[: n // 10]        # exact duplicates -> zero distances
    return p


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 64, 1000, 5000])
def test_oracle_bruteforce_matches_kdtree(n):
    p = _cloud(n, n, dup=n >= 1000)
    a = ko.mean_dist2_bruteforce(p)
    b = ko.mean_dist2_kdtree(p)
    np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 255, 256, 257, 1025, 12000, 60000])
def test_distcuda2_matches_oracle(n):
    from simple_knn._C import distCUDA2
    p = _cloud(n, 7 * n + 1, dup=n >= 1000)
    out = distCUDA2(torch.from_numpy(p).cuda()).cpu().numpy()
    ref = ko.mean_dist2_bruteforce(p)
    assert out.shape == (n,) and out.dtype == np.float32
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-7)
    if n >= 1000:      # duplicated points have a zero-distance neighbour: their mean is at most 2/3 of an undup'd one
        dup = np.r_[np.arange(n // 10), np.arange(n // 2, n // 2 + n // 10)]
        assert np.median(out[dup]) < np.median(np.delete(out, dup))
    # the reference's use of it: scales = log(sqrt(clamp_min(dist2, 1e-7)))  (scene/gaussian_model.py:444-445)
    scales = torch.log(torch.sqrt(torch.clamp_min(torch.from_numpy(out), 1e-7)))
    assert torch.isfinite(scales).all()


@pytest.mark.gpu
def test_distcuda2_refuses_cpu_tensors():
    from simple_knn._C import distCUDA2
    from scgaussian_amd._lib import ScgError
    with pytest.raises(ScgError):
        distCUDA2(torch.zeros(10, 3))
