"""Cross-view depth consistency: the torch twin against the numpy restatement of
utils/geo_check.py, on a scene with known geometry."""
import numpy as np
import pytest
import torch

from oracle import geo_check_oracle as orc
from scgaussian_amd import geo_check as gc


def _scene(n=8, H=20, W=28, seed=0):
    """Cameras on a small arc looking at a slanted plane; depth maps rendered analytically, then corrupted."""
    rng = np.random.default_rng(seed)
    f = 30.0
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])
    intrs = np.repeat(K[None], n, 0)
    exts = np.zeros((n, 4, 4))
    depths = np.zeros((n, H, W), dtype=np.float32)
    nrm, d0 = np.array([0.1, -0.05, 1.0]), 6.0                 # plane nrm . X = d0 (world)
    for i in range(n):
        ang = 0.06 * (i - n / 2)
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = np.array([0.25 * (i - n / 2), 0.05 * i, 0.0])
        E = np.eye(4)
        E[:3, :3], E[:3, 3] = R, t                             # world -> camera
        exts[i] = E
        ys, xs = np.mgrid[0:H, 0:W]
        rays = np.linalg.inv(K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])    # camera rays, z = 1
        # camera point z*ray -> world R^T (z ray - t); plane: nrm . world = d0
        z = (d0 + nrm @ (R.T @ t)) / (nrm @ (R.T @ rays))
        depths[i] = z.reshape(H, W)
    bad = rng.random(depths.shape) < 0.04
    depths[bad] *= rng.uniform(1.05, 1.6, size=int(bad.sum())).astype(np.float32)      # inconsistent outliers
    depths[0, :3, :3] = 0.0                                                              # holes: divisions by zero
    return intrs, exts, depths, bad


def test_reproject_and_geocheck_match_numpy_restatement():
    intrs, exts, depths, bad = _scene()
    ti, te, td = (torch.from_numpy(a) for a in (intrs, exts, depths))
    assert np.array_equal(orc.get_pairs(exts, 4), gc.get_pairs(te, 4).numpy())
    ref = orc.reproject_with_depth(depths[2], intrs[2], exts[2], depths[5], intrs[5], exts[5])
    got = gc.reproject_with_depth(td[2], ti[2], te[2], td[5], ti[5], te[5])
    for r, g in zip(ref, got):
        fin = np.isfinite(r)
        assert np.array_equal(fin, torch.isfinite(g).numpy())
        assert np.allclose(r[fin], g.numpy()[fin], rtol=1e-5, atol=1e-4)
    fd, fm = orc.geocheck(intrs, exts, depths, view_thresh=2, num_src=5)
    gd, gm = gc.geocheck(ti, te, td, view_thresh=2, num_src=5)
    assert (fm != gm.numpy()).mean() < 0.002                   # threshold ties may flip a pixel
    same = fm == gm.numpy()
    assert np.allclose(fd[same], gd.numpy()[same], rtol=1e-5, atol=1e-5)
    # the check does its job: clean pixels mostly survive, corrupted ones mostly do not
    inner = np.zeros_like(bad)
    inner[:, 4:-4, 6:-6] = True
    assert fm[inner & ~bad].mean() > 0.8 and fm[inner & bad].mean() < 0.1


@pytest.mark.gpu
def test_geocheck_on_device_matches_cpu():
    intrs, exts, depths, _ = _scene(n=6, H=24, W=32, seed=1)
    ti, te, td = (torch.from_numpy(a) for a in (intrs, exts, depths))
    cd, cm = gc.geocheck(ti, te, td, view_thresh=2, num_src=4)
    dev = torch.device("cuda", 0)
    gd, gm = gc.geocheck(ti.to(dev), te.to(dev), td.to(dev), view_thresh=2, num_src=4)
    assert (cm != gm.cpu()).float().mean() < 0.005
    same = cm == gm.cpu()
    assert torch.allclose(cd[same], gd.cpu()[same], rtol=1e-4, atol=1e-4)
