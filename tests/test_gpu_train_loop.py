"""End-to-end: the gradients of the HIP path drive Adam to fit a perturbed scene back to its ground-truth renders
(the reference's train.py loop reduced to the hot path)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fit_improves_psnr():
    spec = importlib.util.spec_from_file_location("fit_synthetic", os.path.join(ROOT, "examples", "fit_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hist = mod.fit(iters=150, P=3000, W=192, H=128, verbose=False)
    first, last = hist[0], hist[-1]
    assert last[1] < 0.6 * first[1], hist          # loss down by > 40 %
    assert last[2] > first[2] + 3.0, hist          # PSNR up by > 3 dB
