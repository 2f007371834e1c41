import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # The in-tree HIP library is a build product (git-ignored): a fresh checkout has none.  Build it once before the
    # first test (hipcc cross-compiles gfx950 without a GPU, ~30 s); a tree that already has it is left alone — the
    # build step's source digest makes this a no-op.
    from scgaussian_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from scgaussian_amd import build as _build
        _build.build()


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: the gpu-marked tests are skipped, not failed."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (run with -m gpu through gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def ref_pieces():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "ref_pieces.npz"))


def pytest_sessionfinish(session, exitstatus):
    """Tolerance census: every parity comparison of the run (tensor-scale error, share of elements outside the
    element-wise 1e-4 bound, worst ratio) -> gpurun_out/tolerance_census.json, so the thresholds in parity_utils are
    set from measurements."""
    try:
        import json
        import parity_utils as pu
        if pu.CENSUS:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            rows = sorted(pu.CENSUS, key=lambda r: -r[2])
            with open(os.path.join(out, "tolerance_census.json"), "w") as fh:
                json.dump({"comparisons": len(rows), "max_share_outside": rows[0][2], "max_tensor_scale_error": max(r[1] for r in rows),
                           "worst": [dict(what=r[0], nrm_err=r[1], share_outside=r[2], worst_ratio=r[3], elements=r[4]) for r in rows[:40]]},
                          fh, indent=1)
    except Exception:               # never let bookkeeping fail a run
        pass
