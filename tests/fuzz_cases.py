"""One seeded random configuration per call — shared by the driver-run suite (tests/test_gpu_fuzz.py: a slice of the seeds)
and by the long-form fuzzers (tools/fuzz_parity.py, tools/fuzz_fused.py: hundreds to thousands of seeds, builder-run)."""
from __future__ import annotations

import numpy as np
import torch

import parity_utils as pu
from scgaussian_amd import synthetic as syn

MODES = ("sh_sr", "col_sr", "sh_cov", "col_cov")          # SH / precomputed colours x scale+rot / cov3D


def parity_one(seed: int):
    """HIP path vs the CPU oracle on random configuration `seed` (odd image sizes, splat scales, SH degrees 0-3, cameras, the
    four input modes).  Returns (status, errs, info): status 'ok' | 'flip' (one or two pixels differ by a 1-ulp alpha / T
    threshold decision: inherent to two fp32 evaluation orders, reported separately) | 'empty' (nothing visible: HIP gradients
    exactly zero) | 'MISMATCH'; errs = normalised errors per tensor; info = what was run."""
    import test_gpu_parity as T
    P, W, H, deg, bg, mod, camspec, sd, lsm = T._random_config(seed)
    sc = syn.make_scene(P, W, H, seed=sd, log_scale_mean=lsm)
    cam = T._cam(camspec, W, H)
    grads = syn.make_upstream_grads(W, H, seed=20 + seed)
    mode = MODES[(seed // 7) % 4]
    info = (mode, (P, W, H, deg, bg, mod, camspec))
    status = "ok"
    try:
        o = pu.run_oracle(sc, cam, deg, bg, mod, mode=mode, grads=grads)
    except RuntimeError as e:                      # nothing visible: the oracle's outputs do not depend on its inputs
        if "does not require grad" not in str(e):
            raise
        o = pu.run_oracle(sc, cam, deg, bg, mod, mode=mode)
        o["grads"] = {}
        h0 = pu.run_hip(sc, cam, deg, bg, mod, mode=mode, grads=grads)
        if not all(float(g.abs().max()) == 0.0 for g in h0["grads"].values()):
            return "MISMATCH", {"gradient of an empty render": 1.0}, info
        status = "empty"
    fs = T._stages(sc, cam, deg, bg, mod, mode=mode)
    b = o["aux"]["binning"]
    ints_ok = torch.equal(fs["radii"].cpu(), o["radii"]) and np.array_equal(pu.as_u32(fs["point_list"]), b["point_list"]) \
        and np.array_equal(pu.as_u32(fs["ranges"]), b["ranges"])
    h = pu.run_hip(sc, cam, deg, bg, mod, mode=mode, grads=grads)
    errs = {k: pu.nrm_err(h[k], o[k]) for k in ("color", "depth", "alpha")}
    errs.update({"d" + k: pu.nrm_err(h["grads"][k], g) for k, g in o["grads"].items()})
    if not ints_ok:
        errs["INTEGER STAGES DIFFER"] = 1.0
        return "MISMATCH", errs, info
    if max(errs.values()) >= pu.REL_TOL:
        n_pix = int(((h["alpha"].cpu() - o["alpha"]).abs() > 1e-5).sum())
        return ("flip" if n_pix <= 2 else "MISMATCH"), errs, info
    return status, errs, info


def fused_one(seed: int) -> int:
    """The one-call path (scg_forward) against the staged calls on random scene `seed` — uniform (odd image sizes, any Gaussian
    count incl. less than a 256-block) for odd seeds, clustered (lists up to ~300 000 entries, tied depths) for even ones — bit
    for bit: sorted lists, ranges, images, final_T, n_contrib, radii; with the sort / the histogram kept apart too.  Raises
    AssertionError on a difference; returns the longest per-tile list."""
    import test_gpu_parity as T
    rng = np.random.default_rng(9000 + seed)
    if seed % 2:
        P, W, H, deg, bg, mod, camspec, sd, lsm = T._random_config(seed)
        sc = syn.make_scene(P, W, H, seed=sd, log_scale_mean=lsm)
        cam = T._cam(camspec, W, H)
    else:
        W, H = int(rng.integers(20, 400)), int(rng.integers(17, 300))
        P = int(rng.choice([1, 63, 255, 257, 3000, 20000, 110000, 300000]))
        spread = float(rng.choice([0.01, 0.05, 0.3, 1.0, 2.5]))
        tied = bool(rng.integers(0, 2))
        g = torch.Generator().manual_seed(seed)
        xy = (torch.rand(P, 2, generator=g) - 0.5) * spread
        z = (torch.randint(0, int(rng.integers(2, 200)), (P,), generator=g).float() * 0.05 + 3.0) if tied else \
            (torch.rand(P, generator=g) * float(rng.uniform(0.01, 9.0)) + 3.0)
        means = torch.cat([xy * z[:, None], z[:, None]], 1)
        sc = syn.Scene(means, torch.full((P, 3), float(rng.choice([0.002, 0.004, 0.02, 0.3]))),
                       torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1), torch.full((P, 1), 0.02),
                       torch.rand(P, 16, 3, generator=g) * 0.1)
        cam, deg, bg = syn.default_camera(W, H), int(rng.integers(0, 4)), (0.1, 0.0, 0.2)
    counts = T._fused_vs_staged(sc, cam, deg, bg)
    return int(counts.max()) if len(counts) else 0
