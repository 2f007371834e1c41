#!/usr/bin/env python3
"""Long form of tests/test_gpu_parity.py::test_random_configurations_forward_and_backward: many seeded random
configurations (odd image sizes, splat scales, SH degrees, cameras), HIP path vs the CPU oracle — integer stages bit
for bit, images and gradients within 1e-4.  Usage: tools/fuzz_parity.py [first_seed] [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import parity_utils as pu
import test_gpu_parity as T
from scgaussian_amd import synthetic as syn

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
worst = {}
bad = 0
flips = 0
t0 = time.time()
for seed in range(first, first + count):
    P, W, H, deg, bg, mod, camspec, sd, lsm = T._random_config(seed)
    sc = syn.make_scene(P, W, H, seed=sd, log_scale_mean=lsm)
    cam = T._cam(camspec, W, H)
    grads = syn.make_upstream_grads(W, H, seed=20 + seed)
    mode = ("sh_sr", "col_sr", "sh_cov", "col_cov")[(seed // 7) % 4]        # SH / precomputed colours x scale+rot / cov3D
    try:
        o = pu.run_oracle(sc, cam, deg, bg, mod, mode=mode, grads=grads)
    except RuntimeError as e:                      # nothing visible: the oracle's outputs do not depend on its inputs
        if "does not require grad" not in str(e):
            raise
        o = pu.run_oracle(sc, cam, deg, bg, mod, mode=mode)
        o["grads"] = {}
        h0 = pu.run_hip(sc, cam, deg, bg, mod, mode=mode, grads=grads)
        assert all(float(g.abs().max()) == 0.0 for g in h0["grads"].values()), ("non-zero gradient of an empty render", seed)
        print("nothing visible: seed", seed, (P, W, H), "- HIP gradients are exactly zero", flush=True)
    fs = T._stages(sc, cam, deg, bg, mod, mode=mode)
    b = o["aux"]["binning"]
    ok = torch.equal(fs["radii"].cpu(), o["radii"]) and np.array_equal(pu.as_u32(fs["point_list"]), b["point_list"]) \
        and np.array_equal(pu.as_u32(fs["ranges"]), b["ranges"])
    h = pu.run_hip(sc, cam, deg, bg, mod, mode=mode, grads=grads)
    errs = {k: pu.nrm_err(h[k], o[k]) for k in ("color", "depth", "alpha")}
    errs.update({"d" + k: pu.nrm_err(h["grads"][k], g) for k, g in o["grads"].items()})
    if ok and max(errs.values()) >= pu.REL_TOL:
        # a splat whose alpha is within an ulp of the 1/255 cut (or a transmittance within an ulp of 1e-4) is blended by
        # one fp32 implementation and skipped by the other: ONE pixel differs by ~1/255 and takes its gradients along.
        # Inherent to comparing two fp32 evaluation orders (1 in ~1200 configurations here); reported, not a failure.
        n_pix = int(((h["alpha"].cpu() - o["alpha"]).abs() > 1e-5).sum())
        if n_pix <= 2:
            flips += 1
            print("threshold flip: seed", seed, mode, (P, W, H), n_pix, "pixel(s) differ", flush=True)
            continue
    for k, v in errs.items():
        worst[k] = max(worst.get(k, 0.0), v)
    if (seed - first) % 500 == 499:
        print("...", seed - first + 1, "done", flush=True)
    if not ok or max(errs.values()) >= pu.REL_TOL:
        bad += 1
        print("MISMATCH seed", seed, mode, (P, W, H, deg, bg, mod, camspec), "integers ok" if ok else "INTEGER STAGES DIFFER", errs,
              flush=True)
print(f"{count} configurations, {bad} mismatches, {flips} threshold flips, {time.time() - t0:.0f} s; worst normalised errors:",
      {k: float(f"{v:.2e}") for k, v in worst.items()})
sys.exit(1 if bad else 0)
