#!/usr/bin/env python3
"""The one-call path (scg_forward: geometry kernel that builds the slice histograms, column scan, scatter that publishes the
tile starts, forward blend that sorts its own tiles) against the staged calls (one kernel per step) on random scenes — uniform
(odd image sizes, any Gaussian count incl. less than a 256-block) and clustered (lists up to ~100 000 entries, tied
depths) — bit for bit: sorted lists, ranges, images, final_T, n_contrib, radii; with the sort / the histogram kept apart too.
Usage: tools/fuzz_fused.py [first_seed] [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_gpu_parity as T
from scgaussian_amd import synthetic as syn

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
longest = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    try:
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
        longest = max(longest, int(counts.max()) if len(counts) else 0)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:200], flush=True)
print(f"{count} scenes, {bad} mismatches, longest list {longest}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
