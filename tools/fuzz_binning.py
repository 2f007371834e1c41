#!/usr/bin/env python3
"""Tile-first binning (per-tile bucket / radix / long-list sorts) against the global 64-bit radix sort on random
CLUSTERED scenes — stacks of thousands to ~100 000 Gaussians on a few tiles, continuous or heavily tied depths — bit
for bit (sorted ids, ranges, sorted keys).  Usage: tools/fuzz_binning.py [first_seed] [count]"""
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
    rng = np.random.default_rng(5000 + seed)
    W, H = int(rng.integers(40, 200)), int(rng.integers(30, 150))
    P = int(rng.choice([3000, 9000, 20000, 50000, 110000]))
    spread = float(rng.choice([0.01, 0.05, 0.3, 1.0]))
    tied = bool(rng.integers(0, 2))
    g = torch.Generator().manual_seed(seed)
    xy = (torch.rand(P, 2, generator=g) - 0.5) * spread
    z = (torch.randint(0, int(rng.integers(2, 200)), (P,), generator=g).float() * 0.05 + 3.0) if tied else \
        (torch.rand(P, generator=g) * float(rng.uniform(0.01, 9.0)) + 3.0)
    means = torch.cat([xy * z[:, None], z[:, None]], 1)
    sc = syn.Scene(means, torch.full((P, 3), float(rng.choice([0.002, 0.004, 0.02]))),
                   torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1), torch.full((P, 1), 0.02),
                   torch.rand(P, 16, 3, generator=g) * 0.1)
    cam = syn.default_camera(W, H)
    a, b = (T._stages(sc, cam, 0, (0.0, 0.0, 0.0), algo=algo) for algo in (0, 1))
    same = a["num_rendered"] == b["num_rendered"] and torch.equal(a["point_list"], b["point_list"]) and \
        torch.equal(a["ranges"], b["ranges"]) and torch.equal(a["keys_sorted"], b["keys_sorted"])
    r = a["ranges"].cpu().numpy().astype(np.int64)
    longest = max(longest, int((r[:, 1] - r[:, 0]).max()))
    if not same:
        bad += 1
        print("MISMATCH seed", seed, (P, W, H, spread, tied), flush=True)
print(f"{count} scenes, {bad} mismatches, longest list {longest}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
