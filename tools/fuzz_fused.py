#!/usr/bin/env python3
"""Long form of tests/test_gpu_fuzz.py: the one-call path (scg_forward: geometry kernel that builds the slice histograms,
column scan, scatter that publishes the tile starts, forward blend that sorts its own tiles) against the staged calls (one
kernel per step) on random scenes, bit for bit (tests/fuzz_cases.py fused_one).  Usage: tools/fuzz_fused.py [first_seed] [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fuzz_cases as F

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad, longest = 0, 0
t0 = time.time()
for seed in range(first, first + count):
    try:
        longest = max(longest, F.fused_one(seed))
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:200], flush=True)
print(f"{count} scenes, {bad} mismatches, longest list {longest}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
