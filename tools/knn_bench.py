#!/usr/bin/env python3
"""Time the distCUDA2 replacement: ms per call and pair evaluations per second for a few N."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simple_knn._C import distCUDA2   # noqa: E402

for n in (2000, 12000, 100000, 400000):
    p = torch.randn(n, 3, device="cuda")
    for _ in range(2):
        distCUDA2(p)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10 if n <= 100000 else 3
    a.record()
    for _ in range(reps):
        distCUDA2(p)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"N={n:7d}  {ms:9.3f} ms/call  {n * n / (ms * 1e-3) / 1e12:6.2f} T pairs/s")
