import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
for nt in (8, 16, 32, 64):
    if nt > (os.cpu_count() or 1): break
    torch.set_num_threads(nt)
    os.cpu_count_orig = os.cpu_count
    t = time.time()
    _cc = os.cpu_count
    os.cpu_count = lambda: nt
    r = bench.cpu_baseline(200000, 1008, 756, 3, 32)
    os.cpu_count = _cc
    print(nt, "threads:", r["value"], "iters/s measured_s", r["measured_s"], "wall", round(time.time() - t, 1), flush=True)
