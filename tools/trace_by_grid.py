#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 kernel_trace.csv, separated by grid size; the kernels that run one workgroup per
slice of the Gaussians (same grid for every image size) by the forward they belong to (tools/_workload_tag.py)."""
import collections
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _workload_tag as wt
rows = list(csv.DictReader(open(sys.argv[1])))
tag = wt.tags(rows, "Grid_Size_X")
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('scg::', '').replace('void ', '')[:40]
    of = tag.get(int(r["Dispatch_Id"])) if wt.short(r['Kernel_Name']) in wt.AMBIGUOUS else None
    agg[(n, int(r['Grid_Size_X']), of)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, g, of), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if not n.startswith('at::') and not n.startswith('__amd'):
        where = f" (forward of blend grid {of})" if of else ""
        print(f"{n:40s} grid {g:8d} calls {len(v):4d} avg_us {sum(v)/len(v):8.2f} min {min(v):8.2f} max {max(v):8.2f}{where}")
