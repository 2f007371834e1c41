#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 kernel_trace.csv, separated by grid size AND LDS allocation (the kernels that run one
workgroup per slice of the Gaussians have the same grid for every image size; their LDS follows the tile count)."""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('scg::', '').replace('void ', '')[:40]
    agg[(n, int(r['Grid_Size_X']), int(r.get('LDS_Block_Size', 0) or 0))].append(
        (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, g, lds), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if not n.startswith('at::') and not n.startswith('__amd'):
        print(f"{n:40s} grid {g:8d} lds {lds:6d} calls {len(v):4d} avg_us {sum(v)/len(v):8.2f} min {min(v):8.2f} max {max(v):8.2f}")
