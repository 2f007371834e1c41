#!/usr/bin/env python3
"""Per-kernel, per-grid-size durations from a rocprofv3 kernel_trace.csv (separates the S2 and S3 launches)."""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('scg::', '').replace('void ', '')[:40]
    agg[(n, int(r['Grid_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if not n.startswith('at::') and not n.startswith('__amd'):
        print(f"{n:40s} grid {g:8d} calls {len(v):4d} avg_us {sum(v)/len(v):8.2f} min {min(v):8.2f} max {max(v):8.2f}")
