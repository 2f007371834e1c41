#!/usr/bin/env python3
"""Idle gaps between the kernels of a training step, from a rocprofv3 kernel_trace.csv of bench.py.

    tools/gap_table.py <k_kernel_trace.csv> [blend grid of the workload, default 774144 = S2] [first:last step of the trace]

A step = the launches from one geometry_hist_kernel to the next (one view: forward + backward).  For every pair of consecutive
launches of a step the median / p90 of (start of the later kernel - end of the earlier one) over the steps of the trace, and the
median step span (first start -> last end) against the sum of its kernels."""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
blend_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 774144
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.split("(")[0].replace("scg::", "").replace("void ", "").split("<")[0].strip()[:34]


# steps that belong to the workload: those whose blend_backward has the workload's grid
steps, cur = [], None
for r in rows:
    n = short(r["Kernel_Name"])
    if n == "geometry_hist_kernel":
        if cur:
            steps.append(cur)
        cur = []
    if cur is None:
        continue                                   # (what precedes the first one-call forward: uploads, the staged first render)
    cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])))
if cur:
    steps.append(cur)
steps = [s for s in steps if any(n == "blend_backward_kernel" and g == blend_grid for n, _, _, g in s)]
# keep the launches up to and including geometry_backward (what follows belongs to the next step's host work)
trimmed = []
for s in steps:
    names = [n for n, *_ in s]
    if "geometry_backward_kernel" not in names:
        continue
    trimmed.append(s[: names.index("geometry_backward_kernel") + 1])
steps = trimmed
if len(sys.argv) > 3:
    a, b = (int(x) if x else None for x in sys.argv[3].split(":"))
    steps = steps[a:b]
print(f"{len(steps)} training steps of blend grid {blend_grid}" + (f" (steps {sys.argv[3]} of the trace)" if len(sys.argv) > 3 else ""))
gaps = collections.OrderedDict()
spans, sums, between = [], [], []
prev_end = None
for s in steps:
    spans.append((s[-1][2] - s[0][1]) / 1e3)
    sums.append(sum(e - b for _, b, e, _ in s) / 1e3)
    if prev_end is not None:
        between.append((s[0][1] - prev_end) / 1e3)
    prev_end = s[-1][2]
    for (n0, _, e0, _), (n1, b1, _, _) in zip(s, s[1:]):
        gaps.setdefault((n0, n1), []).append((b1 - e0) / 1e3)
for (a, b), v in gaps.items():
    v = sorted(v)
    print(f"  {a:34s} -> {b:34s} n {len(v):3d}  median {statistics.median(v):6.2f} us  p90 {v[int(0.9 * (len(v) - 1))]:6.2f}")
print(f"  step span (first start -> last end) median {statistics.median(spans):.1f} us; sum of its kernels {statistics.median(sums):.1f} us; "
      f"gaps inside a step {statistics.median(spans) - statistics.median(sums):.1f} us")
if between:
    print(f"  end of a step -> start of the next step's first kernel: median {statistics.median(between):.1f} us "
          f"(the caller's zeros_like / host work between steps)")
