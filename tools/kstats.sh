#!/bin/bash
# rocprofv3 kernel stats of a command, printed as a compact table.  Usage: tools/kstats.sh <outdir-name> <cmd...>
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -o k -- "$@" > $R/gpurun_out/$name/cmd.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/$name/k_kernel_stats.csv")))
for r in rows[:22]:
    n=r["Name"].split("(")[0].replace("scg::","").replace("void ","")[:48]
    print(f"{n:48s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f} {float(r['Percentage']):5.1f}%")
PY
