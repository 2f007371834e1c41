#!/bin/bash
# Collect PMC counters for bench.py, one rocprofv3 pass per counter group (counters only with --kernel-trace).
# Usage: tools/pmc.sh <outdir-name> [bench args...]
# (the legs that run the path at OTHER settings — by_sh_degree, render_glue, captured_step, moving_scene — are switched off: their
#  kernels carry the same names and would be averaged into the headline workload's counters; the SH degrees the reference trains at
#  get passes of their own: tools/pmc.sh <tag>/pmc_deg0 --sh-degree 0 --no-s3, summarised as S2_deg0 by pmc_summary.py ...@deg0)
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/$name/p$i -o c -- python $R/bench.py --steps 6 --warmup 2 --sustained-steps 0 --no-cpu-baseline --no-full-iteration --no-small --no-clustered --no-rccl-floor --no-by-degree --no-render-glue --no-graph --no-moving-scene "$@" > $R/gpurun_out/$name/p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$R/gpurun_out/$name/p*/c_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].replace("scg::","").replace("void ","")[:34]
        if n.startswith(("at::","__amd")): continue
        agg[(n, r.get("Grid_Size", r.get("Grid_Size_X","")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (n,g),cs in sorted(agg.items()):
    print(f"{n:34s} grid {g:>9s} " + " ".join(f"{k}={sum(v)/len(v):.4g}" for k,v in sorted(cs.items())))
PY
