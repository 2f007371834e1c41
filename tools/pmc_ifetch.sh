#!/bin/bash
# Extra PMC pass: instruction-fetch, branch, scalar-pipe and LDS-queue counters of the blend kernels (profiling aid).
# Usage: tools/pmc_ifetch.sh <outdir-name> [bench args...]
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VSKIPPED SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/$name/p$i -o c -- python $R/bench.py --steps 6 --warmup 2 --sustained-steps 0 --no-cpu-baseline --no-full-iteration --no-small "$@" > $R/gpurun_out/$name/p$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$R/gpurun_out/$name/p*/c_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].replace("scg::","").replace("void ","")[:34]
        if not n.startswith("blend"): continue
        agg[(n, r.get("Grid_Size", r.get("Grid_Size_X","")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (n,g),cs in sorted(agg.items()):
    print(f"{n:34s} grid {g:>9s} " + " ".join(f"{k}={sum(v)/len(v):.4g}" for k,v in sorted(cs.items())))
PY
