#!/bin/bash
# BASELINE cfg4: N independent jobs, ONE PROCESS PER GPU, no collectives ("LLFF all 8 scenes, one scene per GPU").
# The reference's host code addresses its GPU as "cuda:0" (utils/general_utils.py:139), so every process is given
# exactly one device through HIP_VISIBLE_DEVICES: inside the process that device is cuda:0.
#
#   tools/launch_cfg4.sh [-n NGPUS] [-o OUTDIR] -- <command> [args...]
#
# `{i}` in the command is replaced by the job index (e.g. a scene name list: -s scenes/{i}).  With no command the
# synthetic stand-in runs:  python bench.py --workload S2 (headline leg only)
# (one independent replica per GPU; aggregate throughput = sum of the per-GPU lines — "scaling": replicas only).
# Job i runs on GPU (i mod visible GPUs), so -n 8 on a 1-GPU box is a dry run of the launcher (8 processes time-share
# the one device).  Exit status: 0 when every job exited 0.
set -u
N=8
OUT=gpurun_out/cfg4
while [ $# -gt 0 ]; do
  case "$1" in
    -n) N=$2; shift 2 ;;
    -o) OUT=$2; shift 2 ;;
    --) shift; break ;;
    *) break ;;
  esac
done
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
NDEV=$(python3 - <<'PY'
import torch
print(max(torch.cuda.device_count(), 1))
PY
)
if [ $# -eq 0 ]; then
  set -- python "$ROOT/bench.py" --workload S2 --no-s3 --no-full-iteration --no-cpu-baseline --no-small --no-clustered --no-rccl-floor --steps 100 --warmup 20
fi
pids=()
for i in $(seq 0 $((N - 1))); do
  dev=$((i % NDEV))
  cmd=()
  for a in "$@"; do cmd+=("${a//\{i\}/$i}"); done
  ( export HIP_VISIBLE_DEVICES=$dev CUDA_VISIBLE_DEVICES= ROCR_VISIBLE_DEVICES= WORLD_SIZE=1 RANK=0 LOCAL_RANK=0
    unset CUDA_VISIBLE_DEVICES ROCR_VISIBLE_DEVICES
    export PYTHONPATH="$ROOT:${PYTHONPATH:-}"
    exec "${cmd[@]}" > "$OUT/job$i.out" 2> "$OUT/job$i.err" ) &
  pids+=($!)
done
rc=0
for i in "${!pids[@]}"; do
  if ! wait "${pids[$i]}"; then echo "job $i failed (see $OUT/job$i.err)"; rc=1; fi
done
python3 - "$OUT" "$N" <<'PY' || rc=1
import json, sys, glob, os
out, n = sys.argv[1], int(sys.argv[2])
tot, lines = 0.0, 0
for i in range(n):
    try:
        last = [l for l in open(os.path.join(out, f"job{i}.out")) if l.startswith("{")][-1]
        d = json.loads(last)
        tot += d["value"]; lines += 1
        print(f"job {i}: {d['value']:.1f} {d['unit']}  ({d['ms_per_step']} ms/step)")
    except Exception as e:
        print(f"job {i}: no bench line ({e})")
# the aggregate always says how many of the n jobs are IN it; fewer than n is a failed launch (exit status 1)
print(json.dumps({"cfg4_jobs": n, "jobs_reporting": lines, "aggregate_value": round(tot, 2), "scaling": "replicas only"}))
sys.exit(0 if lines == n else 1)
PY
exit $rc
