#!/bin/bash
# forward blend walking its sorted ids from LDS (product) vs the previous commit; 1 024-entry variant (nine workgroups per CU by LDS again)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h8
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_call or sorts_its_own or fallback or odd_shapes or twin or skipped" 2>&1 | tail -3 | tee gpurun_out/h8/tests.txt
for w in S2 S3; do
timeout 900 python tools/ab_inproc.py --workload $w --mode render --libs ,prev,m1k --reps 4 --steps 150 --warm 400 > gpurun_out/h8/ab_${w}r.txt 2>&1
tail -7 gpurun_out/h8/ab_${w}r.txt
done
timeout 900 python tools/ab_inproc.py --workload S2 --libs ,prev,m1k --reps 4 --steps 150 --warm 400 > gpurun_out/h8/ab_S2.txt 2>&1
tail -7 gpurun_out/h8/ab_S2.txt
timeout 900 python tools/ab_inproc.py --workload S4 --libs ,prev --reps 3 --steps 100 --warm 300 > gpurun_out/h8/ab_S4.txt 2>&1
tail -5 gpurun_out/h8/ab_S4.txt
