#!/bin/bash
# blend_backward at seven waves per SIMD (72 registers) vs six; parity on the new build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h2
for w in S2 S4; do
timeout 600 python tools/ab_inproc.py --workload $w --libs ,w6,prev --reps 4 --steps 150 --warm 400 > gpurun_out/h2/ab_$w.txt 2>&1
tail -8 gpurun_out/h2/ab_$w.txt
done
timeout 300 python tools/probes/backward_timeline.py S2 > gpurun_out/h2/tl_S2.txt 2>&1; tail -28 gpurun_out/h2/tl_S2.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py -x -q -m gpu > gpurun_out/h2/tests.txt 2>&1
tail -3 gpurun_out/h2/tests.txt
