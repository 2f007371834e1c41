#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h10
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_call or sorts_its_own or fallback or odd_shapes or twin or skipped" 2>&1 | tail -2 | tee gpurun_out/h10/tests.txt
timeout 600 python tools/fuzz_fused.py 0 300 2>&1 | tail -1 | tee gpurun_out/h10/fuzz_fused.txt
for w in S3 S2; do timeout 600 python tools/ab_inproc.py --workload $w --mode render --reps 3 --steps 150 --warm 400 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a gpurun_out/h10/ab.txt; done
timeout 600 python tools/ab_inproc.py --workload S4 --reps 3 --steps 100 --warm 300 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a gpurun_out/h10/ab.txt
