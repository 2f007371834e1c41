#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h9
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/h9/tests.txt
timeout 900 python tools/fuzz_fused.py 0 400 2>&1 | tail -2 | tee gpurun_out/h9/fuzz_fused.txt
timeout 600 python tools/fuzz_parity.py 0 150 2>&1 | tail -2 | tee gpurun_out/h9/fuzz_parity.txt
timeout 300 python tools/probes/blend_timeline.py S3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/h9/blend_timeline_S3.txt
