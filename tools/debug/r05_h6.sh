#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/h6
timeout 900 python -m pytest tests/test_image_loss.py tests/test_match_loss.py tests/test_reference_fixtures.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/h6/tests.txt
python tools/debug/r05_fulliter.py default 2>&1 | grep -v amdgpu.ids | tee gpurun_out/h6/fulliter.txt
python tools/debug/r05_fulliter.py fused 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/h6/fulliter.txt
bash tools/kstats.sh h6_ks_fused python $R/tools/debug/r05_fulliter.py fused | head -24 > gpurun_out/h6/kstats_fused.txt
rm -f gpurun_out/h6_ks_*/k_kernel_trace.csv
cat gpurun_out/h6/kstats_fused.txt | cut -c1-120
