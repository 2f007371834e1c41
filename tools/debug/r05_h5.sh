#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/h5
python bench.py --workload S4 --no-s3 --no-full-iteration --no-cpu-baseline --no-small --no-rccl-floor > gpurun_out/h5/bench_S4.json 2>/dev/null
python tools/debug/r05_fulliter.py default 2>&1 | grep -v amdgpu.ids | tee gpurun_out/h5/fulliter.txt
python tools/debug/r05_fulliter.py fused 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/h5/fulliter.txt
bash tools/kstats.sh h5_ks_default python $R/tools/debug/r05_fulliter.py default | head -28 > gpurun_out/h5/kstats_default.txt
bash tools/kstats.sh h5_ks_fused python $R/tools/debug/r05_fulliter.py fused | head -28 > gpurun_out/h5/kstats_fused.txt
rm -f gpurun_out/h5_ks_*/k_kernel_trace.csv
cat gpurun_out/h5/kstats_default.txt
