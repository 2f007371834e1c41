O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
timeout 600 python tools/fuzz_binning.py 0 60 2>&1 | tail -2
timeout 600 python tools/fuzz_fused.py 0 300 2>&1 | tail -2
for w in S2 S4; do python tools/ab_inproc.py --workload $w --libs r4, --reps 4 2>&1 | grep -v amdgpu.ids; done > $O/ab_train.txt; cat $O/ab_train.txt
python tools/ab_inproc.py --workload S3 --mode render --libs r4, --reps 4 2>&1 | grep -v amdgpu.ids > $O/ab_S3.txt; cat $O/ab_S3.txt
for w in S2 S3 S4; do python tools/ab_inproc.py --workload $w --mode render --switch FUSED_HIST=True,False --reps 4 2>&1 | grep -v amdgpu.ids; done > $O/ab_fused_hist.txt; cat $O/ab_fused_hist.txt
