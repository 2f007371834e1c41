O=gpurun_out/r05i; mkdir -p $O
for m in train render; do python tools/ab_inproc.py --workload S4 --mode $m --switch FUSED_SORT=True,False --reps 4 2>&1 | grep -v amdgpu.ids; done | tee $O/ab_S4_dense_fused.txt
python tools/scenario_sweep.py 2>&1 | grep -v amdgpu.ids | tail -12
