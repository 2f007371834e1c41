R=$(pwd); O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_loop.py tests/test_gpu_views.py -x -q 2>&1 | tail -4
for w in S2 S1; do python tools/ab_inproc.py --workload $w --switch EVENTLESS_WAIT=True,False --reps 6 2>&1 | grep -v amdgpu.ids; done | tee $O/ab_eventless.txt
tools/kstats.sh r05h/kstats python $R/bench.py --gpus 1 --steps 20 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-full-iteration --no-small --no-clustered --no-rccl-floor --no-s3 > $O/kstats.txt 2>&1
python tools/gap_table.py gpurun_out/r05h/kstats/k_kernel_trace.csv 774144 5:25 | tee $O/gap_table_S2.txt
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_S2_driver_args_$i.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_S2_driver_args_$i.json')); print(d['value'], d['ms_per_step'], d['stage_ms'], d.get('sustained',{}).get('ms_per_step'), d['small_workloads']['S1']['ms_per_step'], d['small_workloads']['S1'].get('ms_per_step_engine_direct'))"; done
