#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/h7
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/h7/bench_S2_driver_args.json 2> gpurun_out/h7/bench_S2_driver_args.err
python bench.py > gpurun_out/h7/bench_S2.json 2> gpurun_out/h7/bench_S2.err
python bench.py --workload S4 --no-s3 --no-full-iteration --no-cpu-baseline --no-small --no-rccl-floor > gpurun_out/h7/bench_S4.json 2>/dev/null
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > gpurun_out/h7/gpu_tests.txt
tail -2 gpurun_out/h7/gpu_tests.txt
