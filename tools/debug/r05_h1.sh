#!/bin/bash
# dense fused blend with an 8-wave sort (SCG_EXP_DENSE8) vs the product; the new switch test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h1
timeout 900 python -m pytest tests/test_gpu_workloads.py -x -q -m gpu -k "every_switch or two_forwards or tile_cost" > gpurun_out/h1/tests.txt 2>&1
tail -3 gpurun_out/h1/tests.txt
SCG_LIB_PATH=$PWD/scgaussian_amd/libscg_raster_d8.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_call or fused or dense" > gpurun_out/h1/tests_d8.txt 2>&1
tail -3 gpurun_out/h1/tests_d8.txt
timeout 600 python tools/ab_inproc.py --workload S4 --libs ,d8 --reps 4 --steps 100 --warm 300 > gpurun_out/h1/ab_S4.txt 2>&1
tail -12 gpurun_out/h1/ab_S4.txt
timeout 600 python tools/ab_inproc.py --workload S4 --mode render --libs ,d8 --reps 4 --steps 100 --warm 300 > gpurun_out/h1/ab_S4r.txt 2>&1
tail -8 gpurun_out/h1/ab_S4r.txt
