set -x
R=$(pwd)
O=gpurun_out/r05a; mkdir -p $O
tools/kstats.sh r05a/kstats_S4 python $R/bench.py --workload S4 --steps 30 --warmup 5 --sustained-steps 0 --no-s3 --no-cpu-baseline --no-full-iteration --no-small --no-clustered --no-rccl-floor > $O/kstats_S4.txt 2>&1
python tools/trace_by_grid.py gpurun_out/r05a/kstats_S4/k_kernel_trace.csv > $O/kernels_by_grid_S4.txt 2>&1
cat $O/kernels_by_grid_S4.txt
tools/pmc.sh r05a/pmc_S4 --workload S4 --no-s3 > $O/pmc_S4.txt 2>&1
tail -30 $O/pmc_S4.txt
