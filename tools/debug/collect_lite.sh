tag=r03t
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/gpu_tests.log
cp gpurun_out/tolerance_census.json $O/tolerance_census.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_S2_driver_args.json 2> $O/bench_S2_driver_args.err
python bench.py > $O/bench_S2.json 2> $O/bench_S2.err
python bench.py --workload S4 --no-s3 --no-full-iteration --no-cpu-baseline --no-small > $O/bench_S4.json 2>/dev/null
python bench.py --workload S1 --no-s3 --no-full-iteration --no-cpu-baseline --no-small > $O/bench_S1.json 2>/dev/null
python bench.py --workload S2r8 --no-s3 --no-full-iteration --no-cpu-baseline --no-small > $O/bench_S2r8.json 2>/dev/null
tools/kstats.sh $tag/kstats python $R/bench.py --steps 30 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-full-iteration --no-small > $O/kstats.txt 2>&1
tools/pmc.sh $tag/pmc > $O/pmc_counters.txt 2>&1
timeout 600 python tools/fuzz_parity.py 0 300 > $O/fuzz_parity.txt 2>&1
tail -2 $O/gpu_tests.log; tail -1 $O/fuzz_parity.txt; du -sh gpurun_out
