#!/bin/bash
# One GPU-box pass that produces everything profiles/ keeps for a round.  Usage: tools/collect_round.sh <tag>
# (run through gpurun from the repository root; results land in gpurun_out/<tag>/)
tag=${1:-rXX}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/gpu_tests.log
cp gpurun_out/tolerance_census.json $O/tolerance_census.json 2>/dev/null
# counters first: bench.py prints them only when profiles/pmc_summary.json carries the stamp of the kernels it runs
tools/pmc.sh $tag/pmc > $O/pmc_counters.txt 2>&1
tools/pmc.sh $tag/pmc_S4 --workload S4 --no-s3 > $O/pmc_counters_S4.txt 2>&1
for d in 0 1 2; do tools/pmc.sh $tag/pmc_deg$d --sh-degree $d --no-s3 > $O/pmc_counters_deg$d.txt 2>&1; done
python tools/pmc_summary.py gpurun_out/$tag/pmc,gpurun_out/$tag/pmc_S4 gpurun_out/$tag/pmc_deg0@deg0 gpurun_out/$tag/pmc_deg1@deg1 gpurun_out/$tag/pmc_deg2@deg2 --out profiles/pmc_summary.json > /dev/null 2>&1
cp profiles/pmc_summary.json $O/pmc_summary.json      # (profiles/ does not travel back: copy it from here, or re-run pmc_summary.py at home)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_S2_driver_args.json 2> $O/bench_S2_driver_args.err
python bench.py > $O/bench_S2.json 2> $O/bench_S2.err
python bench.py --workload S4 --no-s3 --no-full-iteration --no-cpu-baseline --no-small --no-rccl-floor > $O/bench_S4.json 2>/dev/null
python bench.py --workload S1 --no-s3 --no-full-iteration --no-cpu-baseline --no-small --no-rccl-floor > $O/bench_S1.json 2>/dev/null
python bench.py --workload S2r8 --no-s3 --no-full-iteration --no-cpu-baseline --no-small --no-rccl-floor > $O/bench_S2r8.json 2>/dev/null
tools/kstats.sh $tag/kstats python $R/bench.py --steps 30 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-full-iteration --no-small --no-clustered --no-rccl-floor --no-by-degree --no-render-glue --no-graph --no-moving-scene > $O/kstats.txt 2>&1
python tools/trace_by_grid.py gpurun_out/$tag/kstats/k_kernel_trace.csv > $O/kernels_by_grid.txt 2>&1
# idle gaps between the launches of a step, inside the driver's timed region (its command: steps 5..24 of the trace)
tools/kstats.sh $tag/kstats_driver python $R/bench.py --gpus 1 --steps 20 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-full-iteration --no-small --no-clustered --no-rccl-floor --no-by-degree --no-render-glue --no-graph --no-moving-scene --no-s3 > /dev/null 2>&1
python tools/gap_table.py gpurun_out/$tag/kstats_driver/k_kernel_trace.csv 774144 5:25 > $O/gap_table_S2.txt 2>&1
# cfg5's per-view shape (S4): kernel stats and counters of its own
tools/kstats.sh $tag/kstats_S4 python $R/bench.py --workload S4 --steps 30 --warmup 5 --sustained-steps 0 --no-s3 --no-cpu-baseline --no-full-iteration --no-small --no-clustered --no-rccl-floor --no-by-degree --no-render-glue --no-graph --no-moving-scene > $O/kstats_S4.txt 2>&1
python tools/trace_by_grid.py gpurun_out/$tag/kstats_S4/k_kernel_trace.csv > $O/kernels_by_grid_S4.txt 2>&1
# per-wave timelines of the kernels (probe build, tools/probes/*_timeline.py)
python -m scgaussian_amd.build --tag=tl -DSCG_PROBE_TIMELINE > /dev/null 2>&1
(for w in S2 S3 S4; do python tools/probes/geometry_timeline.py $w; python tools/probes/scatter_timeline.py $w; python tools/probes/blend_timeline.py $w; done; for w in S2 S4; do python tools/probes/backward_timeline.py $w; done) 2>&1 | grep -v amdgpu.ids > $O/kernel_timelines.txt
for w in S2 S4; do python tools/ab_inproc.py --workload $w --switch BWD_ORDER_HINT=True,False --reps 6 2>&1 | grep -v amdgpu.ids; done > $O/ab_bwd_order_hint.txt
for w in S2 S1; do python tools/ab_inproc.py --workload $w --switch EVENTLESS_WAIT=True,False --reps 6 2>&1 | grep -v amdgpu.ids; done > $O/ab_eventless_wait.txt
# (bench.py starts its own two ranks: no torchrun)
timeout 300 python bench.py --gpus 2 --workload S4 --steps 20 --warmup 5 --dist-backend gloo > $O/bench_gpus2_gloo_one_gpu_S4.json 2> $O/bench_gpus2.err
timeout 400 tools/launch_cfg4.sh -n 4 -o $O/cfg4 > $O/cfg4.log 2>&1
[ -x tools/probes/clock_probe ] || hipcc --offload-arch=gfx950 -O3 -o tools/probes/clock_probe tools/probes/clock_probe.hip 2>/dev/null
./tools/probes/clock_probe > $O/clock_probe.txt 2>&1
python tools/host_profile.py S1 > $O/host_profile_S1.txt 2>&1
python tools/host_split.py S1 > $O/host_split_S1.txt 2>&1
SCG_AUTOGRAD_SINGLE_THREAD=1 python tools/host_profile.py S1 > $O/host_profile_S1_single_thread.txt 2>&1
timeout 900 python tools/fuzz_parity.py 0 ${FUZZ_N:-600} > $O/fuzz_parity.txt 2>&1
timeout 600 python tools/fuzz_binning.py 0 ${FUZZ_B:-100} > $O/fuzz_binning.txt 2>&1
timeout 900 python tools/fuzz_fused.py 0 ${FUZZ_F:-1000} > $O/fuzz_fused.txt 2>&1
if [ -n "${COLLECT_PROBES:-}" ]; then
# what bounds the blend kernels: instruction-supply probe, fetch / branch / scalar counters, 
# trips a finer cull would save, hand-written vs compiler-written forward trip on the same box
[ -x tools/probes/ifetch_probe ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/probes/ifetch_probe tools/probes/ifetch_probe.hip 2>/dev/null
./tools/probes/ifetch_probe > $O/ifetch_probe.txt 2>&1
tools/pmc_ifetch.sh $tag/pmc_ifetch 2>&1 | grep -v amdgpu.ids > $O/pmc_ifetch.txt
for w in S2 S3 S4; do python tools/probes/cull_granularity.py $w 2>/dev/null | tail -1; done > $O/cull_granularity.txt
python -m scgaussian_amd.build --tag=cxx -DSCG_FWD_TRIP_CXX > /dev/null 2>&1
(for i in 1 2; do ABLATE_ARGS="--no-small" tools/ablate.sh cxx 2>&1 | grep -v amdgpu.ids; done) > $O/ab_forward_trip.txt
fi
for w in clustered30 clustered60; do python tools/ab_inproc.py --workload $w --switch SPLIT_LONG_LISTS=True,False --reps 4 --steps 100 --warm 300 2>&1 | grep -v amdgpu.ids; done > $O/ab_split_long_lists.txt
python tools/ab_inproc.py --workload S2 --switch SKIP_IDLE_RARE_SORT=True,False --reps 6 2>&1 | grep -v amdgpu.ids > $O/ab_skip_rare_sort.txt
tail -2 $O/gpu_tests.log; tail -2 $O/fuzz_parity.txt; tail -1 $O/fuzz_binning.txt
