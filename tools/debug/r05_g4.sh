timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_workloads.py -x -q 2>&1 | tail -3
for w in S2 S3 S4; do python tools/ab_inproc.py --workload $w --mode render --libs prev, --reps 4 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05_ab_dynamic_chunks.txt
python tools/probes/geometry_timeline.py S4 2>&1 | grep -v amdgpu.ids | grep -v XCD | tail -9
