for w in S2 S3 S4; do python tools/ab_inproc.py --workload $w --mode render --libs r4,coal, --reps 4 --steps 100 --warm 300 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05_coal_probe.txt
