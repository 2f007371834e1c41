O=gpurun_out/r05c; mkdir -p $O
for w in S2 S3 S4; do timeout 600 python tools/ab_inproc.py --workload $w --mode render --libs ,g1,g2,g3,s1,s2,s3 --reps 2 --steps 100 --warm 200 2>&1 | grep -v amdgpu.ids; done > $O/probes_geo_scatter.txt; cat $O/probes_geo_scatter.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "camera_is_known or speculative or one_call or skipped_rare" 2>&1 | tail -5
