for w in S2 S3 S4; do python tools/probes/blend_timeline.py $w 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05_blend_timeline.txt
