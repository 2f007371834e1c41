for w in S2 S3; do python tools/probes/blend_timeline.py $w 2>&1 | grep -v amdgpu.ids | grep -v "^    "; done | tee gpurun_out/r05_blend_timeline2.txt
