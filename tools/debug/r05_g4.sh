for w in S2 S3 S4; do python tools/probes/scatter_timeline.py $w 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05_scatter_timeline.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "camera_is_known" 2>&1 | tail -5
