timeout 1200 python -m pytest tests/test_gpu_workloads.py tests/test_gpu_parity.py tests/test_gpu_views.py tests/test_gpu_train_loop.py -x -q 2>&1 | tail -4
python tools/probes/geometry_timeline.py S3 2>&1 | grep -v amdgpu.ids | tail -11
