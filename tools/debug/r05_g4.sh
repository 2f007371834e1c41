timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_fused.py 0 300 2>&1 | tail -1
timeout 600 python tools/fuzz_parity.py 0 200 2>&1 | tail -2
