python -m pytest tests/test_gpu_parity.py -x -q -k "ico or fib or sphere" 2>&1 | tail -15
