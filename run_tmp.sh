( time timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -k randomised ) > gpurun_out/rand.log 2>&1; echo rc=$? >> gpurun_out/rand.log
