timeout -k 10 300 python -X faulthandler -m pytest tests/test_reductions.py -x -q > gpurun_out/red.log 2>&1; echo rc=$? >> gpurun_out/red.log
