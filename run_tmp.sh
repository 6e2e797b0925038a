set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_v3.json
cat gpurun_out/bench_v3.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch3 -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --rk4-steps 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc_hit3 -o h --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --rk4-steps 0 --no-cpu-baseline > /dev/null 2>&1
ls $R/gpurun_out/pmc_fetch3 $R/gpurun_out/pmc_hit3
