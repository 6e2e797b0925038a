set -e
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/r01_v4_bench_qu30.json
cat gpurun_out/r01_v4_bench_qu30.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v4 -o v4 --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --rk4-steps 0 --no-cpu-baseline > $R/gpurun_out/prof_v4_bench.json 2>$R/gpurun_out/prof_v4.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch4 -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --rk4-steps 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc_hit4 -o h --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --rk4-steps 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_rk4 -o rk4 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --rk4-steps 4 --no-cpu-baseline > /dev/null 2>&1
ls $R/gpurun_out/prof_v4 $R/gpurun_out/pmc_fetch4 $R/gpurun_out/pmc_hit4 $R/gpurun_out/prof_rk4
