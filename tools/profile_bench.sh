#!/bin/bash
# Collects the measurement artefacts behind bench.py's roofline numbers on the GPU box:
#   1. the bench line itself,
#   2. rocprofv3 --kernel-trace --stats (per-kernel average durations),
#   3. two PMC passes (FETCH_SIZE; WRITE_SIZE + L2 hit/miss) in their own runs -- counters never share a run with
#      the trace domains, and FETCH_SIZE / WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# and condenses them into gpurun_out/<tag>_{bench.json,kernel_stats.csv,pmc.json}; copy those into profiles/.
#   usage (through gpurun):  bash tools/profile_bench.sh <tag> [bench.py arguments]
set -e -o pipefail
TAG=${1:?tag}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
# (the profiled runs measure the named workload only: no realistic block, whose kernels would enter the per-kernel averages)
ARGS="--steps 8 --warmup 2 --rk4-steps 2 --no-cpu-baseline --no-live-traffic --realistic none $*"
python3 bench.py --steps 20 --warmup 3 $* > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "[profile] bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 bench.py $ARGS > $OUT/${TAG}_trace.log 2>&1
echo "[profile] kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python3 bench.py $ARGS > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "[profile] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python3 bench.py $ARGS > $OUT/${TAG}_pmc_write.log 2>&1
echo "[profile] WRITE_SIZE pass done"
python3 tools/summarise_profile.py $TAG $*
