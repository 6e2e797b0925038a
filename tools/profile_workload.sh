#!/bin/bash
# The full counter set of one workload, one box: bench line, kernel stats, HBM traffic (tools/profile_bench.sh), SQ
# wave-cycle counters (tools/pmc_diag.sh sq), texture-addresser / L1 counters (tools/pmc_diag.sh ta-tcp) ->
# gpurun_out/<tag>_{bench.json,kernel_stats.csv,pmc.json,sq_counters.txt,ta_tcp_counters.txt}
#   usage (through gpurun): bash tools/profile_workload.sh <tag> <workload> [more bench.py args]
set -e -o pipefail
TAG=${1:?tag}; W=${2:?workload}; shift 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/profile_bench.sh $TAG --workload $W "$@"
bash tools/pmc_diag.sh $TAG sq --workload $W "$@" > gpurun_out/${TAG}_sq_counters.txt
echo "[profile_workload] $W: SQ counters done"
bash tools/pmc_diag.sh $TAG ta-tcp --workload $W "$@" > gpurun_out/${TAG}_ta_tcp_counters.txt
echo "[profile_workload] $W: TA / TCP counters done"
