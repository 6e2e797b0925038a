#!/bin/bash
# rocprofv3 --kernel-trace --stats of the same bench.py run under several builds of the library (omega_amd/lib/libomega_amd<suffix>.so;
# "" = the default build): the per-kernel averages side by side.  How round 4 found that one launch of the narrow-table path had
# become 23 % slower although the RHS as a whole was faster.
#   usage (through gpurun): bash tools/ab_kernel_stats.sh <tag> <workload> "<bench args>" <suffix> [<suffix> ...]
set -e -o pipefail
TAG=${1:?tag}; W=${2:?workload}; ARGS=$3; shift 3
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "$@"; do
  OMEGA_AMD_LIB=$PWD/omega_amd/lib/libomega_amd$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_ks$v -o t -- \
     python3 bench.py --workload $W --steps 8 --warmup 2 --rk4-steps 0 --no-cpu-baseline $ARGS > gpurun_out/${TAG}_ks$v.log 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/${TAG}_ks$v/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "Body" in n:
        print("lib'$v'", n[:150].replace("OMEGA::", ""), r["Calls"], r["AverageNs"])
PY
done
