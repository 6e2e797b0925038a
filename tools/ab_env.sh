#!/bin/bash
# A/B of one environment switch on one box:  bash tools/ab_env.sh <tag> <VAR=value> <workload ...>
set -e -o pipefail
TAG=${1:?tag}; SW=${2:?VAR=value}; shift 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for w in "$@"; do
   for rep in 1 2; do
      python3 bench.py --workload $w --no-cpu-baseline --rk4-steps 0 > gpurun_out/${TAG}_ab_${w}_default_$rep.json 2>/dev/null
      env $SW python3 bench.py --workload $w --no-cpu-baseline --rk4-steps 0 > gpurun_out/${TAG}_ab_${w}_switch_$rep.json 2>/dev/null
   done
   python3 - <<PY
import json
f = lambda k, r: json.loads(open(f"gpurun_out/${TAG}_ab_${w}_{k}_{r}.json").read())["ms_per_step"]
print("[ab] $w default", [round(f("default", r), 4) for r in (1, 2)], "$SW", [round(f("switch", r), 4) for r in (1, 2)], flush=True)
PY
done
