#!/bin/bash
# A/B of a bench.py argument on one box:  bash tools/ab_args.sh <tag> "<extra args>" <workload ...>
set -e -o pipefail
TAG=${1:?tag}; EX=${2:?args}; shift 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for w in "$@"; do
   for rep in 1 2; do
      python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/${TAG}_arg_${w}_default_$rep.json 2>/dev/null
      python3 bench.py --workload $w --no-cpu-baseline $EX > gpurun_out/${TAG}_arg_${w}_switch_$rep.json 2>/dev/null
   done
   python3 - <<PY
import json
def f(k, r):
    d = json.loads(open(f"gpurun_out/${TAG}_arg_${w}_{k}_{r}.json").read())
    return round(d["ms_per_step"], 4), round((d.get("rk4") or {}).get("ms_per_step") or 0, 3)
print("[ab] $w default (RHS ms, RK4 ms)", [f("default", r) for r in (1, 2)], "$EX", [f("switch", r) for r in (1, 2)], flush=True)
PY
done
