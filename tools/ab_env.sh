#!/bin/bash
# A/B of library options on one box:  bash tools/ab_env.sh <tag> <Name=value[,Name=value]> <workload ...>   (RK4=0: RHS only)
# (options of omega_amd/csrc/Tuning.h, applied through omg_set_option by omega_amd/__init__.py from OMEGA_AMD_OPTIONS)
set -e -o pipefail
TAG=${1:?tag}; SW=${2:?Name=value}; shift 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RK=""; [ "${RK4:-1}" = 0 ] && RK="--rk4-steps 0"
for w in "$@"; do
   for rep in 1 2; do
      python3 bench.py --workload $w --no-cpu-baseline $RK > gpurun_out/${TAG}_ab_${w}_default_$rep.json 2>/dev/null
      OMEGA_AMD_OPTIONS=$SW python3 bench.py --workload $w --no-cpu-baseline $RK > gpurun_out/${TAG}_ab_${w}_switch_$rep.json 2>/dev/null
   done
   python3 - <<PY
import json
def f(k, r):
    d = json.loads(open(f"gpurun_out/${TAG}_ab_${w}_{k}_{r}.json").read())
    return round(d["ms_per_step"], 4), round((d.get("rk4") or {}).get("ms_per_step") or 0, 3)
print("[ab] $w default (RHS ms, RK4 ms)", [f("default", r) for r in (1, 2)], "$SW", [f("switch", r) for r in (1, 2)], flush=True)
PY
done
