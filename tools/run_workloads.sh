#!/bin/bash
# bench.py --workload <name> --no-cpu-baseline for every configuration size -> gpurun_out/<tag>_wl_<name>.json
#   usage (through gpurun):  bash tools/run_workloads.sh <tag> [workload ...]
set -e -o pipefail
TAG=${1:?tag}; shift || true
cd "$(dirname "$0")/.."
WL=${*:-qu30 ec30to60 qu240 qu30_eighth ico7 orrs18to6_eighth ico8}
mkdir -p gpurun_out
for w in $WL; do
   extra=""
   case "$w" in orrs18to6_eighth|fib7|fib7_coast) extra="--rk4-steps 0";; esac
   timeout -k 10 900 python3 bench.py --workload $w --no-cpu-baseline --no-live-traffic --realistic none $extra > gpurun_out/${TAG}_wl_$w.json 2> gpurun_out/${TAG}_wl_$w.err
   python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_wl_$w.json").read())
print("[workloads] $w", round(d["ms_per_step"], 4), "ms", round(d["roofline"]["rhs"]["frac"], 4), "rk4", (d.get("rk4") or {}).get("ms_per_step"), flush=True)
PY
done
