#!/bin/bash
# several builds of the library against each other on one box, alternating, RHS + RK4:
#   bash tools/ab_libs.sh <tag> <workload> <variant suffix ...>     ("" = the default build)
set -e -o pipefail
TAG=${1:?tag}; W=${2:?workload}; shift 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
   for v in "$@"; do
      OMEGA_AMD_LIB=$PWD/omega_amd/lib/libomega_amd$v.so python3 bench.py --workload $W --no-cpu-baseline > gpurun_out/${TAG}_libs_${W}_x${v}_$rep.json 2>/dev/null
      python3 - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_libs_${W}_x${v}_$rep.json").read())
print("[ab] $W lib'$v'", round(d["ms_per_step"], 4), {a: round(b, 3) for a, b in d["roofline"]["kernels_ms"].items()}, round((d.get("rk4") or {}).get("ms_per_step") or 0, 3), flush=True)
PY
   done
done
