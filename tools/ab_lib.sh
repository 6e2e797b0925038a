#!/bin/bash
# A/B of two builds of the library on one box:  bash tools/ab_lib.sh <tag> <variant suffix> <workload ...>
#   (make -C omega_amd/csrc VARIANT=_x EXTRA=-D... builds omega_amd/lib/libomega_amd_x.so next to the default one)
set -e -o pipefail
TAG=${1:?tag}; VAR=${2:?variant}; shift 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for w in "$@"; do
   for rep in 1 2; do
      python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/${TAG}_lib_${w}_default_$rep.json 2>/dev/null
      OMEGA_AMD_LIB=$PWD/omega_amd/lib/libomega_amd$VAR.so python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/${TAG}_lib_${w}_variant_$rep.json 2>/dev/null
   done
   python3 - <<PY
import json
def f(k, r):
    d = json.loads(open(f"gpurun_out/${TAG}_lib_${w}_{k}_{r}.json").read())
    return round(d["ms_per_step"], 4), {a: round(b, 3) for a, b in d["roofline"]["kernels_ms"].items()}, round((d.get("rk4") or {}).get("ms_per_step") or 0, 3)
for k in ("default", "variant"):
    for r in (1, 2):
        print("[ab] $w", k, *f(k, r), flush=True)
PY
done
