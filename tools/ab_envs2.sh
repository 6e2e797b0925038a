#!/bin/bash
# option settings against the default on one box, RHS only, every setting twice (alternating):
#   bash tools/ab_envs2.sh <workload> "<Name=v,...>" ...
set -e -o pipefail
W=${1:?workload}; shift
cd "$(dirname "$0")/.."
run() { OMEGA_AMD_OPTIONS=$1 python3 bench.py --workload $W --no-cpu-baseline --rk4-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['rhs']['frac'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
for rep in 1 2; do
   echo "[ab] $W default: $(run "")"
   for s in "$@"; do echo "[ab] $W $s: $(run "$s")"; done
done
