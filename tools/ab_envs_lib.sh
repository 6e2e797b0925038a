#!/bin/bash
# option settings against the default on one box with a given library variant, RHS only, each setting twice (alternating):
#   bash tools/ab_envs_lib.sh <variant suffix> <workload> "<Name=v,...>" ...
set -e -o pipefail
V=${1?variant}; W=${2:?workload}; shift 2
cd "$(dirname "$0")/.."
export OMEGA_AMD_LIB=$PWD/omega_amd/lib/libomega_amd$V.so
run() { OMEGA_AMD_OPTIONS=$1 python3 bench.py --workload $W --no-cpu-baseline --rk4-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
for rep in 1 2; do
   echo "[ab] $W lib'$V' default: $(run "")"
   for s in "$@"; do echo "[ab] $W lib'$V' $s: $(run "$s")"; done
done
