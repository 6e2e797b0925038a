#!/bin/bash
# several option settings (omega_amd/csrc/Tuning.h) against the default on one box, RHS only:
#   bash tools/ab_envs.sh <tag> <workload> "<Name=v,Name=v>" ["<Name=v,...>" ...]
set -e -o pipefail
TAG=${1:?tag}; W=${2:?workload}; shift 2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { OMEGA_AMD_OPTIONS=$1 python3 bench.py --workload $W --no-cpu-baseline --rk4-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"; }
echo "[ab] $W default: $(run "")"
for s in "$@"; do echo "[ab] $W $s: $(run "$s")"; done
echo "[ab] $W default: $(run "")"
