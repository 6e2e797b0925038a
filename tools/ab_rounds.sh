#!/bin/bash
# The round-2, round-3 and current libraries against each other on ONE box, alternating, same bench.py, RHS + RK4
# (omega_amd/lib/libomega_amd_r2.so / _r3.so are builds of the commits 8560ec1 / c2a4657 made with `git worktree`):
#   bash tools/ab_rounds.sh <tag> <workload> [reps]
# The old libraries know the Morton ("curve") local order only; the current one runs both that and its default (k-d).
set -e -o pipefail
TAG=${1:?tag}; W=${2:?workload}; REPS=${3:-3}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_rounds_$W.jsonl
: > $OUT
one() { # label lib-suffix extra-args
   OMEGA_AMD_LIB=$PWD/omega_amd/lib/libomega_amd$2.so python3 bench.py --workload $W --no-cpu-baseline $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r={'build':'$1','rhs_ms':round(d['ms_per_step'],4),'rhs_frac':d['roofline']['rhs']['frac'],'kernels_ms':d['roofline']['kernels_ms'],'rk4_ms':(d.get('rk4') or {}).get('ms_per_step'),'sypd':d.get('sypd')}
print(json.dumps(r))" | tee -a $OUT
}
for rep in $(seq $REPS); do
   one "r2 (8560ec1), curve order" _r2 "--local-order curve"
   one "r3 (c2a4657), curve order" _r3 "--local-order curve"
   one "r4, curve order" "" "--local-order curve"
   one "r4, k-d order (default)" "" ""
done
