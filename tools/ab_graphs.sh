#!/bin/bash
cd "$(dirname "$0")/.."
run() { OMEGA_AMD_OPTIONS=$2 python3 bench.py --workload $1 --no-cpu-baseline --steps 200 --warmup 20 --rk4-steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['rk4']['ms_per_step'],4), d['config'].get('hip_graph'))"; }
for w in qu240 ico5 qu30_eighth; do
  for rep in 1 2; do
    echo "[graphs] $w default: $(run $w "")"
    echo "[graphs] $w Graphs=1: $(run $w "Graphs=1")"
  done
done
