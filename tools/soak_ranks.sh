#!/bin/bash
# Soak of bench.py's own N > 1 launcher on ONE GPU (--single-device, peer wire): the same 4-rank run of a quarter of the
# QU30-sized mesh twelve times over; every run must pass its wire check, give overlapped == sequential and the same
# global state sums as the first (how round 3's zero-fill race showed: one run in five).  No retry: the first run that
# fails or differs ends the script non-zero.     usage (through gpurun): bash tools/soak_ranks.sh
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ref=""
for i in $(seq 1 12); do
  timeout -k 10 200 python bench.py --gpus 4 --single-device --workload qu30_quarter --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r05_soak_$i.json 2> gpurun_out/r05_soak_$i.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "[soak] run $i rc=$rc"; tail -5 gpurun_out/r05_soak_$i.err; exit $rc; fi
  sums=$(python -c "
import json; d=json.load(open('gpurun_out/r05_soak_$i.json')); oc=d['rk4']['overlap_check']; print(oc['overlapped_equals_sequential'], d['rk4']['error'], d['config']['halo_wire_check'][-16:], sum(d['rk4']['state_checksums_after_2_steps']))")
  echo "[soak] $i $sums"
  if [ -z "$ref" ]; then ref="$sums"; elif [ "$ref" != "$sums" ]; then echo "[soak] MISMATCH against run 1"; exit 9; fi
done
echo "[soak] 12 runs identical"
