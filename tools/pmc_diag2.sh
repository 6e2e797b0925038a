#!/bin/bash
# Diagnostic PMC passes for the texture-addresser / L1 path of the RHS kernels: each pass in its own run with at most
# two counters of one block (more than a block has slots for aborts rocprofv3 with signal 6), bounded by timeout.  A
# pass that fails or times out ends the script with its log tail and a non-zero exit: nothing is swallowed.
#   usage (through gpurun): bash tools/pmc_diag2.sh <tag> [bench.py args]
set -o pipefail
TAG=${1:?tag}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
ARGS="--steps 4 --warmup 1 --rk4-steps 0 --no-cpu-baseline --no-live-traffic --realistic none $*"
pass() { # name counters...
   n=$1; shift
   timeout -k 10 100 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$n -o p -- python3 bench.py $ARGS > $OUT/${TAG}_$n.log 2>&1
   rc=$?
   echo "[diag2] pass $n rc=$rc"
   if [ $rc -ne 0 ]; then
      echo "[diag2] pass $n FAILED (rc $rc; 124 = timeout): last lines of its log" >&2
      tail -20 $OUT/${TAG}_$n.log >&2
      exit $rc     # no further GPU step after a failed one
   fi
}
# (at most two counters of one block per pass: more "exceeds the capabilities of the hardware" and rocprofv3 aborts)
pass ta1 TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp2 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections, sys
sys.path.insert(0, ".")
from tools.summarise_profile import short
for sub in ("ta1", "ta2", "tcp1", "tcp2"):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/${TAG}_%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(rows.items()):
        if "Body" not in k:
            continue
        print(sub, k[:70], {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
