#!/bin/bash
# Diagnostic PMC passes for the RHS kernels, each in its own bench.py run under rocprofv3 (counters never share a run with
# the trace domains; a pass holds at most what one hardware block has slots for -- more aborts rocprofv3 with signal 6,
# "exceeds the capabilities of the hardware"), bounded by timeout.  A pass that fails or times out ends the script with
# its log tail and a non-zero exit: no further GPU step after a failed one.
#   sq      what the waves spend their cycles on (SQ_WAIT_ANY = parked at s_waitcnt, SQ_WAIT_INST_ANY = issue stall,
#           SQ_ACTIVE_INST_ANY = issuing; MI355X_MICROARCH.md "rocprofv3 PMC slots")
#   ta-tcp  texture-addresser / L1 path (four passes of two counters of one block)
#   usage (through gpurun): bash tools/pmc_diag.sh <tag> <sq|ta-tcp|all> [bench.py args]      (one script for the former
#   pmc_diag.sh + pmc_diag2.sh; a variant build is selected with OMEGA_AMD_LIB=.../libomega_amd_x.so in the environment)
set -o pipefail
TAG=${1:?tag}; MODE=${2:?sq|ta-tcp|all}; shift 2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ARGS="--steps 4 --warmup 1 --rk4-steps 0 --no-cpu-baseline --no-live-traffic --realistic none $*"
PASSES=""
pass() { # name counters...
   n=$1; shift
   timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${TAG}_$n -o p -- python3 bench.py $ARGS > $OUT/${TAG}_$n.log 2>&1
   rc=$?
   echo "[diag] pass $n rc=$rc"
   if [ $rc -ne 0 ]; then
      echo "[diag] pass $n FAILED (rc $rc; 124 = timeout): last lines of its log" >&2
      tail -20 $OUT/${TAG}_$n.log >&2
      exit $rc
   fi
   PASSES="$PASSES $n"
}
if [ "$MODE" = sq ] || [ "$MODE" = all ]; then
   pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE
fi
if [ "$MODE" = ta-tcp ] || [ "$MODE" = all ]; then
   pass ta1 TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE
   pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
   pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
   pass tcp2 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
fi
DIAG_OUT=$OUT DIAG_TAG=$TAG DIAG_PASSES="$PASSES" python3 - <<'PY'
import collections, csv, glob, os, sys
sys.path.insert(0, ".")
from tools.summarise_profile import short
out, tag = os.environ["DIAG_OUT"], os.environ["DIAG_TAG"]
for sub in os.environ["DIAG_PASSES"].split():
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{tag}_{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(rows.items()):
        if "Body" in k:
            print(sub, k[:70], {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
