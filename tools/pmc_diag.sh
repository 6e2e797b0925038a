#!/bin/bash
# Diagnostic PMC pass for the RHS kernels: what the waves spend their cycles on (SQ_WAIT_ANY = parked at s_waitcnt,
# SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   usage (through gpurun): bash tools/pmc_diag.sh <tag> [bench.py args]
set -e -o pipefail
TAG=${1:?tag}; shift || true
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
ARGS="--steps 4 --warmup 1 --rk4-steps 0 --no-cpu-baseline --no-live-traffic --realistic none $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_sq -o s -- python3 bench.py $ARGS > $OUT/${TAG}_sq.log 2>&1
echo "[diag] SQ pass done"
# (TCP / TA counters are collected by tools/pmc_diag2.sh, two counters of one block per pass: asking for more than the
# block has hardware slots for makes rocprofv3 abort with signal 6 -- "exceeds the capabilities of the hardware")
python3 - <<PY
import csv, glob, collections, sys
sys.path.insert(0, ".")
from tools.summarise_profile import short
for sub in ("sq",):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/${TAG}_%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(rows.items()):
        if "Body" not in k:
            continue
        print(sub, k[:70], {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
