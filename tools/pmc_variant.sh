#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export OMEGA_AMD_LIB=$PWD/omega_amd/lib/libomega_amd_x.so
ARGS="--steps 8 --warmup 2 --rk4-steps 0 --no-cpu-baseline --no-live-traffic --realistic none --workload qu30"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04ac_fetch -o f -- python3 bench.py $ARGS > gpurun_out/r04ac_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r04ac_write -o w -- python3 bench.py $ARGS > gpurun_out/r04ac_write.log 2>&1
python3 - <<'PY'
import csv, glob, collections
def agg(pat, col):
    f = glob.glob(pat, recursive=True)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == col:
            d[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}
fe = agg("gpurun_out/r04ac_fetch/**/*counter_collection.csv", "FETCH_SIZE")
wr = agg("gpurun_out/r04ac_write/**/*counter_collection.csv", "WRITE_SIZE")
hit = agg("gpurun_out/r04ac_write/**/*counter_collection.csv", "TCC_HIT_sum")
mis = agg("gpurun_out/r04ac_write/**/*counter_collection.csv", "TCC_MISS_sum")
for k in fe:
    if "Body" in k:
        print("[pmc]", k, "fetch GB", round(fe[k] * 1024 * 2 / 1e9, 3), "write GB", round(wr.get(k, 0) * 1024 / 1e9, 3), "L2 hit", round(hit.get(k, 0) / max(hit.get(k, 0) + mis.get(k, 0), 1), 3))
PY
