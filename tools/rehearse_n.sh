#!/bin/bash
# bench.py with N ranks on ONE GPU (--single-device, PeerWire) under several option sets: the overlapped-vs-sequential verdict
#   usage (through gpurun): bash tools/rehearse_n.sh <tag> <N> <workload> "<opts>" ["<opts>" ...]     ("" = defaults)
set -o pipefail
TAG=${1:?tag}; N=${2:?ranks}; W=${3:?workload}; shift 3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
port=29550
for o in "$@"; do
   port=$((port + 1))
   name=$((port - 29550))_$(echo "${o:-default}" | tr ',=' '__')
   OMEGA_AMD_OPTIONS="$o" timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
      --master-port $port bench.py --gpus $N --steps 5 --warmup 2 --single-device --workload $W \
      > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
   python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_${name}.json").read())
    oc = (d["rk4"].get("overlap_check") or {})
    print("[rehearse] ${name}:", d["rk4"]["ms_per_step"], d["rk4"]["error"], oc.get("overlapped_equals_sequential"), oc.get("checksums_h_u_tracers", [None])[:3], flush=True)
    if "sequential_checksums_h_u_tracers" in oc:
        a, b = oc["checksums_h_u_tracers"], oc["sequential_checksums_h_u_tracers"]
        print("[rehearse]    differing fields (0 = h, 1 = u, 2.. = tracers):", [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y], flush=True)
except Exception as e:
    print("[rehearse] ${o:-default}: no record", e, flush=True)
PY
done
