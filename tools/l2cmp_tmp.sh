set -e
cd /root/repo
export TMPDIR=/tmp
for v in _r3 ""; do
  OMEGA_AMD_LIB=$PWD/omega_amd/lib/libomega_amd$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_l2cmp$v -o t -- python3 bench.py --workload fib7 --local-order curve --steps 8 --warmup 2 --rk4-steps 0 --no-cpu-baseline > gpurun_out/r04_l2cmp$v.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r04_l2cmp$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "Del2" in n or "L1PV" in n or "FinalTracer" in n:
        print("lib'$v'", n[:140].replace("OMEGA::",""), r["Calls"], r["AverageNs"])
PY
done
