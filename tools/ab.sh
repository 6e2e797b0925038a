#!/bin/bash
# Alternating A/B on ONE box (through gpurun): the same bench.py workload under several VARIANTS, `reps` rounds of
# all variants in turn (array placement moves a kernel by up to 8 % between processes: only alternating runs on one
# box compare builds or options -- DESIGN.md section 5).  One script for what used to be ten (ab_args / ab_env /
# ab_envs / ab_envs2 / ab_envs_lib / ab_graphs / ab_kernel_stats / ab_lib / ab_libs / ab_rounds).
#
#   bash tools/ab.sh <tag> <workload> [--reps N] [--rhs-only] [--kernel-stats] [--args "<bench args for all>"] -- <variant> ...
#
# A variant is "key=value;key=value" with the keys
#   name=<label>            (default: the spec itself)
#   lib=<suffix>            omega_amd/lib/libomega_amd<suffix>.so  (make -C omega_amd/csrc VARIANT=<suffix> EXTRA=-D...)
#   opt=<Name=v,Name=v>     library options (omega_amd/csrc/Tuning.h, applied through omg_set_option)
#   args=<bench.py args>    e.g. args=--local-order curve
# and "" or "default" = the default build, options and arguments.  Examples:
#   bash tools/ab.sh r05 qu30 -- default "opt=TracerPatch=0"
#   bash tools/ab.sh r05 ico7 --reps 3 -- default "lib=_x" "lib=_x;opt=TracerPatch=0"
#   bash tools/ab.sh r05 qu30 -- "lib=_r2;args=--local-order curve" "lib=_r3;args=--local-order curve" default
#   bash tools/ab.sh r05 qu240 --args "--steps 200 --warmup 20 --rk4-steps 100" -- default "opt=Graphs=1"
# --kernel-stats runs each variant once under rocprofv3 --kernel-trace --stats instead and prints the per-kernel
# averages side by side.  Output: one JSON line per run in gpurun_out/<tag>_ab_<workload>.jsonl, echoed.
set -e -o pipefail
TAG=${1:?tag}; W=${2:?workload}; shift 2
REPS=2; RHS=""; KS=0; COMMON=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
   case "$1" in
      --reps) REPS=$2; shift 2 ;;
      --rhs-only) RHS="--rk4-steps 0"; shift ;;
      --kernel-stats) KS=1; shift ;;
      --args) COMMON=$2; shift 2 ;;
      *) echo "unknown option $1" >&2; exit 2 ;;
   esac
done
[ "$1" = "--" ] && shift
[ $# -gt 0 ] || { echo "no variants" >&2; exit 2; }
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_ab_$W.jsonl
: > $OUT
field() { # spec key
   echo "$1" | tr ';' '\n' | sed -n "s/^$2=//p" | head -1
}
one() { # spec
   local spec=$1; [ "$spec" = default ] && spec=""
   local name lib opt args
   name=$(field "$spec" name); lib=$(field "$spec" lib); opt=$(field "$spec" opt); args=$(field "$spec" args)
   [ -n "$name" ] || name=${spec:-default}
   local libpath=$PWD/omega_amd/lib/libomega_amd$lib.so
   if [ $KS = 1 ]; then
      local d=gpurun_out/${TAG}_ks_$(echo "$name" | tr -c 'A-Za-z0-9_' _)
      OMEGA_AMD_LIB=$libpath OMEGA_AMD_OPTIONS=$opt rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- \
         python3 bench.py --workload $W --steps 8 --warmup 2 --rk4-steps 0 --no-cpu-baseline --no-live-traffic --realistic none $COMMON $args > $d.log 2>&1
      AB_NAME="$name" AB_DIR="$d" python3 - <<'PY' | tee -a $OUT
import csv, glob, json, os
f = glob.glob(os.environ["AB_DIR"] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = {r["Name"].replace("OMEGA::", "")[:160]: (int(r["Calls"]), float(r["AverageNs"]) / 1e6) for r in csv.DictReader(open(f)) if "Body" in r["Name"]}
print(json.dumps({"variant": os.environ["AB_NAME"], "kernel_avg_ms": {k: round(v[1], 4) for k, v in rows.items()}, "calls": {k: v[0] for k, v in rows.items()}}))
PY
   else
      OMEGA_AMD_LIB=$libpath OMEGA_AMD_OPTIONS=$opt python3 bench.py --workload $W --no-cpu-baseline --no-live-traffic --realistic none $RHS $COMMON $args 2>/dev/null |
         AB_NAME="$name" python3 -c "
import json, os, sys
d = json.loads(sys.stdin.read())
print(json.dumps({'variant': os.environ['AB_NAME'], 'rhs_ms': round(d['ms_per_step'], 4), 'rhs_frac': d['roofline']['rhs']['frac'],
                  'kernels_ms': d['roofline']['kernels_ms'], 'rk4_ms': (d.get('rk4') or {}).get('ms_per_step'), 'sypd': d.get('sypd'),
                  'hip_graph': d['config'].get('hip_graph')}))" | tee -a $OUT
   fi
}
[ $KS = 1 ] && REPS=1
for rep in $(seq $REPS); do
   for v in "$@"; do one "$v"; done
done
