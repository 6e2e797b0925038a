#!/usr/bin/env python
"""Condenses the rocprofv3 outputs of tools/profile_bench.sh into the small files kept under profiles/:
<tag>_kernel_stats.csv (the --stats table) and <tag>_pmc.json (HBM bytes per launch and per kernel:
FETCH_SIZE doubled -- gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md "HBM" -- plus WRITE_SIZE,
which is exact; L2 hit rate).  The JSON records the hash of the kernel sources it was measured on; bench.py only
quotes `roofline.traffic` from a file whose hash matches the sources it runs."""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "omega_amd", "csrc", "kernels", "*")) +
                   [os.path.join(ROOT, "omega_amd", "csrc", "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _template_args(s):
    """top-level template arguments of the text after an opening '<'"""
    depth, cur, out = 0, "", []
    for ch in s:
        if ch == "<":
            depth += 1
        elif ch == ">":
            if depth == 0:
                break
            depth -= 1
        elif ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
            continue
        cur += ch
    out.append(cur.strip())
    return out


def short(name):
    """void OMEGA::tileKernel<OMEGA::FusedCell3Body<6, true, false>, ...>(...) -> FusedCell3Body<6, true, false>;
    tileKernel2<A, B, T> (two independent sweeps in one launch) -> A+B"""
    if "tileKernel2<" in name:
        a = _template_args(name[name.index("tileKernel2<") + len("tileKernel2<"):])
        return "+".join(x.replace("OMEGA::", "") for x in a[:2])
    if "tileKernelV<" in name:      # tileKernelV<T, A, B, ...>: any number of independent sweeps in one launch
        a = _template_args(name[name.index("tileKernelV<") + len("tileKernelV<"):])
        return "+".join(x.replace("OMEGA::", "") for x in a[1:])
    if "tileKernel<" in name:
        a = _template_args(name[name.index("tileKernel<") + len("tileKernel<"):])
        return a[0].replace("OMEGA::", "")
    return name.split("(")[0].replace("OMEGA::", "").replace("void ", "").strip()


def counters(dirname):
    rows = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return rows


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "gpurun_out")
    fetch, write = counters(os.path.join(out, tag + "_pmc_fetch")), counters(os.path.join(out, tag + "_pmc_write"))
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate runs) of "
                    "`python3 bench.py --steps 8 --warmup 2 --rk4-steps 2 --no-cpu-baseline " + " ".join(sys.argv[2:]) +
                    "`; FETCH_SIZE (KB) x 1024 x 2 (gfx950 half-count of 16-B-per-lane reads) + WRITE_SIZE (KB) x 1024",
           "kernel_source_sha": kernel_source_sha(), "bench_args": sys.argv[2:]}
    total = 0.0
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, {}).get("FETCH_SIZE", []), write.get(k, {}).get("WRITE_SIZE", [])
        if not f or not w or "Body" not in k and "Fn" not in k:
            continue
        fb, wb = 2.0 * 1024.0 * sum(f) / len(f), 1024.0 * sum(w) / len(w)
        hit, miss = sum(write[k].get("TCC_HIT_sum", [0])), sum(write[k].get("TCC_MISS_sum", [0]))
        res[k] = {"launches_sampled": len(f), "FETCH_SIZE_KB_raw_mean": sum(f) / len(f),
                  "fetch_bytes_per_launch_gfx950_corrected_x2": fb, "WRITE_SIZE_KB_mean": sum(w) / len(w),
                  "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb,
                  "L2_hit_rate": hit / (hit + miss) if hit + miss else None}
    with open(os.path.join(out, tag + "_pmc.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    stats = glob.glob(os.path.join(out, tag + "_trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, tag + "_kernel_stats.csv"))
    print("[profile] wrote", tag + "_pmc.json", "and", tag + "_kernel_stats.csv" if stats else "(no kernel stats found)")


if __name__ == "__main__":
    main()
