"""gpurun_out/<tag>_wl_<name>.json (tools/run_workloads.sh) -> one table, e.g. profiles/r03_workloads.json
   python tools/collect_workloads.py <tag> <out.json>"""
import glob
import json
import os
import sys


def main():
    tag, out = sys.argv[1], sys.argv[2]
    table = {}
    for f in sorted(glob.glob(f"gpurun_out/{tag}_wl_*.json")):
        name = os.path.basename(f)[len(tag) + 4:-5]
        try:
            d = json.loads(open(f).read())
        except ValueError:
            continue
        c, rk = d["config"], d.get("rk4") or {}
        table[name] = {"cells": c["cells"], "levels": c["levels"], "tracers": c["tracers"], "boundary_edges": c.get("boundary_edges", 0),
                       "rhs_ms": round(d["ms_per_step"], 4), "frac_of_8TBs_on_B_staged": d["roofline"]["rhs"]["frac"],
                       "rk4_ms": rk.get("ms_per_step"), "sypd": d.get("sypd"), "dt_s": rk.get("dt_s"),
                       "kernels_ms": d["roofline"]["kernels_ms"], "kernel_paths": c.get("kernel_paths"),
                       "steps": d["steps"], "warmup": d["warmup"], "untimed_settle_ms": (c.get("untimed_settle") or {}).get("ms")}
    with open(out, "w") as fh:
        json.dump(table, fh, indent=1)
    for k, v in table.items():
        print(k, v["rhs_ms"], v["frac_of_8TBs_on_B_staged"], v["rk4_ms"])


if __name__ == "__main__":
    main()
