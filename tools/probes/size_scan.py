"""Per-kernel time of the fused RHS against the mesh size: T = a + b * cells per kernel (least squares) -- the fixed part
`a` is what a small per-GPU share pays for ramp and drain of every launch.

   python tools/probes/size_scan.py [--levels 80] [--tracers 6]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import omega_amd as oa  # noqa: E402
from omega_amd.meshgen import planar_hex, reorder_cells_morton, synthetic_state  # noqa: E402


def run(nx, ny, K, NT, steps):
    g = reorder_cells_morton(planar_hex(nx, ny, 30e3))
    gm = oa.GlobalMesh(g)
    decomp = oa.Decomp(gm, 1, 0, 3, local_order="curve")
    mesh = oa.HorzMesh(decomp, K)
    cell_id, edge_id = decomp.get_array("CellID"), decomp.get_array("EdgeID")
    hg, ug, trg = synthetic_state(g, K, NT)

    def to_local(glob, ids, rows):
        out = np.zeros(glob.shape[:-2] + (rows, glob.shape[-1]))
        out[..., : rows - 1, :] = glob[..., ids[: rows - 1] - 1, :]
        return out
    h, u, tr = to_local(hg, cell_id, mesh.NCellsSize), to_local(ug, edge_id, mesh.NEdgesSize), to_local(trg, cell_id, mesh.NCellsSize)
    state = oa.OceanState(mesh, None, K, 2)
    tracers = oa.Tracers(mesh, None, K, NT, 2)
    aux = oa.AuxiliaryState(mesh, None, K, NT)
    tend = oa.Tendencies(mesh, K, NT, oa.default_config())
    state.copy_to_device(h, u, 0)
    tracers.copy_to_device(tr, 0)
    stream = oa.Stream()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.08:     # settle (bench.py --settle-ms)
        for _ in range(4):
            tend.compute_all_tendencies(state, aux, tracers, stream=stream)
        oa.device_synchronize()
    ev0, ev1 = oa.Event(), oa.Event()
    ev0.record(stream)
    for _ in range(steps):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    ev1.record(stream)
    oa.device_synchronize()
    rhs = ev0.elapsed_ms(ev1) / steps
    tend.kernel_timing(True)
    for _ in range(steps):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    oa.device_synchronize()
    tend.kernel_timing(False)
    k = dict(tend.collect_kernel_times())
    return int(g["nCells"]), rhs, k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, default=80)
    ap.add_argument("--tracers", type=int, default=6)
    a = ap.parse_args()
    oa.device_init(0)
    rows = []
    for nx, ny in ((120, 120), (170, 170), (340, 170), (340, 340), (480, 480), (680, 680)):
        n, rhs, k = run(nx, ny, a.levels, a.tracers, 40 if nx * ny < 200000 else 20)
        rows.append({"cells": n, "rhs_ms": round(rhs, 4), "kernels_ms": {kk: round(v, 4) for kk, v in k.items()}})
    names = list(rows[0]["kernels_ms"])
    x = np.array([r["cells"] for r in rows], float)
    fit = {}
    for nm in names + ["rhs"]:
        y = np.array([r["rhs_ms"] if nm == "rhs" else r["kernels_ms"][nm] for r in rows])
        A = np.stack([np.ones_like(x), x], 1)
        (c0, c1), *_ = np.linalg.lstsq(A, y, rcond=None)
        fit[nm] = {"fixed_us": round(1e3 * c0, 1), "ns_per_cell": round(1e6 * c1, 3)}
    print(json.dumps({"probe": "size_scan", "levels": a.levels, "tracers": a.tracers, "rows": rows, "fit": fit}))


if __name__ == "__main__":
    main()
