"""What bench.py's `value` can scale like: ONE process plays rank r of an N-way graph partition of the workload mesh (HaloWidth 4,
k-d order, per-rank state synthesis -- exactly bench.py's set-up) and times the plain fused RHS on that rank's local mesh
(owned + halo cells: an evaluation sweeps all of them and contains no exchange).  The N-GPU `value` is
cells x levels / max over ranks of this time; the slowest of the ranks played here bounds it from above.

   python tools/probes/rank_rhs.py [--parts 1,2,4,8] [--ranks 0,3] [--steps 20]
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
from omega_amd.meshgen import planar_hex, synthetic_state_rows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default="1,2,4,8")
    ap.add_argument("--ranks", default="0,3")
    ap.add_argument("--nx", type=int, default=680)
    ap.add_argument("--levels", type=int, default=80)
    ap.add_argument("--tracers", type=int, default=6)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--halo-width", type=int, default=4, help="HaloWidth of the N > 1 decompositions (bench.py: 4; the reference's default: 3)")
    a = ap.parse_args()
    K, NT = a.levels, a.tracers
    oa.device_init(0)
    g = planar_hex(a.nx, a.nx, 30e3)
    gm = oa.GlobalMesh(g)
    stream = oa.Stream()
    out = {"probe": "rank_rhs", "cells": int(g["nCells"]), "levels": K, "tracers": NT, "steps": a.steps, "parts": {}}
    t1 = None
    for n in [int(x) for x in a.parts.split(",")]:
        cell_task = oa.partition_cells(gm, n, "graph")[0] if n > 1 else None
        rec = {}
        for r in sorted({min(int(x), n - 1) for x in a.ranks.split(",")}):
            d = oa.Decomp(gm, n, r, a.halo_width if n > 1 else 3, cell_task=cell_task, local_order="kd")
            mesh = oa.HorzMesh(d, K)
            cells0 = d.get_array("CellID")[: mesh.NCellsAll] - 1
            edges0 = d.get_array("EdgeID")[: mesh.NEdgesAll] - 1
            hh, uu, tt = synthetic_state_rows(g, K, NT, cells0, edges0)
            h, u = np.zeros((mesh.NCellsSize, K)), np.zeros((mesh.NEdgesSize, K))
            tr = np.zeros((NT, mesh.NCellsSize, K))
            h[: mesh.NCellsAll], u[: mesh.NEdgesAll], tr[:, : mesh.NCellsAll] = hh, uu, tt
            state, tracers = oa.OceanState(mesh, None, K, 2), oa.Tracers(mesh, None, K, NT, 2)
            aux, tend = oa.AuxiliaryState(mesh, None, K, NT), oa.Tendencies(mesh, K, NT, oa.default_config())
            state.copy_to_device(h, u, 0)
            tracers.copy_to_device(tr, 0)
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) < 0.06:      # settle, as bench.py does
                for _ in range(4):
                    tend.compute_all_tendencies(state, aux, tracers, stream=stream)
                oa.device_synchronize()
            best = []
            for _ in range(3):
                for _ in range(3):
                    tend.compute_all_tendencies(state, aux, tracers, stream=stream)
                oa.device_synchronize()
                e0, e1 = oa.Event(), oa.Event()
                e0.record(stream)
                for _ in range(a.steps):
                    tend.compute_all_tendencies(state, aux, tracers, stream=stream)
                e1.record(stream)
                oa.device_synchronize()
                best.append(e0.elapsed_ms(e1) / a.steps)
            rec[str(r)] = {"owned_cells": mesh.NCellsOwned, "local_cells": mesh.NCellsAll, "rhs_ms": round(min(best), 4),
                           "irregular_edges": mesh.get_int("NIrregularEdges")}
            del state, tracers, aux, tend, mesh, d
        worst = max(v["rhs_ms"] for v in rec.values())
        if n == 1:
            t1 = worst
        rec["slowest_rank_ms"] = worst
        rec["value_scaling_bound"] = round(t1 / worst, 3) if t1 else None
        out["parts"][str(n)] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
