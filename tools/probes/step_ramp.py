"""Per-evaluation device times right after a synchronisation: how long the first evaluations of a timed region take
compared with the steady state (clock ramp / wake-up after the idle the barrier causes).

   python tools/probes/step_ramp.py [--nx 240] [--steps 40] [--idle-ms 0]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=240)
    ap.add_argument("--levels", type=int, default=80)
    ap.add_argument("--tracers", type=int, default=6)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    K, NT = a.levels, a.tracers
    oa.device_init(0)
    g = reorder_cells_morton(planar_hex(a.nx, a.nx, 30e3))
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
    out = {}
    for idle_ms in (0, 1, 10, 100):
        for _ in range(a.warmup):
            tend.compute_all_tendencies(state, aux, tracers, stream=stream)
        oa.device_synchronize()
        time.sleep(idle_ms * 1e-3)
        evs = [oa.Event() for _ in range(a.steps + 1)]
        t0 = time.perf_counter()
        evs[0].record(stream)
        for i in range(a.steps):
            tend.compute_all_tendencies(state, aux, tracers, stream=stream)
            evs[i + 1].record(stream)
        t_host = time.perf_counter() - t0
        oa.device_synchronize()
        wall = time.perf_counter() - t0
        per = [round(evs[i].elapsed_ms(evs[i + 1]), 4) for i in range(a.steps)]
        out[f"idle_{idle_ms}ms"] = {"per_step_ms": per, "host_enqueue_ms_total": round(1e3 * t_host, 3), "wall_ms_total": round(1e3 * wall, 3),
                                    "sum_ms": round(sum(per), 3)}
    print(json.dumps({"probe": "step_ramp", "cells": int(g["nCells"]), "levels": K, "tracers": NT, "runs": out}))


if __name__ == "__main__":
    main()
