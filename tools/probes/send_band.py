"""What one rank of an N-way run saves when the overlapped RK4 stages skip the halo cells whose results the exchange
replaces (option SendBand, MeshView::BandSendCells).

ONE process plays one rank of an N-part decomposition of the workload mesh; the wire is replaced by a function that
moves nothing (the halo then keeps whatever the receive buffers hold -- the STATE of this probe is meaningless, its
kernel sequence and sizes are exactly the rank's).  RK4 steps are timed with the options SendBand, BandOnComm and ShrinkSweeps on and off, alternating.

   python tools/probes/send_band.py [--parts 8] [--rank 0] [--nx 680] [--levels 80] [--tracers 6] [--steps 6] [--rounds 4]
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
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--nx", type=int, default=680)
    ap.add_argument("--levels", type=int, default=80)
    ap.add_argument("--tracers", type=int, default=6)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--halo-width", type=int, default=4)
    ap.add_argument("--local-order", default="kd", choices=["curve", "hilbert", "kd", "global"])
    a = ap.parse_args()
    K, NT = a.levels, a.tracers
    oa.device_init(0)
    g = reorder_cells_morton(planar_hex(a.nx, a.nx, 30e3))
    gm = oa.GlobalMesh(g)
    cell_task, _ = oa.partition_cells(gm, a.parts, "graph")
    decomp = oa.Decomp(gm, a.parts, a.rank, a.halo_width, cell_task=cell_task, local_order=a.local_order)
    mesh = oa.HorzMesh(decomp, K)
    halo = oa.Halo(decomp)
    halo.set_transport(lambda *args: 0)          # a wire that moves nothing
    cell_id, edge_id = decomp.get_array("CellID"), decomp.get_array("EdgeID")
    hg, ug, trg = synthetic_state(g, K, NT)

    def to_local(glob, ids, rows):
        out = np.zeros(glob.shape[:-2] + (rows, glob.shape[-1]))
        out[..., : rows - 1, :] = glob[..., ids[: rows - 1] - 1, :]
        return out
    h, u, tr = to_local(hg, cell_id, mesh.NCellsSize), to_local(ug, edge_id, mesh.NEdgesSize), to_local(trg, cell_id, mesh.NCellsSize)
    state = oa.OceanState(mesh, halo, K, 2)
    tracers = oa.Tracers(mesh, halo, K, NT, 2)
    aux = oa.AuxiliaryState(mesh, halo, K, NT)
    tend = oa.Tendencies(mesh, K, NT, oa.default_config())
    stream = oa.Stream()
    stepper = oa.TimeStepper("RungeKutta4", 600.0, tend, aux, mesh, halo, tracers)
    stepper.set_option("OverlapHaloExchange", True)
    configs = {"all": (1, 1, 1), "send_band+band_on_comm": (1, 1, 0), "send_band": (1, 0, 0), "full_band": (0, 0, 0)}
    res = {k: [] for k in configs}
    for rnd in range(a.rounds):
        for name, (sb, bc, sh) in configs.items():
            oa.set_option("SendBand", sb)
            oa.set_option("BandOnComm", bc)
            oa.set_option("ShrinkSweeps", sh)
            state.copy_to_device(h, u, 0)
            tracers.copy_to_device(tr, 0)
            stepper.do_step(state, stream=stream)
            oa.device_synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                stepper.do_step(state, stream=stream)
            oa.device_synchronize()
            res[name].append(round(1e3 * (time.perf_counter() - t0) / a.steps, 4))
    oa.set_option("SendBand", 1)
    oa.set_option("BandOnComm", 1)
    oa.set_option("ShrinkSweeps", 1)
    base = min(res["full_band"])
    out = {"probe": "send_band", "parts": a.parts, "rank": a.rank, "cells_global": int(g["nCells"]), "levels": K, "tracers": NT,
           "halo_width": a.halo_width,
           "NCellsOwned": int(mesh.NCellsOwned), "NCellsAll": int(mesh.NCellsAll),
           "NBandCells": mesh.get_int("NBandCells"), "NBandSendCells": mesh.get_int("NBandSendCells"),
           "NInteriorCells": mesh.get_int("NInteriorCells"), "NIrregularEdges": mesh.get_int("NIrregularEdges"),
           "NIrregularOwned": mesh.get_int("NIrregularOwned"),
           "rk4_ms": res, "rk4_ms_min": {k: min(v) for k, v in res.items()},
           "gain_vs_full_band": {k: round(1.0 - min(v) / base, 4) for k, v in res.items()},
           "note": "one process = one rank's kernel sequence; the wire moves nothing, so the state is not meaningful"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
