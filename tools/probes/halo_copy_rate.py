"""Pack + unpack rate of the halo job-table kernels on one rank of an N-way decomposition, alone on the GPU (a wire that
moves nothing): exchange of the tracer array [NT][NCellsSize][K].

   python tools/probes/halo_copy_rate.py [--parts 8] [--nx 680] [--levels 80] [--tracers 6]
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
from omega_amd.meshgen import planar_hex, reorder_cells_morton  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--nx", type=int, default=680)
    ap.add_argument("--levels", type=int, default=80)
    ap.add_argument("--tracers", type=int, default=6)
    a = ap.parse_args()
    K, NT = a.levels, a.tracers
    oa.device_init(0)
    g = reorder_cells_morton(planar_hex(a.nx, a.nx, 30e3))
    gm = oa.GlobalMesh(g)
    cell_task, _ = oa.partition_cells(gm, a.parts, "graph")
    decomp = oa.Decomp(gm, a.parts, 0, 4, cell_task=cell_task, local_order="curve")
    mesh = oa.HorzMesh(decomp, K)
    halo = oa.Halo(decomp)
    halo.set_transport(lambda *args: 0)
    kp = oa.level_pitch(K)
    buf = oa.DeviceBuffer(np.zeros((NT, mesh.NCellsSize, kp)))
    stream = oa.Stream()
    n_halo = mesh.NCellsAll - mesh.NCellsOwned
    for _ in range(3):
        halo.exchange(buf.ptr, NT, mesh.NCellsSize, K, 0, stream=stream, row_pitch=kp)
    oa.device_synchronize()
    reps = 50
    ev0, ev1 = oa.Event(), oa.Event()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(reps):
        halo.exchange(buf.ptr, NT, mesh.NCellsSize, K, 0, stream=stream, row_pitch=kp)
    ev1.record(stream)
    oa.device_synchronize()
    wall = (time.perf_counter() - t0) / reps
    dev = ev0.elapsed_ms(ev1) / reps
    recv_bytes = n_halo * NT * K * 8
    send_rows = halo.recv_rows(NT, 0, 0) if hasattr(halo, "recv_rows") else None
    print(json.dumps({"probe": "halo_copy_rate", "parts": a.parts, "halo_cells": int(n_halo), "recv_MB": round(recv_bytes / 1e6, 2),
                      "exchange_device_us": round(1e3 * dev, 1), "exchange_wall_us": round(1e6 * wall, 1),
                      "note": "pack kernel + (no wire) + unpack kernel per exchange; each moves about recv_MB in and out"}))


if __name__ == "__main__":
    main()
