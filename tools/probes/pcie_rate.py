"""The PCIe-inclusive rate of one RHS evaluation, for the record: the boundary hands over DEVICE
arrays -- as the reference's Kokkos views are -- so nothing crosses PCIe inside a step; this probe times what it would
cost if the state were uploaded before and the tendencies downloaded after every evaluation (pageable numpy arrays
through OceanState::copyToDevice / copyToHost, the only host <-> device path the boundary has).

   python tools/probes/pcie_rate.py [--nx 680] [--levels 80] [--tracers 6]
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
    ap.add_argument("--nx", type=int, default=680)
    ap.add_argument("--levels", type=int, default=80)
    ap.add_argument("--tracers", type=int, default=6)
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
    stream = oa.Stream()
    up, rhs, down = [], [], []
    for _ in range(4):
        oa.device_synchronize()
        t0 = time.perf_counter()
        state.copy_to_device(h, u, 0)
        tracers.copy_to_device(tr, 0)
        oa.device_synchronize()
        t1 = time.perf_counter()
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
        oa.device_synchronize()
        t2 = time.perf_counter()
        out = [tend.get(0), tend.get(1), tend.get(2)]
        oa.device_synchronize()
        t3 = time.perf_counter()
        up.append(t1 - t0), rhs.append(t2 - t1), down.append(t3 - t2)
    nbytes_up = h.nbytes + u.nbytes + tr.nbytes
    nbytes_down = sum(x.nbytes for x in out)
    cl = g["nCells"] * K
    rec = {"probe": "pcie_rate", "cells": int(g["nCells"]), "levels": K, "tracers": NT,
           "upload_GB": round(nbytes_up / 1e9, 3), "upload_s": round(min(up), 4), "upload_GBps": round(nbytes_up / min(up) / 1e9, 2),
           "rhs_ms": round(1e3 * min(rhs), 3),
           "download_GB": round(nbytes_down / 1e9, 3), "download_s": round(min(down), 4),
           "download_GBps": round(nbytes_down / min(down) / 1e9, 2),
           "resident_cell_level_updates_per_s": round(cl / min(rhs), 1),
           "pcie_inclusive_cell_level_updates_per_s": round(cl / (min(up) + min(rhs) + min(down)), 1),
           "note": "pageable host arrays, synchronous copies, no overlap; the product path keeps every array on the device"}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
