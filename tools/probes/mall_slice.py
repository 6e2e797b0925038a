"""What would it be worth if the fused RHS's intermediates were read back from the 256 MiB memory-side cache?

The three dependency levels exchange KE, Div, the vertex pair, the PV sums, Del2Div / Del2RelVort and Del2Tracers through
HBM.  A cache-blocked walk (block of cells x one 16-level chunk: level 1 -> 2 -> 3 before the next block) would find
them in the Infinity Cache.  This probe measures the upper bound of that with the PRODUCTION kernels and no new
machinery: option ProbeSlice cuts every launch into (blocks x level chunks) launches of the same kernels
(KernelCommon.h: SliceWindow) and issues them

   order 1  block by block, L1 -> L2 -> L3 per block      (intermediates resident when read)
   order 2  the same launches level by level               (every intermediate evicted before it is read)

The blocks ignore the dependencies across their rims (timings only).  Same launches, same sizes, same ramps and tails:
order 2 - order 1 = what the residency buys; order 1 against the plain 3-launch evaluation = what the cutting costs.

   python tools/probes/mall_slice.py [--nx 680 --ny 680] [--blocks 8,16,32,64] [--steps 10]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import omega_amd as oa  # noqa: E402
from omega_amd.meshgen import planar_hex, synthetic_state_rows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=680)
    ap.add_argument("--ny", type=int, default=680)
    ap.add_argument("--levels", type=int, default=80)
    ap.add_argument("--tracers", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--blocks", default="8,16,32,64")
    a = ap.parse_args()
    K, NT = a.levels, a.tracers
    oa.device_init(0)
    g = planar_hex(a.nx, a.ny, 30e3)
    decomp = oa.Decomp(oa.GlobalMesh(g), 1, 0, 3, local_order="kd")
    mesh = oa.HorzMesh(decomp, K)
    cells0 = decomp.get_array("CellID")[: mesh.NCellsAll] - 1
    edges0 = decomp.get_array("EdgeID")[: mesh.NEdgesAll] - 1
    hh, uu, tt = synthetic_state_rows(g, K, NT, cells0, edges0)
    h, u = np.zeros((mesh.NCellsSize, K)), np.zeros((mesh.NEdgesSize, K))
    tr = np.zeros((NT, mesh.NCellsSize, K))
    h[: mesh.NCellsAll], u[: mesh.NEdgesAll], tr[:, : mesh.NCellsAll] = hh, uu, tt
    state, tracers = oa.OceanState(mesh, None, K, 2), oa.Tracers(mesh, None, K, NT, 2)
    aux, tend = oa.AuxiliaryState(mesh, None, K, NT), oa.Tendencies(mesh, K, NT, oa.default_config())
    state.copy_to_device(h, u, 0)
    tracers.copy_to_device(tr, 0)
    stream = oa.Stream()

    def timed(reps=3):
        best = []
        for _ in range(reps):
            for _ in range(2):
                tend.compute_all_tendencies(state, aux, tracers, stream=stream)
            oa.device_synchronize()
            e0, e1 = oa.Event(), oa.Event()
            e0.record(stream)
            for _ in range(a.steps):
                tend.compute_all_tendencies(state, aux, tracers, stream=stream)
            e1.record(stream)
            oa.device_synchronize()
            best.append(e0.elapsed_ms(e1) / a.steps)
        return round(min(best), 4), [round(x, 4) for x in best]

    # settle
    for _ in range(20):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    oa.device_synchronize()
    out = {"probe": "mall_slice", "cells": int(g["nCells"]), "levels": K, "tracers": NT, "steps": a.steps,
           "plain_three_launches_ms": timed(), "blocks": {}}
    ref = tend.get(0)[: mesh.NCellsOwned].copy()
    cell_bytes = 653 * 16     # measured HBM bytes per cell-level (profiles/r04p_qu30_pmc.json) x one level chunk
    for nb in [int(x) for x in a.blocks.split(",")]:
        rec = {"cells_per_block": int(g["nCells"] // nb), "block_chunk_traffic_MB": round(cell_bytes * g["nCells"] / nb / 1e6, 1),
               "launches_per_evaluation": 3 * 5 * nb}
        for order, name in ((1, "block_by_block_ms"), (2, "level_by_level_ms")):
            oa.set_option("ProbeBlocks", nb)
            oa.set_option("ProbeSlice", order)
            rec[name] = timed()
            oa.set_option("ProbeSlice", 0)
        # (one block: no rims -- the cut launches must then give the plain evaluation's bits)
        if nb == 1:
            rec["bits_equal_plain"] = bool(np.array_equal(tend.get(0)[: mesh.NCellsOwned], ref))
        rec["residency_gain_pct"] = round(100 * (1 - rec["block_by_block_ms"][0] / rec["level_by_level_ms"][0]), 2)
        rec["vs_plain_pct"] = round(100 * (rec["block_by_block_ms"][0] / out["plain_three_launches_ms"][0] - 1), 2)
        out["blocks"][str(nb)] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
