"""Shared test fixture: a mesh + synthetic state on the product path and on the oracle."""
import numpy as np

import omega_amd as oa
from omega_amd.meshgen import synthetic_state
from oracle import oracle as O


def to_local(glob, ids_1based, rows_size):
    """Scatter a global [..., nGlobal, K] array into local order with a zero sentinel row."""
    n = rows_size - 1
    out = np.zeros(glob.shape[:-2] + (rows_size, glob.shape[-1]))
    out[..., :n, :] = glob[..., ids_1based[:n] - 1, :]
    return out


class Problem:
    """One rank's view of a global mesh: product objects (if a GPU is present) + oracle."""

    def __init__(self, g, K, NT, nparts=1, rank=0, device=True, config=None, seed=20251003, halo_width=3,
                 oracle=True, local_order="global", partition="rcb"):
        self.g, self.K, self.NT = g, K, NT
        self.gm = oa.GlobalMesh(g)
        cell_task = oa.partition_cells(self.gm, nparts, "graph")[0] if (partition == "graph" and nparts > 1) else None
        self.decomp = oa.Decomp(self.gm, nparts, rank, halo_width, cell_task=cell_task, local_order=local_order)
        self.mesh = oa.HorzMesh(self.decomp, K, host_only=not device)
        self.cell_id = self.decomp.get_array("CellID")
        self.edge_id = self.decomp.get_array("EdgeID")
        self.vertex_id = self.decomp.get_array("VertexID")
        hg, ug, trg = synthetic_state(g, K, NT, seed)
        self.h = to_local(hg, self.cell_id, self.mesh.NCellsSize)
        self.u = to_local(ug, self.edge_id, self.mesh.NEdgesSize)
        self.tr = to_local(trg, self.cell_id, self.mesh.NCellsSize)
        self.config_over = dict(config or {})
        # oracle on the product's own local mesh arrays (skipped for the full-size property tests)
        if oracle:
            self.omesh = O.Mesh(self.mesh.local_arrays(), K)
            self.oracle = O.Oracle(self.omesh, NT, O.default_config(**self.config_over))
        if device:
            cfg = oa.default_config(**self.config_over)
            self.halo = oa.Halo(self.decomp) if nparts > 1 else None
            self.state = oa.OceanState(self.mesh, self.halo, K, 2)
            self.tracers = oa.Tracers(self.mesh, self.halo, K, NT, 2)
            self.aux = oa.AuxiliaryState(self.mesh, self.halo, K, NT)
            self.aux.set_options(cfg.FluxThicknessUpwind, cfg.FluxTracerUpwind, cfg.WindInterpIsotropic)
            self.tend = oa.Tendencies(self.mesh, K, NT, cfg)
            self.state.copy_to_device(self.h, self.u, 0)
            if NT > 0:
                self.tracers.copy_to_device(self.tr, 0)


def max_rel_diff(a, b, scale=None):
    """max |a-b| / max(|a|,|b|,scale); scale defaults to 1e-300 (pure relative)."""
    d = np.abs(a - b)
    den = np.maximum(np.maximum(np.abs(a), np.abs(b)), 1e-300 if scale is None else scale)
    with np.errstate(invalid="ignore"):
        r = d / den
    r[~np.isfinite(r)] = 0.0 if np.array_equal(np.isnan(a), np.isnan(b)) else np.inf
    return float(r.max()) if r.size else 0.0


def poison_tendencies(P):
    """Every value of the three tendency arrays NaN (test/ocn/TendenciesTest.cpp:159-163 does so before its evaluation) --
    the whole device arrays, row padding included; only the zero sentinel row, which no kernel writes, stays zero."""
    pitch = oa.level_pitch(P.K)
    for which, rows, planes in ((0, P.mesh.NCellsSize, 1), (1, P.mesh.NEdgesSize, 1), (2, P.mesh.NCellsSize, max(P.NT, 1))):
        ptr, _ = P.tend.device_ptr(which)
        poison = np.full((planes, rows, pitch), np.nan)
        poison[:, -1, :] = 0.0
        oa.copy_to_device(ptr, poison)

