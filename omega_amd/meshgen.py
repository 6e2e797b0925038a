"""Synthetic MPAS-convention meshes (input synthesiser for tests and bench.py).

No MPAS mesh file exists offline, so every workload in this
repo runs on meshes generated here.  The arrays follow the MPAS mesh-file
conventions that Omega's ``Decomp``/``HorzMesh`` read
(reference: components/omega/src/base/Decomp.cpp:108-395 for the connectivity
names, components/omega/src/ocn/HorzMesh.cpp:424-523 for the geometry names),
except that indices are 0-based and a missing neighbour is ``-1`` (the file
convention is 1-based with 0 = missing; ``Decomp`` converts to 0-based and maps
missing -> the sentinel row, Decomp.cpp:553-574).

This module is *not* on the product path: the library takes mesh arrays through
the C ABI (include/omega_amd.h) exactly as a NetCDF reader would supply them.
"""
from __future__ import annotations

import numpy as np

I4 = np.int32


def _trisk_edges_on_edge(mesh: dict) -> None:
    """edgesOnEdge / nEdgesOnEdge / weightsOnEdge by the TRiSK construction
    (Thuburn et al. 2009; Ringler et al. 2010 eq. 33), generic for any Voronoi
    mesh given edgesOnCell (CCW), kiteAreasOnVertex, areaCell, dvEdge, dcEdge.

    For edge e and each of its two cells i (in cellsOnEdge order) walk the other
    edges e' of i counter-clockwise starting after e, accumulating
    R = sum(kite(i, v) / areaCell(i)) over the vertices passed, and set
    w(e,e') = s_e(i) * (1/2 - R) * n_{e'}(i) * dv(e') / dc(e)
    where n_{e'}(i) = +1 if the normal of e' points out of cell i and
    s_e(i) = +1 for i = cellsOnEdge(e,0), -1 otherwise ... sign fixed so that the
    reconstructed component is along t = k x n (checked in tests/test_meshgen.py).
    """
    nE = mesh["nEdges"]
    maxE = mesh["maxEdges"]
    eoc = mesh["edgesOnCell"]
    nEoc = mesh["nEdgesOnCell"]
    coe = mesh["cellsOnEdge"]
    voe = mesh["verticesOnEdge"]
    cov = mesh["cellsOnVertex"]
    kite = mesh["kiteAreasOnVertex"]
    areaC = mesh["areaCell"]
    dv = mesh["dvEdge"]
    dc = mesh["dcEdge"]

    eoe = np.full((nE, 2 * maxE), -1, dtype=I4)
    woe = np.zeros((nE, 2 * maxE), dtype=np.float64)
    neoe = np.zeros(nE, dtype=I4)

    edges = np.arange(nE, dtype=np.int64)
    for side in range(2):
        cell = coe[:, side].astype(np.int64)
        valid = cell >= 0
        cell_s = np.where(valid, cell, 0)
        n = nEoc[cell_s].astype(np.int64)
        # position of e in edgesOnCell(cell)
        pos = np.zeros(nE, dtype=np.int64)
        found = np.zeros(nE, dtype=bool)
        for j in range(maxE):
            hit = (eoc[cell_s, j] == edges) & (j < n)
            pos = np.where(hit, j, pos)
            found |= hit
        assert np.all(found | ~valid), "edge not found on its own cell"
        R = np.zeros(nE, dtype=np.float64)
        prev = edges.copy()
        s_e = 1.0 if side == 0 else -1.0
        for k in range(1, maxE):
            act = valid & (k < n)
            cur = eoc[cell_s, (pos + k) % np.maximum(n, 1)].astype(np.int64)
            cur_s = np.where(act, cur, 0)
            prev_s = np.where(act, prev, 0)
            # shared vertex between prev and cur
            v = np.full(nE, -1, dtype=np.int64)
            for a in range(2):
                for b in range(2):
                    m = voe[prev_s, a] == voe[cur_s, b]
                    v = np.where(m & act, voe[prev_s, a], v)
            assert np.all((v >= 0) | ~act), "consecutive edges share no vertex"
            v_s = np.where(act, v, 0)
            # kite area of (vertex v, cell)
            ka = np.zeros(nE, dtype=np.float64)
            for j in range(cov.shape[1]):
                ka = np.where(cov[v_s, j] == cell_s, kite[v_s, j], ka)
            R = R + np.where(act, ka / areaC[cell_s], 0.0)
            n_out = np.where(coe[cur_s, 0] == cell_s, 1.0, -1.0)
            w = s_e * (0.5 - R) * n_out * dv[cur_s] / dc
            col = neoe.astype(np.int64)
            rows = np.nonzero(act)[0]
            eoe[rows, col[rows]] = cur_s[rows].astype(I4)
            woe[rows, col[rows]] = w[rows]
            neoe = neoe + act.astype(I4)
            prev = np.where(act, cur, prev)
    mesh["edgesOnEdge"] = eoe
    mesh["weightsOnEdge"] = woe
    mesh["nEdgesOnEdge"] = neoe


def planar_hex(nx: int, ny: int, dc: float = 1.0, *, f0: float = 1.0e-4,
               bottom_depth: float = 2.0) -> dict:
    """Doubly periodic planar mesh of regular hexagons, ``nx`` x ``ny`` cells.

    Layout follows the MPAS-Tools ``planar_hex``/``periodic_hex`` convention the
    reference's planar test mesh ("PlanarPeriodic48x48.nc",
    components/omega/doc/devGuide/QuickStart.md:154) was made with: row-major
    cells, odd rows shifted by dc/2, each cell owning its W, SW, SE edges
    (3c, 3c+1, 3c+2; angleEdge 0, pi/3, 2pi/3) and its lower-left and bottom
    vertices (2c, 2c+1).  Lx = nx*dc, Ly = ny*dc*sqrt(3)/2; ``ny`` must be even.
    """
    if ny % 2:
        raise ValueError("ny must be even for y-periodicity")
    nC, nE, nV = nx * ny, 3 * nx * ny, 2 * nx * ny
    maxE = 6
    col, row = np.meshgrid(np.arange(nx), np.arange(ny))
    col = col.ravel()
    row = row.ravel()
    c = np.arange(nC, dtype=np.int64)

    def cid(r, q):
        return (np.mod(r, ny) * nx + np.mod(q, nx)).astype(np.int64)

    odd = (row % 2) == 1
    W = cid(row, col - 1)
    E = cid(row, col + 1)
    SW = np.where(odd, cid(row - 1, col), cid(row - 1, col - 1))
    SE = np.where(odd, cid(row - 1, col + 1), cid(row - 1, col))
    NW = np.where(odd, cid(row + 1, col), cid(row + 1, col - 1))
    NE = np.where(odd, cid(row + 1, col + 1), cid(row + 1, col))

    m = {"nCells": nC, "nEdges": nE, "nVertices": nV, "maxEdges": maxE,
         "vertexDegree": 3, "on_a_sphere": False,
         "x_period": nx * dc, "y_period": ny * dc * np.sqrt(3.0) / 2.0, "dc": dc}
    m["nEdgesOnCell"] = np.full(nC, 6, dtype=I4)
    m["cellsOnCell"] = np.stack([W, SW, SE, E, NE, NW], axis=1).astype(I4)
    m["edgesOnCell"] = np.stack(
        [3 * c, 3 * c + 1, 3 * c + 2, 3 * E, 3 * NE + 1, 3 * NW + 2], axis=1).astype(I4)
    m["verticesOnCell"] = np.stack(
        [2 * c, 2 * c + 1, 2 * E, 2 * NE + 1, 2 * NE, 2 * NW + 1], axis=1).astype(I4)

    coe = np.empty((nE, 2), dtype=I4)
    coe[0::3, 0], coe[0::3, 1] = W, c
    coe[1::3, 0], coe[1::3, 1] = SW, c
    coe[2::3, 0], coe[2::3, 1] = SE, c
    m["cellsOnEdge"] = coe
    voe = np.empty((nE, 2), dtype=I4)     # v0 -> v1 along t = k x n
    voe[0::3, 0], voe[0::3, 1] = 2 * c, 2 * NW + 1
    voe[1::3, 0], voe[1::3, 1] = 2 * c + 1, 2 * c
    voe[2::3, 0], voe[2::3, 1] = 2 * E, 2 * c + 1
    m["verticesOnEdge"] = voe

    cov = np.empty((nV, 3), dtype=I4)
    cov[0::2] = np.stack([c, W, SW], axis=1)
    cov[1::2] = np.stack([c, SW, SE], axis=1)
    m["cellsOnVertex"] = cov
    eov = np.empty((nV, 3), dtype=I4)
    eov[0::2] = np.stack([3 * c, 3 * W + 2, 3 * c + 1], axis=1)
    eov[1::2] = np.stack([3 * c + 1, 3 * SE, 3 * c + 2], axis=1)
    m["edgesOnVertex"] = eov

    # coordinates (periodic_hex convention, 1-based row/col in the original)
    xC = np.where(odd, dc * (col + 1.0), dc * (col + 1.0) - 0.5 * dc)
    yC = dc * (row + 1.0) * np.sqrt(3.0) / 2.0
    m["xCell"], m["yCell"], m["zCell"] = xC, yC, np.zeros(nC)
    xE = np.empty(nE)
    yE = np.empty(nE)
    xE[0::3], yE[0::3] = xC - 0.5 * dc, yC
    xE[1::3], yE[1::3] = xC - 0.5 * dc * np.cos(np.pi / 3), yC - 0.5 * dc * np.sin(np.pi / 3)
    xE[2::3], yE[2::3] = xC + 0.5 * dc * np.cos(np.pi / 3), yC - 0.5 * dc * np.sin(np.pi / 3)
    m["xEdge"], m["yEdge"], m["zEdge"] = xE, yE, np.zeros(nE)
    xV = np.empty(nV)
    yV = np.empty(nV)
    xV[0::2], yV[0::2] = xC - 0.5 * dc, yC - dc * np.sqrt(3.0) / 6.0
    xV[1::2], yV[1::2] = xC, yC - dc * np.sqrt(3.0) / 3.0
    m["xVertex"], m["yVertex"], m["zVertex"] = xV, yV, np.zeros(nV)
    for el, n in (("Cell", nC), ("Edge", nE), ("Vertex", nV)):
        m["lon" + el] = np.zeros(n)
        m["lat" + el] = np.zeros(n)
    ang = np.empty(nE)
    ang[0::3], ang[1::3], ang[2::3] = 0.0, np.pi / 3.0, 2.0 * np.pi / 3.0
    m["angleEdge"] = ang

    m["areaCell"] = np.full(nC, dc * dc * np.sqrt(3.0) / 2.0)
    m["areaTriangle"] = np.full(nV, dc * dc * np.sqrt(3.0) / 4.0)
    m["kiteAreasOnVertex"] = np.full((nV, 3), dc * dc * np.sqrt(3.0) / 12.0)
    m["dcEdge"] = np.full(nE, dc)
    m["dvEdge"] = np.full(nE, dc * np.sqrt(3.0) / 3.0)
    m["fCell"] = np.full(nC, f0)
    m["fEdge"] = np.full(nE, f0)
    m["fVertex"] = np.full(nV, f0)
    m["bottomDepth"] = np.full(nC, bottom_depth)
    _trisk_edges_on_edge(m)
    return m


def reorder_cells_blocked(mesh: dict, block: int = 16) -> dict:
    """Renumber a planar_hex mesh so cells are stored block by block
    (``block`` x ``block`` cells), with edges and vertices following their
    owning cell (3c.., 2c..).  Real MPAS meshes are likewise sorted for
    locality (MPAS-Tools ``sort_mesh``); physics is numbering independent.
    """
    nC = mesh["nCells"]
    nx = int(round(mesh["x_period"] / mesh["dc"]))
    ny = nC // nx
    col, row = np.meshgrid(np.arange(nx), np.arange(ny))
    col, row = col.ravel(), row.ravel()
    nbx = (nx + block - 1) // block
    key = ((row // block) * nbx + (col // block)).astype(np.int64) * (block * block) \
        + (row % block) * block + (col % block)
    old_of_new = np.argsort(key, kind="stable")
    return permute_mesh(mesh, old_of_new,
                        (3 * old_of_new[:, None] + np.arange(3)).ravel(),
                        (2 * old_of_new[:, None] + np.arange(2)).ravel())


def permute_mesh(mesh: dict, cell_old_of_new, edge_old_of_new, vertex_old_of_new) -> dict:
    """Apply element permutations (new index i holds old element old_of_new[i])."""
    out = dict(mesh)
    perms = {"Cell": np.asarray(cell_old_of_new), "Edge": np.asarray(edge_old_of_new),
             "Vertex": np.asarray(vertex_old_of_new)}
    inv = {}
    for k, p in perms.items():
        q = np.empty(len(p) + 1, dtype=np.int64)
        q[p] = np.arange(len(p))
        q[-1] = -1                       # missing stays missing
        inv[k] = q
    owner = {"nEdgesOnCell": "Cell", "cellsOnCell": "Cell", "edgesOnCell": "Cell",
             "verticesOnCell": "Cell", "areaCell": "Cell", "fCell": "Cell",
             "bottomDepth": "Cell", "cellsOnEdge": "Edge", "verticesOnEdge": "Edge",
             "edgesOnEdge": "Edge", "weightsOnEdge": "Edge", "nEdgesOnEdge": "Edge",
             "angleEdge": "Edge", "dcEdge": "Edge", "dvEdge": "Edge", "fEdge": "Edge",
             "cellsOnVertex": "Vertex", "edgesOnVertex": "Vertex",
             "kiteAreasOnVertex": "Vertex", "areaTriangle": "Vertex", "fVertex": "Vertex"}
    for el in ("Cell", "Edge", "Vertex"):
        for pre in ("x", "y", "z", "lon", "lat"):
            owner[pre + el] = el
    target = {"cellsOnCell": "Cell", "edgesOnCell": "Edge", "verticesOnCell": "Vertex",
              "cellsOnEdge": "Cell", "verticesOnEdge": "Vertex", "edgesOnEdge": "Edge",
              "cellsOnVertex": "Cell", "edgesOnVertex": "Edge"}
    for name, own in owner.items():
        if name not in mesh:
            continue
        a = mesh[name][perms[own]]
        if name in target:
            a = inv[target[name]][a.astype(np.int64)].astype(I4)
        out[name] = np.ascontiguousarray(a)
    return out


def synthetic_state(mesh: dict, nvertlayers: int, ntracers: int, seed: int = 20251003):
    """Synthetic prognostic state on the GLOBAL mesh: smooth fields
    plus seeded noise, h strictly > 1.  Returns h [nCells,K], u [nEdges,K], tr [NT,nCells,K]."""
    rng = np.random.default_rng(seed)
    K = nvertlayers
    nC, nE = mesh["nCells"], mesh["nEdges"]
    kfac = 1.0 + 0.05 * np.arange(K)[None, :] / max(K, 1)
    if mesh.get("on_a_sphere", False):
        lonC, latC, lonE, latE = mesh["lonCell"], mesh["latCell"], mesh["lonEdge"], mesh["latEdge"]
        sC = np.cos(lonC) * np.cos(latC) ** 4
        ux = -np.sin(lonE) ** 2 * np.cos(latE) ** 3
        uy = -4 * np.sin(lonE) * np.cos(lonE) * np.cos(latE) ** 3 * np.sin(latE)
    else:
        ax, ay = 2 * np.pi / mesh["x_period"], 2 * np.pi / mesh["y_period"]
        sC = np.cos(ax * mesh["xCell"]) * np.cos(ay * mesh["yCell"])
        ux = np.sin(ax * mesh["xEdge"]) * np.cos(ay * mesh["yEdge"])
        uy = np.cos(ax * mesh["xEdge"]) * np.sin(ay * mesh["yEdge"])
    # noise is drawn in 8 fixed row-chunks, each from its own child stream of the seed, so the
    # result does not depend on how many threads fill it
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=8)
    NCH = 8

    def fill(out, col_base, lev, amp, offset, stream_id):
        """out[i,k] = offset + col_base[i]*lev[k] + amp*U(-1,1)"""
        n = out.shape[0]
        seeds = np.random.SeedSequence([seed, stream_id]).spawn(NCH)
        bounds = np.linspace(0, n, NCH + 1).astype(np.int64)

        def work(c):
            lo, hi = bounds[c], bounds[c + 1]
            blk = out[lo:hi]
            np.random.default_rng(seeds[c]).random(out=blk)
            blk *= 2.0 * amp
            blk += offset - amp
            blk += col_base[lo:hi, None] * lev
        list(pool.map(work, range(NCH)))

    lev = kfac[0]
    h = np.empty((nC, K))
    fill(h, 0.5 * sC, lev, 0.1, 2.0, 0)
    un = np.cos(mesh["angleEdge"]) * ux + np.sin(mesh["angleEdge"]) * uy
    u = np.empty((nE, K))
    fill(u, un, lev, 0.01, 0.0, 1)
    tr = np.empty((max(ntracers, 1), nC, K))
    for l in range(max(ntracers, 1)):
        fill(tr[l], -sC, lev, 0.05, 2.0 + 0.1 * l, 2 + l)
    pool.shutdown()
    return h, u, tr
