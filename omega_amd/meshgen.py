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


def reorder_cells_morton(mesh: dict, hilbert: bool = False) -> dict:
    """Renumber a planar_hex mesh along the Z-order (Morton) curve of (col, row) -- any aligned run of
    2^k consecutive cells is a compact block, at every scale -- or along the Hilbert curve."""
    nC = mesh["nCells"]
    nx = int(round(mesh["x_period"] / mesh["dc"]))
    ny = nC // nx
    col, row = np.meshgrid(np.arange(nx), np.arange(ny))
    col, row = col.ravel().astype(np.int64), row.ravel().astype(np.int64)
    key = np.zeros(nC, dtype=np.int64)
    if hilbert:
        x, y = col.copy(), row.copy()
        n = 1 << 11
        s = n >> 1
        while s > 0:
            rx = ((x & s) > 0).astype(np.int64)
            ry = ((y & s) > 0).astype(np.int64)
            key += s * s * ((3 * rx) ^ ry)
            flip = (ry == 0) & (rx == 1)
            x = np.where(flip, s - 1 - (x & (s - 1)), x & (s - 1))
            y = np.where(flip, s - 1 - (y & (s - 1)), y & (s - 1))
            swap = ry == 0
            x, y = np.where(swap, y, x), np.where(swap, x, y)
            s >>= 1
    for b in range(16 if not hilbert else 0):
        key |= ((col >> b) & 1) << (2 * b)
        key |= ((row >> b) & 1) << (2 * b + 1)
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


def _hash_uniform(seed: int, stream: int, rows: np.ndarray, k: int) -> np.ndarray:
    """U[0, 1) for every (global row id, level): a counter-based generator (splitmix64 of seed, stream, row * k + level),
    so that a value depends on WHICH element it belongs to and on nothing else -- not on the partition, not on how many
    rows the caller asks for."""
    M = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(stream) * np.uint64(0x9E3779B97F4A7C15)) & M
        x = rows.astype(np.uint64)[:, None] * np.uint64(k) + np.arange(k, dtype=np.uint64)[None, :]
        z = (x + key) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def synthetic_state_rows(mesh: dict, nvertlayers: int, ntracers: int, cell_rows, edge_rows, seed: int = 20251003,
                         tracers=None):
    """The synthetic state of synthetic_state() -- the same smooth fields, h strictly > 1 -- for the given GLOBAL
    (0-based) cell and edge ids only, with noise that is a function of (global id, level) alone: a rank of a partitioned
    run builds exactly its local rows, never a global [nCells, K] array, and an N = 1 and an N = 8 run hold the same
    value for the same element, bit for bit (the reference initialises the state per task too,
    components/omega/src/ocn/OceanState.cpp:65-117).  Returns h [len(cell_rows), K], u [len(edge_rows), K],
    tr [NT, len(cell_rows), K] (`tracers`: an iterable of tracer indices to build instead of all)."""
    K = nvertlayers
    c = np.asarray(cell_rows, dtype=np.int64)
    e = np.asarray(edge_rows, dtype=np.int64)
    lev = 1.0 + 0.05 * np.arange(K) / max(K, 1)
    if mesh.get("on_a_sphere", False):
        lonC, latC, lonE, latE = mesh["lonCell"][c], mesh["latCell"][c], mesh["lonEdge"][e], mesh["latEdge"][e]
        sC = np.cos(lonC) * np.cos(latC) ** 4
        ux = -np.sin(lonE) ** 2 * np.cos(latE) ** 3
        uy = -4 * np.sin(lonE) * np.cos(lonE) * np.cos(latE) ** 3 * np.sin(latE)
    else:
        ax, ay = 2 * np.pi / mesh["x_period"], 2 * np.pi / mesh["y_period"]
        sC = np.cos(ax * mesh["xCell"][c]) * np.cos(ay * mesh["yCell"][c])
        ux = np.sin(ax * mesh["xEdge"][e]) * np.cos(ay * mesh["yEdge"][e])
        uy = np.cos(ax * mesh["xEdge"][e]) * np.sin(ay * mesh["yEdge"][e])

    def fill(rows, col_base, amp, offset, stream):
        out = _hash_uniform(seed, stream, rows, K)
        out *= 2.0 * amp
        out += offset - amp
        out += col_base[:, None] * lev
        return out
    h = fill(c, 0.5 * sC, 0.1, 2.0, 0)
    un = np.cos(mesh["angleEdge"][e]) * ux + np.sin(mesh["angleEdge"][e]) * uy
    u = fill(e, un, 0.01, 0.0, 1)
    which = list(range(max(ntracers, 1))) if tracers is None else list(tracers)
    tr = np.empty((len(which), len(c), K))
    for n, l in enumerate(which):
        tr[n] = fill(c, -sC, 0.05, 2.0 + 0.1 * l, 2 + l)
    return h, u, tr


# ---------------------------------------------------------------------------------------
# Spherical Voronoi mesh (quasi-uniform, pentagons / hexagons / heptagons)
# ---------------------------------------------------------------------------------------
def _sph_tri_area(a, b, c):
    """Area of the unit-sphere triangles (a,b,c) [...,3] (Van Oosterom & Strackee)."""
    num = np.abs(np.einsum("...i,...i->...", a, np.cross(b, c)))
    den = 1.0 + np.einsum("...i,...i->...", a, b) + np.einsum("...i,...i->...", b, c) \
        + np.einsum("...i,...i->...", c, a)
    return 2.0 * np.arctan2(num, den)


def _arc(a, b):
    return np.arctan2(np.linalg.norm(np.cross(a, b), axis=-1), np.einsum("...i,...i->...", a, b))


def icosahedral_points(level: int) -> np.ndarray:
    """Unit vectors of the icosahedral geodesic grid with 10*4**level + 2 points."""
    g = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array([[-1, g, 0], [1, g, 0], [-1, -g, 0], [1, -g, 0], [0, -1, g], [0, 1, g], [0, -1, -g], [0, 1, -g],
                  [g, 0, -1], [g, 0, 1], [-g, 0, -1], [-g, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1)[:, None]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    n = 2 ** level
    a, b = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    keep = (a + b) <= n
    a, b = a[keep].astype(np.float64) / n, b[keep].astype(np.float64) / n
    pts = []
    for (i, j, k) in f:
        q = (1 - a - b)[:, None] * v[i] + a[:, None] * v[j] + b[:, None] * v[k]
        pts.append(q / np.linalg.norm(q, axis=1)[:, None])
    pts = np.concatenate(pts)
    _, idx = np.unique(np.round(pts, 9), axis=0, return_index=True)
    return pts[np.sort(idx)]


def spherical_voronoi(n_cells: int = 0, *, points=None, radius: float = 6371220.0, lloyd: int = 6,
                      bottom_depth: float = 2.0, omega: float = 7.292e-5, sort: bool = True) -> dict:
    """Quasi-uniform Voronoi mesh of the sphere with ``n_cells`` cells in the MPAS conventions of
    ``planar_hex`` (stand-in for the QU240 / EC30to60 meshes, which are downloads).  Generators:
    Fibonacci lattice relaxed by ``lloyd`` Lloyd iterations (scipy SphericalVoronoi); the result has
    mostly hexagons plus pentagons and heptagons, maxEdges 7 or 8.  Conventions: edgesOnCell CCW seen
    from outside, verticesOnCell[j] between edges j and j+1, normal of an edge from cellsOnEdge 0 to 1,
    verticesOnEdge 0 -> 1 along k x n, edgesOnVertex[k] between cellsOnVertex[k] and [k+1] (CCW).
    With ``sort`` the cells are numbered along a Morton curve in (lon, sin lat) for locality."""
    from scipy.spatial import SphericalVoronoi
    if points is not None:          # e.g. icosahedral_points(level): 12 pentagons, hexagons otherwise
        pts = np.array(points, dtype=np.float64)
        n_cells = len(pts)
    else:
        i = np.arange(n_cells) + 0.5
        z = 1.0 - 2.0 * i / n_cells
        phi = i * np.pi * (3.0 - np.sqrt(5.0))
        r = np.sqrt(1.0 - z * z)
        pts = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)

    # Everything below is vectorised over padded [nCells, maxEdges] region tables; the arithmetic per element -- which
    # operands, in which order -- is that of the loop-per-cell generator of rounds 1-4 (kept as tests/meshgen_loops.py),
    # so the arrays are the same bit for bit (tests/test_meshgen_rows.py) at a tenth of the time.
    def regions_ccw(sv):
        """(reg [nC, maxE] vertex ids CCW seen from outside, -1 padded; n [nC])"""
        sv.sort_vertices_of_regions()
        n = np.fromiter(map(len, sv.regions), dtype=np.int64, count=len(sv.regions))
        flat = np.fromiter(chain.from_iterable(sv.regions), dtype=np.int64, count=int(n.sum()))
        width = int(n.max())
        j = np.arange(width)[None, :]
        valid = j < n[:, None]
        reg = np.full((len(n), width), -1, dtype=np.int64)
        reg[valid] = flat
        p = sv.points
        v0, v1 = sv.vertices[reg[:, 0]], sv.vertices[reg[:, 1]]
        flip = np.einsum("ij,ij->i", np.cross(v0 - p, v1 - p), p) < 0
        src = np.where(flip[:, None] & valid, n[:, None] - 1 - j, j)
        reg = np.where(valid, np.take_along_axis(reg, np.where(valid, src, 0), axis=1), -1)
        return reg, n

    def fan(pc, xvert, reg, n):
        """per cell and slot j: the unit-sphere triangle (cell point, v_j, v_j+1) -> (v_j, v_j+1, area), masked slots 0"""
        j = np.arange(reg.shape[1])[None, :]
        valid = j < n[:, None]
        nxt = np.take_along_axis(reg, np.where(valid, (j + 1) % n[:, None], 0), axis=1)
        va, vb = xvert[np.where(valid, reg, 0)], xvert[np.where(valid, nxt, 0)]
        w = _sph_tri_area(np.broadcast_to(pc[:, None, :], va.shape).reshape(-1, 3), va.reshape(-1, 3),
                          vb.reshape(-1, 3)).reshape(valid.shape)
        return va, vb, np.where(valid, w, 0.0), valid

    from itertools import chain
    for _ in range(lloyd):
        sv = SphericalVoronoi(pts, 1.0)
        reg, n = regions_ccw(sv)
        va, vb, w, valid = fan(pts, sv.vertices, reg, n)
        term = np.where(valid[:, :, None], (pts[:, None, :] + va + vb) * w[:, :, None], 0.0)
        cen = term[:, 0, :].copy()
        for j in range(1, reg.shape[1]):      # sequential over the slots, as a sum over axis 0 of [n, 3] runs
            cen = np.where(valid[:, j, None], cen + term[:, j, :], cen)
        # (the norm of ONE vector is sqrt(x.dot(x)) -- a BLAS dot, whose rounding np.linalg.norm(cen, axis=1) does not share)
        pts = cen / np.sqrt(np.fromiter((x.dot(x) for x in cen), dtype=np.float64, count=len(cen)))[:, None]
    if sort:
        lon = np.arctan2(pts[:, 1], pts[:, 0]) + np.pi
        qa = np.minimum((lon / (2 * np.pi) * 1024).astype(np.int64), 1023)
        qb = np.minimum(((pts[:, 2] + 1) / 2 * 1024).astype(np.int64), 1023)
        key = np.zeros(n_cells, dtype=np.int64)
        for b in range(10):
            key |= ((qa >> b) & 1) << (2 * b)
            key |= ((qb >> b) & 1) << (2 * b + 1)
        pts = pts[np.argsort(key, kind="stable")]
    sv = SphericalVoronoi(pts, 1.0)
    reg, nreg = regions_ccw(sv)
    xv = sv.vertices
    nC, nV = n_cells, len(xv)
    maxE = reg.shape[1]
    nEoc = nreg.astype(I4)
    jj = np.arange(maxE)[None, :]
    valid = jj < nreg[:, None]
    nxt = np.take_along_axis(reg, np.where(valid, (jj + 1) % nreg[:, None], 0), axis=1)
    # edges in order of first encounter over (cell ascending, slot ascending); an edge = an unordered vertex pair
    cc = np.broadcast_to(np.arange(nC, dtype=np.int64)[:, None], reg.shape)[valid]
    fa, fb = reg[valid], nxt[valid]
    pair = np.minimum(fa, fb) * nV + np.maximum(fa, fb)
    _, first, inv = np.unique(pair, return_index=True, return_inverse=True)
    nE = len(first)
    rank = np.empty(nE, dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(nE)
    e_flat = rank[inv]
    assert nE == nC + nV - 2 and len(pair) == 2 * nE, "not a closed Voronoi tessellation"
    occ = np.argsort(e_flat, kind="stable")            # the two occurrences of every edge, the first one first
    assert np.array_equal(e_flat[occ[0::2]], np.arange(nE)) and np.array_equal(e_flat[occ[1::2]], np.arange(nE))
    coe = np.stack([cc[occ[0::2]], cc[occ[1::2]]], axis=1).astype(I4)
    voe = np.stack([fa[occ[0::2]], fb[occ[0::2]]], axis=1).astype(I4)   # CCW around cell 0 == along k x n
    eoc = np.full((nC, maxE), -1, dtype=I4)
    voc = np.full((nC, maxE), -1, dtype=I4)
    coc = np.full((nC, maxE), -1, dtype=I4)
    eoc[valid] = e_flat
    voc[valid] = fb                                     # vertex between edge j and edge j+1
    coc[valid] = np.where(coe[e_flat, 0] == cc, coe[e_flat, 1], coe[e_flat, 0])
    regs_valid, regs = valid, reg
    # vertices: the three edges / cells around, counter-clockwise
    xe = pts[coe[:, 0]] + pts[coe[:, 1]]
    xe /= np.linalg.norm(xe, axis=1)[:, None]
    ev = np.concatenate([voe[:, 0], voe[:, 1]]).astype(np.int64)
    ee = np.concatenate([np.arange(nE), np.arange(nE)])
    o = np.lexsort((ee, ev))                            # by vertex, then by edge id
    assert len(o) == 3 * nV and np.array_equal(ev[o].reshape(nV, 3)[:, 0], np.arange(nV)) \
        and (ev[o].reshape(nV, 3) == np.arange(nV)[:, None]).all(), "degenerate Voronoi vertex"
    es = ee[o].reshape(nV, 3)
    ref = xe[es[:, 0]] - xv
    ang = np.zeros((nV, 3))
    for k in (1, 2):
        d = xe[es[:, k]] - xv
        ang[:, k] = np.arctan2(np.einsum("ij,ij->i", np.cross(ref, d), xv), np.einsum("ij,ij->i", ref, d)) % (2 * np.pi)
    swap = ang[:, 2] < ang[:, 1]
    es = np.where(swap[:, None], es[:, [0, 2, 1]], es)
    eov = es.astype(I4)
    cov = np.empty((nV, 3), dtype=I4)
    for k in range(3):    # cell k lies between edge k-1 and edge k (CCW): the cell shared by both
        ca, cb = coe[es[:, (k + 2) % 3]], coe[es[:, k]]
        hit0 = (ca[:, 0] == cb[:, 0]) | (ca[:, 0] == cb[:, 1])
        hit1 = (ca[:, 1] == cb[:, 0]) | (ca[:, 1] == cb[:, 1])
        assert (hit0 ^ hit1).all()
        cov[:, k] = np.where(hit0, ca[:, 0], ca[:, 1])
    m = {"nCells": nC, "nEdges": nE, "nVertices": nV, "maxEdges": maxE, "vertexDegree": 3,
         "on_a_sphere": True, "sphere_radius": radius}
    m["nEdgesOnCell"], m["edgesOnCell"], m["verticesOnCell"], m["cellsOnCell"] = nEoc, eoc, voc, coc
    m["cellsOnEdge"], m["verticesOnEdge"], m["cellsOnVertex"], m["edgesOnVertex"] = coe, voe, cov, eov
    for el, x in (("Cell", pts), ("Edge", xe), ("Vertex", xv)):
        m["x" + el], m["y" + el], m["z" + el] = radius * x[:, 0], radius * x[:, 1], radius * x[:, 2]
        m["lon" + el] = np.mod(np.arctan2(x[:, 1], x[:, 0]), 2 * np.pi)
        m["lat" + el] = np.arcsin(np.clip(x[:, 2], -1, 1))
    R2 = radius * radius
    m["dcEdge"] = radius * _arc(pts[coe[:, 0]], pts[coe[:, 1]])
    m["dvEdge"] = radius * _arc(xv[voe[:, 0]], xv[voe[:, 1]])
    _, _, w, _ = fan(pts, xv, regs, nreg)
    area = w[:, 0].copy()
    for j in range(1, maxE):                  # a sum of < 8 numbers runs in order
        area = np.where(regs_valid[:, j], area + w[:, j], area)
    for c in np.flatnonzero(nreg >= 8):       # (numpy sums 8 and more in blocks: leave those cells to it)
        area[c] = w[c, : nreg[c]].sum()
    m["areaCell"] = R2 * area
    m["areaTriangle"] = R2 * _sph_tri_area(pts[cov[:, 0]], pts[cov[:, 1]], pts[cov[:, 2]])
    kite = np.empty((nV, 3))
    for k in range(3):
        ea, eb = eov[:, (k + 2) % 3], eov[:, k]
        pc = pts[cov[:, k]]
        kite[:, k] = _sph_tri_area(pc, xe[ea], xv) + _sph_tri_area(pc, xv, xe[eb])
    m["kiteAreasOnVertex"] = R2 * kite
    # angle of the edge normal (cell 0 -> cell 1) against local east
    nvec = pts[coe[:, 1]] - pts[coe[:, 0]]
    nvec -= np.einsum("ij,ij->i", nvec, xe)[:, None] * xe
    east = np.stack([-np.sin(m["lonEdge"]), np.cos(m["lonEdge"]), np.zeros(nE)], axis=1)
    north = np.cross(xe, east)
    m["angleEdge"] = np.arctan2(np.einsum("ij,ij->i", nvec, north), np.einsum("ij,ij->i", nvec, east))
    for el in ("Cell", "Edge", "Vertex"):
        m["f" + el] = 2.0 * omega * np.sin(m["lat" + el])
    m["bottomDepth"] = np.full(nC, bottom_depth)
    _trisk_edges_on_edge(m)
    return m


def permute_cell_slots(mesh: dict, fraction: float = 0.01, seed: int = 11) -> dict:
    """A mesh file whose per-cell lists are consistent but NOT in ring order for some cells: for a random `fraction` of
    the cells two (non-adjacent where possible) slots of edgesOnCell / cellsOnCell / verticesOnCell are swapped -- not a
    cyclic shift: consecutive slots then no longer share a vertex.  The reference never relies on that order
    (it only compacts the lists, components/omega/src/base/Decomp.cpp:2030-2064); the sums over a cell's edges run in
    the file's slot order, which is what makes the order part of the result's bits."""
    out = dict(mesh)
    rng = np.random.default_rng(seed)
    n = mesh["nEdgesOnCell"]
    cells = np.flatnonzero(rng.random(mesh["nCells"]) < fraction)
    eoc, coc, voc = (mesh[k].copy() for k in ("edgesOnCell", "cellsOnCell", "verticesOnCell"))
    for c in cells:
        a, b = 0, 2 if n[c] > 3 else 1          # slots 0 and 2: neither a shift nor a reflection of the ring
        for arr in (eoc, coc, voc):
            arr[c, a], arr[c, b] = arr[c, b], arr[c, a]
    out["edgesOnCell"], out["cellsOnCell"], out["verticesOnCell"] = eoc, coc, voc
    out["permutedCells"] = cells
    return out


def pad_max_edges(mesh: dict, max_edges: int) -> dict:
    """The same mesh stored with a larger ``maxEdges`` dimension (mesh files often carry maxEdges = 7
    or more than any cell uses); padding entries are -1 / 0 as in a file."""
    old = mesh["maxEdges"]
    assert max_edges >= old
    out = dict(mesh)
    out["maxEdges"] = max_edges
    for name in ("edgesOnCell", "verticesOnCell", "cellsOnCell"):
        a = np.full((mesh["nCells"], max_edges), -1, dtype=I4)
        a[:, :old] = mesh[name]
        out[name] = a
    e = np.full((mesh["nEdges"], 2 * max_edges), -1, dtype=I4)
    e[:, : 2 * old] = mesh["edgesOnEdge"]
    w = np.zeros((mesh["nEdges"], 2 * max_edges))
    w[:, : 2 * old] = mesh["weightsOnEdge"]
    out["edgesOnEdge"], out["weightsOnEdge"] = e, w
    return out


# ---------------------------------------------------------------------------------------
# Land boundaries: culled meshes (every real ocean mesh -- QU240, EC30to60, QU30, oRRS18to6 -- is one)
# ---------------------------------------------------------------------------------------
def cull(mesh: dict, keep, *, first_cell_valid: bool = True, compact_edges_on_edge: bool = False) -> dict:
    """Remove the cells with ``keep[c] == False`` and renumber, the way a culled MPAS ocean mesh looks
    (MPAS-Tools MpasCellCuller; what Omega then reads: components/omega/src/base/Decomp.cpp:553-574
    missing -> sentinel row, :2043-2064 per-cell edge compaction, :2187-2199 zero EdgesOnEdge entries kept in
    place; components/omega/src/ocn/HorzMesh.cpp:581-602 EdgeMask = 0 on edges with a missing cell):

    * an edge survives if at least one of its cells does, a vertex if at least one of its cells does;
    * ``cellsOnCell`` / ``cellsOnEdge`` / ``cellsOnVertex`` / ``edgesOnVertex`` keep their slots, a removed
      neighbour becomes -1 (0 in the 1-based file convention); every edge and vertex of a surviving cell survives;
    * ``first_cell_valid`` (the culler's convention): a boundary edge whose first cell was removed has its cells and
      its vertices exchanged, so the surviving cell comes first and the normal still points from cell 0 outwards;
      ``angleEdge`` and the ``weightsOnEdge`` entries that involve the edge are flipped with it.  With False the
      edge keeps [missing, cell] (the pattern a partition's outer rim also produces);
    * ``edgesOnEdge``: a removed edge leaves a hole IN PLACE (weight kept, it multiplies the zero sentinel row) --
      or, with ``compact_edges_on_edge``, the surviving entries are moved up and ``nEdgesOnEdge`` reduced.

    Isolated cells, one-cell-wide straits and lakes are all legal results.  Adds ``boundaryEdge`` (0/1) and
    ``cullCellMap`` (old index of every new cell)."""
    keep = np.asarray(keep, dtype=bool)
    nC, nE, nV = mesh["nCells"], mesh["nEdges"], mesh["nVertices"]
    assert keep.shape == (nC,) and keep.any()

    def mapping(kept):
        m = np.full(len(kept) + 1, -1, dtype=np.int64)      # index -1 (missing) -> -1
        m[:-1][kept] = np.arange(int(kept.sum()))
        return m

    cmap = mapping(keep)
    coe_old = mesh["cellsOnEdge"].astype(np.int64)
    cov_old = mesh["cellsOnVertex"].astype(np.int64)
    ekeep = (cmap[coe_old] >= 0).any(axis=1)
    vkeep = (cmap[cov_old] >= 0).any(axis=1)
    emap, vmap = mapping(ekeep), mapping(vkeep)
    kept = {"Cell": keep, "Edge": ekeep, "Vertex": vkeep}
    maps = {"Cell": cmap, "Edge": emap, "Vertex": vmap}

    owner = {"nEdgesOnCell": "Cell", "cellsOnCell": "Cell", "edgesOnCell": "Cell", "verticesOnCell": "Cell",
             "areaCell": "Cell", "fCell": "Cell", "bottomDepth": "Cell", "cellsOnEdge": "Edge",
             "verticesOnEdge": "Edge", "edgesOnEdge": "Edge", "weightsOnEdge": "Edge", "nEdgesOnEdge": "Edge",
             "angleEdge": "Edge", "dcEdge": "Edge", "dvEdge": "Edge", "fEdge": "Edge", "cellsOnVertex": "Vertex",
             "edgesOnVertex": "Vertex", "kiteAreasOnVertex": "Vertex", "areaTriangle": "Vertex", "fVertex": "Vertex"}
    for el in ("Cell", "Edge", "Vertex"):
        for pre in ("x", "y", "z", "lon", "lat"):
            owner[pre + el] = el
    target = {"cellsOnCell": "Cell", "edgesOnCell": "Edge", "verticesOnCell": "Vertex", "cellsOnEdge": "Cell",
              "verticesOnEdge": "Vertex", "edgesOnEdge": "Edge", "cellsOnVertex": "Cell", "edgesOnVertex": "Edge"}
    out = {k: v for k, v in mesh.items() if k not in owner}
    for name, own in owner.items():
        if name not in mesh:
            continue
        a = mesh[name][kept[own]]
        if name in target:
            a = maps[target[name]][a.astype(np.int64)].astype(I4)
        out[name] = np.ascontiguousarray(a)
    out["nCells"], out["nEdges"], out["nVertices"] = int(keep.sum()), int(ekeep.sum()), int(vkeep.sum())
    out["cullCellMap"] = np.nonzero(keep)[0].astype(I4)
    # every edge / vertex of a surviving cell survives
    n = out["nEdgesOnCell"]
    live = np.arange(mesh["maxEdges"])[None, :] < n[:, None]
    assert (out["edgesOnCell"][live] >= 0).all() and (out["verticesOnCell"][live] >= 0).all()

    coe = out["cellsOnEdge"]
    out["boundaryEdge"] = (coe < 0).any(axis=1).astype(I4)
    if first_cell_valid:
        flip = np.nonzero(coe[:, 0] < 0)[0]
        coe[flip] = coe[flip][:, ::-1]
        out["verticesOnEdge"][flip] = out["verticesOnEdge"][flip][:, ::-1]
        ang = out["angleEdge"]
        ang[flip] = np.where(ang[flip] > 0, ang[flip] - np.pi, ang[flip] + np.pi)
        # u_e -> -u_e on the flipped edges: row e (tangent k x n flips) and every entry that refers to e
        sgn = np.ones(out["nEdges"] + 1)
        sgn[flip] = -1.0
        w = out["weightsOnEdge"]
        w *= sgn[:-1, None]
        w *= sgn[out["edgesOnEdge"].astype(np.int64)]        # holes (-1) pick the trailing +1
    if compact_edges_on_edge:
        eoe, w = out["edgesOnEdge"], out["weightsOnEdge"]
        order = np.argsort(eoe < 0, axis=1, kind="stable")   # surviving entries first, original order
        out["edgesOnEdge"] = np.take_along_axis(eoe, order, axis=1)
        w = np.take_along_axis(w, order, axis=1)
        out["nEdgesOnEdge"] = (out["edgesOnEdge"] >= 0).sum(axis=1).astype(I4)
        w[out["edgesOnEdge"] < 0] = 0.0
        out["weightsOnEdge"] = w
    return out


def coast_mask(mesh: dict, kind: str, seed: int = 5) -> np.ndarray:
    """``keep`` masks for :func:`cull`: land shapes that produce every coastal pattern the kernels have to cope with.

    ``island``   a disc of land (and a second, single-cell island);
    ``channel``  planar: two rows of land (walls of a zonal channel); sphere: land poleward of 60 degrees;
    ``strait``   a wall with a one-cell-wide gap in it, plus a wall one cell away from it (a one-cell-wide channel);
    ``lakes``    a land mass holding a one-cell lake, a two-cell lake and a bay one cell wide;
    ``ragged``   random land cells, 18 % of the mesh (isolated cells, diagonal contacts, every vertex pattern);
    ``mixed``    island + ragged patches: the default for parity tests;
    ``continents`` a smooth random field thresholded at 28 % land plus a sprinkle of one-cell islands: the shape of a
                 real ocean mesh (bench.py's culled workloads)."""
    nC = mesh["nCells"]
    if mesh.get("on_a_sphere", False):
        R = mesh["sphere_radius"]
        p = np.stack([mesh["xCell"], mesh["yCell"], mesh["zCell"]], axis=1) / R

        def dist(q):                                   # great-circle distance / mean cell spacing
            q = np.asarray(q, dtype=np.float64)
            q /= np.linalg.norm(q)
            return np.arccos(np.clip(p @ q, -1, 1)) / np.sqrt(4 * np.pi / nC)
        d1, d2, d3 = dist([1, 0.2, 0.3]), dist([-0.5, 1, -0.2]), dist([0, -1, 0.1])
        lat = mesh["latCell"]
        polar = np.abs(lat) > np.pi / 3
        lon = mesh["lonCell"]
        wall = (np.abs(lon - np.pi) < 1.2 * np.sqrt(4 * np.pi / nC)) & (np.abs(lat) < 1.0)
        gap = wall & (np.abs(lat - 0.2) < 0.6 * np.sqrt(4 * np.pi / nC))
    else:
        dc = mesh["dc"]
        x, y = mesh["xCell"] / dc, mesh["yCell"] / (dc * np.sqrt(3.0) / 2.0)
        nx, ny = mesh["x_period"] / dc, mesh["y_period"] / (dc * np.sqrt(3.0) / 2.0)

        def dist(q):
            dx = np.abs(x - q[0] * nx)
            dy = np.abs(y - q[1] * ny)
            dx, dy = np.minimum(dx, nx - dx), np.minimum(dy, ny - dy) * np.sqrt(3.0) / 2.0
            return np.hypot(dx, dy)
        d1, d2, d3 = dist([0.3, 0.35]), dist([0.7, 0.6]), dist([0.55, 0.15])
        row = np.rint(y).astype(np.int64)
        polar = (row % int(ny) == 1) | (row % int(ny) == int(ny) // 2 + 1)
        col = np.rint(x - 0.25).astype(np.int64)
        wall = col == int(nx) // 2
        gap = wall & (row == int(ny) // 3)
    rng = np.random.default_rng(seed)
    rnd = rng.random(nC)
    rad = max(2.0, 0.12 * np.sqrt(nC))
    if kind == "continents":
        f = np.zeros(nC)
        for _ in range(12):
            if mesh.get("on_a_sphere", False):
                q = rng.normal(size=3)
                q /= np.linalg.norm(q)
                f += rng.uniform(0.5, 1.0) * np.cos(rng.integers(1, 5) * np.arccos(np.clip(p @ q, -1, 1)) + rng.uniform(0, 6.3))
            else:
                kx, ky = rng.integers(-3, 4), rng.integers(-3, 4)
                f += rng.uniform(0.5, 1.0) * np.cos(2 * np.pi * (kx * x / nx + ky * y / ny) + rng.uniform(0, 6.3))
        land = (f > np.quantile(f, 0.72)) | (rnd < 0.002)
        keep = ~land
        return keep
    if kind == "island":
        land = (d1 < rad) | (d2 < 0.6)
    elif kind == "channel":
        land = polar
    elif kind == "strait":
        wall2 = np.roll(wall, 2) if not mesh.get("on_a_sphere", False) else (d3 < 1.5)
        land = (wall & ~gap) | wall2
    elif kind == "lakes":
        land = d1 < 1.6 * rad
        order = np.argsort(d1)
        land[order[0]] = False                                   # one-cell lake at the centre
        ring = order[(d1[order] > 0.5 * rad) & (d1[order] < 0.5 * rad + 1.2)]
        land[ring[:2]] = False                                   # a two-cell lake (or two single ones)
        land[(d2 < rad) & (rnd < 0.5)] = True                    # a porous land mass: bays and channels one cell wide
    elif kind == "ragged":
        land = rnd < 0.18
    elif kind == "mixed":
        land = (d1 < rad) | (d2 < 0.6) | ((d3 < 1.5 * rad) & (rnd < 0.3))
    else:
        raise ValueError(kind)
    keep = ~land
    assert keep.any()
    return keep


def zero_boundary_velocity(mesh: dict, u: np.ndarray) -> np.ndarray:
    """No-normal-flow condition on the coast (what an ocean initial state satisfies): u = 0 on boundary edges."""
    if "boundaryEdge" in mesh:
        u = u.copy()
        u[mesh["boundaryEdge"] != 0] = 0.0
    return u
