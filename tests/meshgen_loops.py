"""The round 1-4 spherical mesh generator, element by element in Python loops: kept as the REFERENCE of the vectorised
omega_amd.meshgen.spherical_voronoi (tests/test_meshgen_rows.py requires the same arrays bit for bit -- the committed
golden vectors of tests/golden/ico2_k6_nt2.npz were made on this generator's mesh)."""
import numpy as np

from omega_amd.meshgen import I4, _arc, _sph_tri_area, _trisk_edges_on_edge


def spherical_voronoi_loops(n_cells: int = 0, *, points=None, radius: float = 6371220.0, lloyd: int = 6,
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

    def regions_ccw(sv):
        sv.sort_vertices_of_regions()
        out = []
        for c, reg in enumerate(sv.regions):
            v = sv.vertices[reg]
            p = sv.points[c]
            if np.dot(np.cross(v[0] - p, v[1] - p), p) < 0:
                reg = reg[::-1]
            out.append(list(reg))
        return out

    for _ in range(lloyd):
        sv = SphericalVoronoi(pts, 1.0)
        new = np.empty_like(pts)
        for c, reg in enumerate(regions_ccw(sv)):
            v = sv.vertices[reg]
            p = pts[c]
            w = _sph_tri_area(p[None], v, np.roll(v, -1, axis=0))
            cen = ((p[None] + v + np.roll(v, -1, axis=0)) * w[:, None]).sum(0)
            new[c] = cen / np.linalg.norm(cen)
        pts = new
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
    regs = regions_ccw(sv)
    xv = sv.vertices
    nC, nV = n_cells, len(xv)
    maxE = max(len(rg) for rg in regs)
    nEoc = np.array([len(rg) for rg in regs], dtype=I4)
    eoc = np.full((nC, maxE), -1, dtype=I4)
    voc = np.full((nC, maxE), -1, dtype=I4)
    coc = np.full((nC, maxE), -1, dtype=I4)
    edge_of = {}
    coe, voe = [], []
    for c, rg in enumerate(regs):
        n = len(rg)
        for j in range(n):
            va, vb = rg[j], rg[(j + 1) % n]
            key = (va, vb) if va < vb else (vb, va)
            e = edge_of.get(key)
            if e is None:
                e = len(coe)
                edge_of[key] = e
                coe.append([c, -1])
                voe.append([va, vb])     # CCW around cell 0 == along k x n
            else:
                coe[e][1] = c
            eoc[c, j] = e
            voc[c, j] = vb               # vertex between edge j and edge j+1
    coe = np.array(coe, dtype=I4)
    voe = np.array(voe, dtype=I4)
    nE = len(coe)
    assert (coe >= 0).all() and nE == nC + nV - 2, "not a closed Voronoi tessellation"
    for c in range(nC):
        for j in range(nEoc[c]):
            e = eoc[c, j]
            coc[c, j] = coe[e, 1] if coe[e, 0] == c else coe[e, 0]
    # vertices: the three edges / cells around, counter-clockwise
    eov_l = [[] for _ in range(nV)]
    for e in range(nE):
        eov_l[voe[e, 0]].append(e)
        eov_l[voe[e, 1]].append(e)
    xe = pts[coe[:, 0]] + pts[coe[:, 1]]
    xe /= np.linalg.norm(xe, axis=1)[:, None]
    cov = np.empty((nV, 3), dtype=I4)
    eov = np.empty((nV, 3), dtype=I4)
    for v in range(nV):
        es = eov_l[v]
        assert len(es) == 3, "degenerate Voronoi vertex"
        p = xv[v]
        ref = xe[es[0]] - p
        ang = []
        for e in es:
            d = xe[e] - p
            ang.append(np.arctan2(np.dot(np.cross(ref, d), p), np.dot(ref, d)) % (2 * np.pi))
        es = [es[k] for k in np.argsort(ang)]
        # cell k lies between edge k-1 and edge k (CCW): the cell shared by both
        for k in range(3):
            ea, eb = es[(k + 2) % 3], es[k]
            sh = set(coe[ea]) & set(coe[eb])
            assert len(sh) == 1
            cov[v, k] = sh.pop()
        eov[v] = es
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
    area = np.zeros(nC)
    for c, rg in enumerate(regs):
        v = xv[rg]
        area[c] = _sph_tri_area(pts[c][None], v, np.roll(v, -1, axis=0)).sum()
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
