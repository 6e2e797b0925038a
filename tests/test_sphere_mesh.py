"""Spherical Voronoi test meshes (omega_amd/meshgen.py: stand-ins for the downloadable QU meshes):
MPAS conventions, and the mimetic identities of the TRiSK operators evaluated by the oracle on them
(these fail if an orientation convention -- cellsOnEdge / verticesOnEdge / edge signs -- is off)."""
import ctypes as C

import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import icosahedral_points, spherical_voronoi
from oracle import oracle as O


@pytest.fixture(scope="module", params=["ico3", "fib900"])
def mesh(request):
    if request.param.startswith("ico"):
        return spherical_voronoi(points=icosahedral_points(3), lloyd=2)
    return spherical_voronoi(900, lloyd=4)


def test_topology_and_geometry(mesh):
    m = mesh
    nC, nE, nV = m["nCells"], m["nEdges"], m["nVertices"]
    assert nC - nE + nV == 2 and nV == 2 * nC - 4
    R = m["sphere_radius"]
    assert abs(m["areaCell"].sum() / (4 * np.pi * R * R) - 1) < 1e-12
    assert abs(m["areaTriangle"].sum() / (4 * np.pi * R * R) - 1) < 1e-12
    assert abs(m["kiteAreasOnVertex"].sum() / (4 * np.pi * R * R) - 1) < 1e-3
    # every edge appears once on each of its two cells; cellsOnCell is the cell across
    for c in range(0, nC, 7):
        for j in range(m["nEdgesOnCell"][c]):
            e = m["edgesOnCell"][c, j]
            assert c in m["cellsOnEdge"][e]
            assert m["cellsOnCell"][c, j] == m["cellsOnEdge"][e].sum() - c
    # verticesOnEdge 0 -> 1 runs along k x n (n from cell 0 to cell 1)
    xc = np.stack([m["xCell"], m["yCell"], m["zCell"]], 1)
    xv = np.stack([m["xVertex"], m["yVertex"], m["zVertex"]], 1)
    xe = np.stack([m["xEdge"], m["yEdge"], m["zEdge"]], 1)
    n = xc[m["cellsOnEdge"][:, 1]] - xc[m["cellsOnEdge"][:, 0]]
    t = xv[m["verticesOnEdge"][:, 1]] - xv[m["verticesOnEdge"][:, 0]]
    assert np.all(np.einsum("ij,ij->i", np.cross(n, t), xe) > 0)
    # edgesOnVertex[k] separates cellsOnVertex[k] and [k+1]
    for k in range(3):
        e = m["edgesOnVertex"][:, k]
        a, b = m["cellsOnVertex"][:, k], m["cellsOnVertex"][:, (k + 1) % 3]
        assert np.all(np.sort(m["cellsOnEdge"][e], axis=1) == np.sort(np.stack([a, b], 1), axis=1))
    # TRiSK stencil: n-1 edges per side
    assert np.array_equal(m["nEdgesOnEdge"],
                          m["nEdgesOnCell"][m["cellsOnEdge"][:, 0]] + m["nEdgesOnCell"][m["cellsOnEdge"][:, 1]] - 2)


def test_mimetic_identities_through_the_oracle(mesh):
    """curl(grad phi) = 0 on vertices and sum(div(F) * area) = 0, to rounding."""
    K = 2
    M = O.Mesh.single_rank(mesh, K)
    rng = np.random.default_rng(3)
    nC, nE, nV = mesh["nCells"], mesh["nEdges"], mesh["nVertices"]
    phi = np.zeros((M.NCellsSize, K))
    phi[:nC] = rng.standard_normal((nC, K))
    grad = np.zeros((M.NEdgesSize, K))
    O.lib().orc_gradient_on_edge(C.byref(M.s), M.NEdgesOwned, O._pd(grad), O._pd(phi))
    curl = np.zeros((M.NVerticesSize, K))
    O.lib().orc_curl_on_vertex(C.byref(M.s), M.NVerticesOwned, O._pd(curl), O._pd(grad))
    scale = np.abs(grad[:nE]).max() * mesh["dcEdge"].max() / mesh["areaTriangle"].min()
    assert np.abs(curl[:nV]).max() <= 1e-12 * scale
    F = np.zeros((M.NEdgesSize, K))
    F[:nE] = rng.standard_normal((nE, K))
    div = np.zeros((M.NCellsSize, K))
    O.lib().orc_divergence_on_cell(C.byref(M.s), M.NCellsOwned, O._pd(div), O._pd(F))
    tot = (div[:nC] * mesh["areaCell"][:, None]).sum(0)
    assert np.all(np.abs(tot) <= 1e-10 * (np.abs(div[:nC]) * mesh["areaCell"][:, None]).sum(0))


@pytest.mark.parametrize("nparts", [1, 3, 8])
def test_decomp_on_the_sphere(mesh, nparts):
    gm = oa.GlobalMesh(mesh)
    tot = np.zeros(3, dtype=np.int64)
    for r in range(nparts):
        d = oa.Decomp(gm, nparts, r, 3)
        for i, (arr, n) in enumerate((("CellID", "NCellsOwned"), ("EdgeID", "NEdgesOwned"), ("VertexID", "NVerticesOwned"))):
            tot[i] += d.get_array(arr)[: d.get_int(n)].astype(np.int64).sum()
    n = np.array([mesh["nCells"], mesh["nEdges"], mesh["nVertices"]], dtype=np.int64)
    assert np.array_equal(tot, n * (n + 1) // 2)


def test_rhs_is_finite_on_the_sphere(mesh):
    from tests.problem import Problem
    P = Problem(mesh, 4, 2, device=False)
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    assert np.isfinite(hT).all() and np.isfinite(uT).all() and np.isfinite(trT).all()
    assert np.abs(uT).max() > 0 and np.abs(hT).max() > 0
