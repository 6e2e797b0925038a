"""HorzOperators as product kernels (omega_amd/csrc/kernels/HorzOperators.hip behind omg_horz_*): the reference's
own known answers (O/test/ocn/HorzOperatorsTest.cpp:33-44, RTol 1e-10 :475, planar 48x48 mesh, 16 levels) through
the device arrays, and bit-exact equality with the CPU oracle's functors on planar and spherical meshes."""
import ctypes as C

import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import icosahedral_points, planar_hex, spherical_voronoi
from oracle import oracle as O
from tests.ka_common import check_errors, compute_errors, set_scalar, set_scalar_1d, set_vector_edge
from tests.problem import Problem
from tests.test_oracle_known_answers import HO_RTOL, curl, divergence, gradX, gradY, scalar, vecX, vecY

pytestmark = pytest.mark.gpu
K = 16


@pytest.fixture(scope="module")
def P():
    assert oa.device_count() > 0, "no HIP device"
    oa.device_init(0)
    return Problem(planar_hex(48, 48, 1.0 / 48.0), K, 1)


def test_divergence_known_answer(P):
    M, ops = P.omesh, oa.HorzOperators(P.mesh)
    num = ops.divergence(set_vector_edge(M, K, vecX, vecY), M.NCellsOwned)
    check_errors("Divergence", compute_errors(M, num, set_scalar(M, K, divergence, "Cell"), "Cell"),
                 (0.00124886886594427027, 0.00124886886590974385), HO_RTOL)


def test_gradient_known_answer(P):
    M, ops = P.omesh, oa.HorzOperators(P.mesh)
    num = ops.gradient(set_scalar(M, K, scalar, "Cell"), M.NEdgesOwned)
    check_errors("Gradient", compute_errors(M, num, set_vector_edge(M, K, gradX, gradY), "Edge"),
                 (0.00125026071878537952, 0.00134354611117262204), HO_RTOL)


def test_curl_known_answer(P):
    M, ops = P.omesh, oa.HorzOperators(P.mesh)
    num = ops.curl(set_vector_edge(M, K, vecX, vecY), M.NVerticesOwned)
    check_errors("Curl", compute_errors(M, num, set_scalar(M, K, curl, "Vertex"), "Vertex"),
                 (0.161365663569699946, 0.161348016897141039), HO_RTOL)


def test_tangential_recon_known_answer(P):
    M, ops = P.omesh, oa.HorzOperators(P.mesh)
    num = ops.tangential_recon(set_vector_edge(M, K, vecX, vecY), M.NEdgesOwned)
    check_errors("Recon", compute_errors(M, num, set_vector_edge(M, K, vecX, vecY, "Tangential"), "Edge"),
                 (0.00450897496974901352, 0.00417367308684470691), HO_RTOL)


def test_interp_cell_to_edge_known_answer(P):
    M, ops = P.omesh, oa.HorzOperators(P.mesh)
    sc, exact = set_scalar_1d(M, scalar, "Cell"), set_scalar_1d(M, scalar, "Edge")
    # (values of the oracle pin tests/test_oracle_known_answers.py::test_ho_interp; the reference's
    # HorzOperatorsTest has no interpolation entry, AuxiliaryVarsTest covers it through the wind stress)
    for iso, exp in ((0, (0.0026762081503380526, 0.003058198461518835)),
                     (1, (0.004279097382993937, 0.004200067675522098))):
        num = ops.interp_cell_to_edge(sc, bool(iso), M.NEdgesOwned)
        check_errors("Interp%d" % iso, compute_errors(M, num, exact, "Edge"), exp, HO_RTOL)


@pytest.mark.parametrize("mesh,k", [("hex", 5), ("hex", 80), ("ico3", 12), ("fib400", 6)])
def test_operators_equal_the_oracle_bit_for_bit(mesh, k):
    """Random fields, all local elements, odd / even level counts, pentagons and heptagons."""
    oa.device_init(0)
    g = (planar_hex(20, 16, 30e3) if mesh == "hex" else
         spherical_voronoi(points=icosahedral_points(3), lloyd=2) if mesh == "ico3" else spherical_voronoi(400, lloyd=4))
    Q = Problem(g, k, 1)
    m, M, ops = Q.mesh, Q.omesh, oa.HorzOperators(Q.mesh)
    rng = np.random.default_rng(11)
    vec = np.zeros((m.NEdgesSize, k))
    vec[:-1] = rng.standard_normal((m.NEdgesAll, k))
    sc = np.zeros((m.NCellsSize, k))
    sc[:-1] = rng.standard_normal((m.NCellsAll, k))
    L, pd = O.lib(), O._pd

    def orc(name, n, rows, x, *extra):
        out = np.zeros((rows, k) if x.ndim == 2 else (rows,))
        getattr(L, name)(C.byref(M.s), n, pd(out), pd(x), *extra)
        return out
    assert np.array_equal(ops.divergence(vec), orc("orc_divergence_on_cell", m.NCellsAll, m.NCellsSize, vec))
    assert np.array_equal(ops.gradient(sc), orc("orc_gradient_on_edge", m.NEdgesAll, m.NEdgesSize, sc))
    assert np.array_equal(ops.curl(vec), orc("orc_curl_on_vertex", m.NVerticesAll, m.NVerticesSize, vec))
    assert np.array_equal(ops.tangential_recon(vec), orc("orc_tangential_recon_on_edge", m.NEdgesAll, m.NEdgesSize, vec))
    for iso in (0, 1):
        assert np.array_equal(ops.interp_cell_to_edge(sc[:, 0].copy(), bool(iso)),
                              orc("orc_interp_cell_to_edge", m.NEdgesAll, m.NEdgesSize, sc[:, 0].copy(), iso))
