"""The reference's spherical TENDENCY-TERM and AUXILIARY-VARIABLE known answers, as far as they can be reached offline.

Same situation and same method as tests/test_oracle_sphere_neighbourhood.py: the norms hard-coded in
TendencyTermsTest.cpp:188-204 and AuxiliaryVarsTest.cpp:160-194 (TestSetupSphere) were produced on the Icos480 download;
this repo's icosahedral mesh of the same size must land near them with the same analytic fields and the same harness.
Measured (printed with `-s`): of the 24 norm pairs, the L2 norms of 17 land within 6 % of the reference's values, all
within 0.71 ... 1.54 x; the max norms (which sit at the twelve pentagons) within 0.36 ... 1.86 x.  The bands below are
set around that, so the test pins the oracle's spherical path to "the same discretisation error as the reference's" --
not to its digits, which need the reference's mesh file.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from tests.ka_common import compute_errors
from tests.meshes import named_mesh
from tests.test_oracle_sphere_neighbourhood import R, set_vector_edge as _sve

K = 2
NT = 3
cos, sin = np.cos, np.sin
pd = O._pd


@pytest.fixture(scope="module")
def M():
    return O.Mesh.single_rank(named_mesh("ico4"), K)


def call(name, *a):
    getattr(O.lib(), name)(*a)


def vec_edge(M, fx, fy, comp="Normal"):
    return _sve(M, fx, fy, comp)


def sc(M, f, el, ntr=None, rows=None):
    own = M.a["N" + {"Cell": "Cells", "Edge": "Edges", "Vertex": "Vertices"}[el] + "Owned"]
    size = rows if rows is not None else M.a["N" + {"Cell": "Cells", "Edge": "Edges", "Vertex": "Vertices"}[el] + "Size"]
    vals = f(M.a["Lon" + el][:own], M.a["Lat" + el][:own])
    out = np.zeros(((ntr,) if ntr else ()) + (size, K))
    out[..., :own, :] = vals[:, None]
    return out


def vec_edge_1d(M, fx, fy):
    return np.ascontiguousarray(_sve(M, fx, fy)[:, 0])


def ratios(name, got, ref):
    r = (got[0] / ref[0] if ref[0] else got[0], got[1] / ref[1] if ref[1] else got[1])
    print(f"[sphere] {name:22s} LInf {got[0]:.4e} (x{r[0]:.2f})  L2 {got[1]:.4e} (x{r[1]:.2f})")
    return r


def near(name, got, ref, l2_band=(0.6, 1.7), linf_band=(0.3, 3.0)):
    rinf, r2 = ratios(name, got, ref)
    assert l2_band[0] <= r2 <= l2_band[1], f"{name} L2 {got[1]:.4e} against the reference's {ref[1]:.4e} on Icos480"
    assert linf_band[0] <= rinf <= linf_band[1], f"{name} LInf {got[0]:.4e} against the reference's {ref[0]:.4e} on Icos480"


# ============ TendencyTermsTest.cpp:206-325 (TestSetupSphere fields) ============
def vecX(lon, lat):
    return -R * sin(lon) ** 2 * cos(lat) ** 3


def vecY(lon, lat):
    return -4 * R * sin(lon) * cos(lon) * cos(lat) ** 3 * sin(lat)


def divergence(lon, lat):
    return sin(lon) * cos(lon) * cos(lat) ** 2 * (20 * sin(lat) ** 2 - 6)


def scalar(lon, lat):
    return R * cos(lon) * cos(lat) ** 4


def gradX(lon, lat):
    return -sin(lon) * cos(lat) ** 3


def gradY(lon, lat):
    return -4 * cos(lon) * cos(lat) ** 3 * sin(lat)


def curl(lon, lat):
    return -4 * cos(lon) ** 2 * cos(lat) ** 2 * sin(lat)


def lapX(lon, lat):
    return cos(lat) * (sin(lat) ** 2 * (17 - 37 * sin(lon) ** 2) + 11 * sin(lon) ** 2 - 5) / R


def lapY(lon, lat):
    return sin(lon) * cos(lon) * sin(lat) * cos(lat) * (96 * cos(lat) ** 2 - 22) / R


def layerThick(lon, lat):
    return 2 + cos(lon) * cos(lat) ** 4


def normRelVort(lon, lat):
    return curl(lon, lat) / layerThick(lon, lat)


def normPlanetVort(lon, lat):
    return sin(lat) / layerThick(lon, lat)


def tracerFluxDiv(lon, lat):
    return sin(lon) * cos(lat) ** 2 * (cos(lon) * (8 - 20 * cos(2 * lat))
                                       - 6 * cos(lon) ** 2 * cos(lat) ** 4 * (-2 + 3 * cos(2 * lat))
                                       + cos(lat) ** 4 * sin(lon) ** 2)


def scalarA(lon, lat):
    return R * sin(lon) ** 2 * cos(lat) ** 2


def scalarB(lon, lat):
    return 2. + cos(lon) * sin(lat)


def tracerDiff(lon, lat):
    return (4 * cos(lon) ** 2 - 2 * (1. + 3 * cos(2 * lat)) * sin(lon) ** 2 + 2 * cos(lon) ** 3 * sin(lat)
            - 8 * cos(lon) * cos(lat) ** 2 * sin(lon) ** 2 * sin(lat)) / R


SQ = np.sqrt((3 // 2) / np.pi)     # the reference writes std::sqrt(3 / 2 / Pi): integer 3 / 2 (:297, :302)


def scalarC(lon, lat):
    return -(R / 2) * SQ * cos(lat) * cos(lon)


def tracerHyperDiff(lon, lat):
    return SQ * cos(lat) * cos(lon) / R


def test_tt_thick_flux_div(M):
    flux = vec_edge(M, vecX, vecY)
    ones = np.ones((M.NEdgesSize, K))
    num = np.zeros((M.NCellsOwned, K))
    call("orc_thickness_flux_div_on_cell", C.byref(M.s), M.NCellsOwned, pd(num), pd(ones), pd(flux))
    near("ThickFluxDiv", compute_errors(M, num, sc(M, lambda a, b: -divergence(a, b), "Cell"), "Cell"),
         (0.0136595773989796766, 0.00367052484586384131))


def test_tt_pot_vort_hadv(M):
    def ex(f):
        return lambda a, b: (normRelVort(a, b) + normPlanetVort(a, b)) * layerThick(a, b) * f(a, b)
    exact = vec_edge(M, ex(vecX), ex(vecY), "Tangential")
    num = np.zeros((M.NEdgesOwned, K))
    call("orc_pv_hadv_on_edge", C.byref(M.s), M.NEdgesOwned, pd(num), pd(sc(M, normRelVort, "Edge")),
         pd(sc(M, normPlanetVort, "Edge")), pd(sc(M, layerThick, "Edge")), pd(vec_edge(M, vecX, vecY)))
    near("PotVortHAdv", compute_errors(M, num, exact, "Edge"), (0.0219217796608757037, 0.0122537418367830303))


def test_tt_ke_and_ssh_grad(M):
    ke = sc(M, scalar, "Cell")
    for fn, g, nm in (("orc_ke_grad_on_edge", 1.0, "KEGrad"), ("orc_ssh_grad_on_edge", 9.80665, "SSHGrad")):
        exact = vec_edge(M, lambda a, b: -g * gradX(a, b), lambda a, b: -g * gradY(a, b))
        num = np.zeros((M.NEdgesOwned, K))
        call(fn, C.byref(M.s), M.NEdgesOwned, pd(num), pd(ke))
        near(nm, compute_errors(M, num, exact, "Edge"), (0.00187912292540623471, 0.00149841802817334935),
             l2_band=(0.9, 1.1), linf_band=(0.8, 1.25))


def test_tt_vel_diff_and_hyper_diff(M):
    dv = sc(M, divergence, "Cell")
    rv = sc(M, curl, "Vertex")
    ref = (0.281930203304510130, 0.270530313560271740)
    visc = 1.0e3
    exact = vec_edge(M, lambda a, b: visc * lapX(a, b), lambda a, b: visc * lapY(a, b))
    num = np.zeros((M.NEdgesOwned, K))
    call("orc_velocity_diffusion_on_edge", C.byref(M.s), M.NEdgesOwned, pd(num), pd(dv), pd(rv), C.c_double(visc))
    near("VelDiff", compute_errors(M, num, exact, "Edge"), ref)
    visc4 = 1.2e11
    exact = vec_edge(M, lambda a, b: -visc4 * lapX(a, b), lambda a, b: -visc4 * lapY(a, b))
    num = np.zeros((M.NEdgesOwned, K))
    call("orc_velocity_hyperdiff_on_edge", C.byref(M.s), M.NEdgesOwned, pd(num), pd(dv), pd(rv), C.c_double(visc4),
         C.c_double(1.0))
    near("VelHyperDiff", compute_errors(M, num, exact, "Edge"), ref)


def test_tt_wind_forcing_and_bottom_drag(M):
    rho = 0.987654321
    exact = vec_edge(M, lambda a, b: vecX(a, b) / (scalarB(a, b) * rho), lambda a, b: vecY(a, b) / (scalarB(a, b) * rho))
    exact = exact[: M.NEdgesOwned].copy()
    exact[:, 1:] = 0
    num = np.zeros((M.NEdgesOwned, K))
    call("orc_wind_forcing_on_edge", C.byref(M.s), M.NEdgesOwned, pd(num), pd(vec_edge_1d(M, vecX, vecY)),
         pd(sc(M, scalarB, "Edge")), C.c_double(rho))
    linf, l2 = compute_errors(M, num, exact, "Edge")
    assert linf <= 100 * np.finfo(float).eps and l2 <= 100 * np.finfo(float).eps      # reference: 0 / 0, ATol 100 eps

    coeff = 1.123456789
    exact = vec_edge(M, lambda a, b: -coeff * np.abs(scalarA(a, b)) / scalarB(a, b) * vecX(a, b),
                     lambda a, b: -coeff * np.abs(scalarA(a, b)) / scalarB(a, b) * vecY(a, b))
    exact = exact[: M.NEdgesOwned].copy()
    exact[:, :-1] = 0
    num = np.zeros((M.NEdgesOwned, K))
    call("orc_bottom_drag_on_edge", C.byref(M.s), M.NEdgesOwned, pd(num), pd(vec_edge(M, vecX, vecY)),
         pd(sc(M, lambda a, b: scalarA(a, b) ** 2 / 2, "Cell")), pd(sc(M, scalarB, "Edge")), C.c_double(coeff))
    near("BottomDrag", compute_errors(M, num, exact, "Edge"), (0.0015333449035655053, 0.0014897009917655022))


def test_tt_tracer_terms(M):
    nv = vec_edge(M, vecX, vecY)
    num = np.zeros((NT, M.NCellsOwned, K))
    call("orc_tracer_horz_adv_on_cell", C.byref(M.s), NT, M.NCellsOwned, pd(num), M.NCellsOwned, pd(nv),
         pd(sc(M, lambda a, b: -layerThick(a, b), "Edge", ntr=NT)))
    near("TracerHorzAdv", compute_errors(M, num, sc(M, tracerFluxDiv, "Cell", ntr=NT, rows=M.NCellsOwned), "Cell"),
         (0.0132310202299444034, 0.0038523368564029538))
    num = np.zeros((NT, M.NCellsOwned, K))
    call("orc_tracer_diff_on_cell", C.byref(M.s), NT, M.NCellsOwned, pd(num), M.NCellsOwned,
         pd(sc(M, scalarA, "Cell", ntr=NT)), pd(sc(M, scalarB, "Edge")), C.c_double(1.0))
    near("TracerDiff", compute_errors(M, num, sc(M, tracerDiff, "Cell", ntr=NT, rows=M.NCellsOwned), "Cell"),
         (0.0486107109846934185, 0.00507514214194892694))
    num = np.zeros((NT, M.NCellsOwned, K))
    call("orc_tracer_hyperdiff_on_cell", C.byref(M.s), NT, M.NCellsOwned, pd(num), M.NCellsOwned,
         pd(sc(M, scalarC, "Cell", ntr=NT)), C.c_double(1.0))
    near("TracerHyperDiff", compute_errors(M, num, sc(M, lambda a, b: -tracerHyperDiff(a, b), "Cell", ntr=NT,
                                                      rows=M.NCellsOwned), "Cell"),
         (0.000819552466009620408, 0.00064700084412871962))


# ============ AuxiliaryVarsTest.cpp:160-300 (TestSetupSphere fields) ============
def velX(lon, lat):
    return -sin(lon) ** 2 * cos(lat) ** 3


def velY(lon, lat):
    return -4 * sin(lon) * cos(lon) * cos(lat) ** 3 * sin(lat)


def relVort(lon, lat):
    return curl(lon, lat) / R


def divU(lon, lat):
    return divergence(lon, lat) / R


def del2X(lon, lat):
    return (1 / (R * R) * (cos(lon) ** 2 - sin(lon) ** 2) * cos(lat) * (20 * sin(lat) ** 2 - 6)
            + 4 / (R * R) * cos(lon) ** 2 * (cos(lat) ** 3 - 2 * cos(lat) * sin(lat) ** 2))


def del2Y(lon, lat):
    return (1 / (R * R) * sin(lon) * cos(lon) * sin(lat) * cos(lat) * (80 * cos(lat) ** 2 - 28)
            + 8 / (R * R) * sin(lon) * cos(lon) * sin(lat) * cos(lat))


def del2Div(lon, lat):
    return 1 / R ** 3 * (-2 * sin(lon) * cos(lon) * (28 * sin(lat) ** 2 - 8)
                         + sin(lon) * cos(lon) * ((cos(lat) ** 2 - 2 * sin(lat) ** 2) * (80 * cos(lat) ** 2 - 20)
                                                  - 160 * (sin(lat) * cos(lat)) ** 2))


def del2Curl(lon, lat):
    return 1 / R ** 3 * (-sin(lat) * (cos(lat) ** 2 * (56 * cos(lon) ** 2 - 40)
                                      - 2 * (cos(lon) ** 2 * (28 * sin(lat) ** 2 - 8) - 20 * sin(lat) ** 2 + 6))
                         + sin(lat) * (80 * cos(lat) ** 2 - 20) * (cos(lon) ** 2 - sin(lon) ** 2))


def tracer(lon, lat):
    return 2 - cos(lon) * cos(lat) ** 4


def thickTracer(lon, lat):
    return 4 - cos(lon) ** 2 * cos(lat) ** 8


def del2Tracer(lon, lat):
    return 1 / (R * R) * (10 * cos(lon) * cos(lat) ** 2 * (-1 + 2 * cos(2 * lat))
                          + cos(lon) ** 2 * cos(lat) ** 6 * (-13 + 18 * cos(2 * lat)) - cos(lat) ** 6 * sin(lon) ** 2)


class AVState:
    """initState (AuxiliaryVarsTest.cpp:314-339): h, u, and FVertex = sin(lat)"""

    def __init__(self, M):
        self.h = sc(M, layerThick, "Cell")
        self.u = vec_edge(M, velX, velY)
        n = M.NVerticesOwned
        M.FVertex[:n] = sin(M.LatVertex[:n])
        self.aux = O.Aux(M, NT)


@pytest.fixture(scope="module")
def AV():
    M = O.Mesh.single_rank(named_mesh("ico4"), K)
    return M, AVState(M)


def test_av_kinetic(AV):
    M, S = AV
    call("orc_kinetic_on_cell", C.byref(M.s), M.NCellsOwned, C.byref(S.aux.s), pd(S.u))
    ke = sc(M, lambda a, b: (velX(a, b) ** 2 + velY(a, b) ** 2) / 2, "Cell")
    near("KineticEnergy", compute_errors(M, S.aux["KineticEnergyCell"], ke, "Cell"),
         (0.0143579382532765844, 0.00681096618897046764))
    near("VelocityDiv", compute_errors(M, S.aux["VelocityDivCell"], sc(M, divU, "Cell"), "Cell"),
         (0.0136595773989793799, 0.00367052484586382699))


def test_av_layer_thickness_upwind(AV):
    M, S = AV
    call("orc_layerthick_on_edge", C.byref(M.s), M.NEdgesOwned, C.byref(S.aux.s), pd(S.h), pd(S.u), 1)
    exact = sc(M, layerThick, "Edge")
    near("FluxThick", compute_errors(M, S.aux["FluxLayerThickEdge"], exact, "Edge"), (0.0159821090867812224, 0.010364511516135164))
    near("MeanThick", compute_errors(M, S.aux["MeanLayerThickEdge"], exact, "Edge"),
         (0.000800109287518277435, 0.000406527457820634436))


def test_av_vorticity(AV):
    M, S = AV
    call("orc_vorticity_on_vertex", C.byref(M.s), M.NVerticesOwned, C.byref(S.aux.s), pd(S.h), pd(S.u))
    nrv = lambda a, b: relVort(a, b) / layerThick(a, b)     # noqa: E731
    npv = lambda a, b: sin(b) / layerThick(a, b)            # noqa: E731
    near("RelVortVertex", compute_errors(M, S.aux["RelVortVertex"], sc(M, relVort, "Vertex"), "Vertex"),
         (0.0271404735181343393, 0.0252023166109219786))
    near("NormRelVortVertex", compute_errors(M, S.aux["NormRelVortVertex"], sc(M, nrv, "Vertex"), "Vertex"),
         (0.0348741350737879693, 0.0259506101504540822))
    near("NormPlanetVortVertex", compute_errors(M, S.aux["NormPlanetVortVertex"], sc(M, npv, "Vertex"), "Vertex"),
         (0.00451268952953497778, 0.00101771171197261793))
    call("orc_vorticity_on_edge", C.byref(M.s), M.NEdgesOwned, C.byref(S.aux.s))
    near("NormRelVortEdge", compute_errors(M, S.aux["NormRelVortEdge"], sc(M, nrv, "Edge"), "Edge"),
         (0.0125376497261775952, 0.00307521304930552519))
    near("NormPlanetVortEdge", compute_errors(M, S.aux["NormPlanetVortEdge"], sc(M, npv, "Edge"), "Edge"),
         (0.00495174534686814403, 0.000855432390947949515))


def test_av_velocity_del2(AV):
    M, S = AV
    call("orc_veldel2_on_edge", C.byref(M.s), M.NEdgesOwned, C.byref(S.aux.s), pd(sc(M, divU, "Cell")), pd(sc(M, relVort, "Vertex")))
    near("Del2", compute_errors(M, S.aux["Del2Edge"], vec_edge(M, del2X, del2Y), "Edge"),
         (0.00360406641962622652, 0.00313406628499444213))
    call("orc_veldel2_on_cell", C.byref(M.s), M.NCellsOwned, C.byref(S.aux.s))
    near("Del2Div", compute_errors(M, S.aux["Del2DivCell"], sc(M, del2Div, "Cell"), "Cell"),
         (0.0177782108439020134, 0.00751922684420262138))
    call("orc_veldel2_on_vertex", C.byref(M.s), M.NVerticesOwned, C.byref(S.aux.s))
    near("Del2RelVort", compute_errors(M, S.aux["Del2RelVortVertex"], sc(M, del2Curl, "Vertex"), "Vertex"),
         (0.0915578492503972413, 0.0246736311927726465))


def test_av_tracer_upwind(AV):
    M, S = AV
    tr = sc(M, tracer, "Cell", ntr=NT)
    call("orc_tracer_on_edge", C.byref(M.s), NT, M.NEdgesOwned, C.byref(S.aux.s), pd(S.u), pd(S.h), pd(tr), 1)
    near("HTracers", compute_errors(M, S.aux["HTracersEdge"], sc(M, thickTracer, "Edge", ntr=NT), "Edge"),
         (0.01603249913425972, 0.00546762028673672059))
    call("orc_tracer_on_cell", C.byref(M.s), NT, M.NCellsOwned, C.byref(S.aux.s), pd(sc(M, layerThick, "Edge")), pd(tr))
    near("Del2Tracers", compute_errors(M, S.aux["Del2TracersCell"], sc(M, del2Tracer, "Cell", ntr=NT), "Cell"),
         (0.0081206665417422382, 0.004917863312407276))
