"""GPU parity: every HIP kernel path against the CPU oracle on identical mesh + state.

Tolerance: BASELINE.json asks for tendencies within 1e-12 relative of the CPU reference.
The kernels are built with -ffp-contract=off and keep the reference's operation order, so
on owned elements the results are expected to be BIT-IDENTICAL to the oracle; the tests
assert exact equality and report the relative difference if that ever fails
(RTOL = 1e-12 is the contractual bound, asserted as well).
"""
import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import planar_hex, spherical_voronoi, icosahedral_points, pad_max_edges
from oracle import oracle as O
from tests.problem import Problem, max_rel_diff, poison_tendencies

pytestmark = pytest.mark.gpu
RTOL = 1e-12

AUX_2D = ("KineticEnergyCell", "VelocityDivCell", "FluxLayerThickEdge", "MeanLayerThickEdge", "SshCell",
          "RelVortVertex", "NormRelVortVertex", "NormPlanetVortVertex", "NormRelVortEdge", "NormPlanetVortEdge",
          "Del2Edge", "Del2DivCell", "Del2RelVortVertex")
OWNED = {"C": "NCellsOwned", "E": "NEdgesOwned", "V": "NVerticesOwned"}


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert oa.device_count() > 0, "no HIP device: GPU tests need a real MI355X"
    oa.device_init(0)


def check(name, got, ref, n_owned):
    g, r = got[..., :n_owned, :], ref[..., :n_owned, :]
    rel = max_rel_diff(g, r)
    assert rel <= RTOL, f"{name}: max rel diff {rel:.3e} > {RTOL}"
    assert np.array_equal(g, r), f"{name}: within {RTOL} (rel {rel:.3e}) but not bit-identical"


CASES = [
    # (nx, ny, dc, K, NT, config overrides)
    (16, 16, 1.0, 4, 1, {}),                                   # BASELINE configs[0]
    (16, 16, 30e3, 5, 2, {}),                                  # odd K -> scalar (W=1) path
    (24, 20, 30e3, 60, 2, {}),                                 # K=60 (30 level-pairs per column)
    (20, 24, 30e3, 80, 6, {}),                                 # K=80, NT=6 as QU30
    (16, 16, 30e3, 8, 3, {"FluxThicknessUpwind": 1, "FluxTracerUpwind": 1}),
    (16, 16, 30e3, 8, 2, {"WindForcingTendencyEnable": 1, "BottomDragTendencyEnable": 1, "BottomDragCoeff": 1.0e-3,
                          "WindInterpIsotropic": 0}),
    (16, 16, 30e3, 6, 2, {"VelHyperDiffTendencyEnable": 0, "TracerHyperDiffTendencyEnable": 0, "EddyDiff4": 3.0}),
    (16, 16, 30e3, 6, 2, {"EddyDiff4": 2.5e9, "PVTendencyEnable": 0, "KETendencyEnable": 0}),
    (16, 16, 30e3, 130, 1, {}),                                # KV = 65 > 64: level loop inside the thread
    # spherical Voronoi meshes (stand-ins for BASELINE configs[1]: QU240 is a download):
    ("ico3", 0, 0, 12, 2, {}),                                 # icosahedral, 642 cells: 12 pentagons, MaxEdges 6
    ("fib1500", 0, 0, 10, 2, {}),                              # 1500 cells with pentagons AND heptagons, MaxEdges 7
    ("ico4", 0, 0, 60, 2, {}),                                 # 2562 cells, 60 levels, T+S
    ("hex24pad8", 0, 0, 6, 2, {}),                             # planar hexagons stored with maxEdges = 8: tables compacted to 6
    ("ico3pad8", 0, 0, 6, 1, {}),                              # same mesh stored with maxEdges = 8: pentagons
                                                               # fall outside the ring kernels' valences
    # the DEFAULT wind interpolation (isotropic, kite-weighted: InterpCellToEdge, HorzOperators.h:161-180)
    (16, 16, 30e3, 8, 2, {"WindForcingTendencyEnable": 1, "BottomDragTendencyEnable": 1, "BottomDragCoeff": 1.0e-3,
                          "WindInterpIsotropic": 1}),
    ("ico3", 0, 0, 6, 1, {"WindForcingTendencyEnable": 1, "WindInterpIsotropic": 1}),   # same, on the sphere
    (20, 24, 30e3, 80, 37, {}),                                # BASELINE configs[4]: 80 levels, 37 BGC tracers
    ("ico3", 0, 0, 80, 37, {}),                                # 37 tracers with the pentagon ring launches
    # tracer counts around the switches of the tracer loops (LDS tile patches from 4 tracers on; three tracers per trip in
    # level 1 where 3 divides the count and the tables are 6 wide; other thread geometry above 8 tracers)
    (20, 24, 30e3, 80, 4, {}),
    ("ico4", 0, 0, 20, 5, {}),
    (24, 20, 30e3, 12, 9, {}),
    ("fib1500", 0, 0, 16, 12, {}),
    ("ico4", 0, 0, 80, 3, {}),
    # meshes with cells whose per-cell lists are not in ring order (MeshView::BadCells: generic bodies over a list)
    ("hex32x24_perm3", 0, 0, 8, 2, {}),
    ("hex24x20_perm10", 0, 0, 80, 6, {}),
    ("ico4_perm2", 0, 0, 12, 2, {}),
    ("fib1500_perm5", 0, 0, 6, 1, {}),
    ("hex24x20_coast_mixed_perm5", 0, 0, 6, 2, {"FluxThicknessUpwind": 1}),
    # land boundaries (culled meshes, as every real ocean mesh is: omega_amd/meshgen.py cull / coast_mask)
    ("hex32x24_coast_mixed", 0, 0, 8, 2, {}),                  # island + single-cell island + ragged patch
    ("hex24x20_coast_lakes", 0, 0, 6, 2, {}),                  # one- and two-cell lakes, one-cell-wide bays
    ("hex24x20_coast_strait", 0, 0, 80, 6, {}),                # walls with a one-cell gap, a one-cell-wide channel
    ("hex24x20_coast_ragged_raw", 0, 0, 5, 1, {}),             # random land; boundary edges keep [missing, cell]; odd K
    ("hex20x16_coast_channel_compact", 0, 0, 60, 2, {"FluxThicknessUpwind": 1, "FluxTracerUpwind": 1}),
    ("ico4_coast_mixed", 0, 0, 12, 2, {}),                     # sphere with pentagons on and off the coast
    ("fib1500_coast_ragged", 0, 0, 10, 2, {}),                 # 7-wide tables, hexagons dominant, random land
    ("ico3_pad8_coast_island_raw_compact", 0, 0, 6, 1, {}),    # stored with maxEdges = 8
    ("hex24x20_coast_mixed", 0, 0, 8, 2, {"WindForcingTendencyEnable": 1, "BottomDragTendencyEnable": 1,
                                          "BottomDragCoeff": 1.0e-3, "WindInterpIsotropic": 1}),
]

def sphere(name):
    """Named test meshes (tests/meshes.py), generated once per session."""
    from tests.meshes import named_mesh
    alias = {"ico3pad8": "ico3_pad8", "hex24pad8": "hex24x20_pad8"}
    return named_mesh(alias.get(name, name))


def _mk(case):
    nx, ny, dc, K, NT, cfg = case
    P = Problem(sphere(nx) if isinstance(nx, str) else planar_hex(nx, ny, dc), K, NT, config=cfg)
    if cfg.get("WindForcingTendencyEnable"):
        rng = np.random.default_rng(7)
        zs = np.zeros(P.mesh.NCellsSize)
        ms = np.zeros(P.mesh.NCellsSize)
        zs[:-1] = rng.uniform(-0.1, 0.1, P.mesh.NCellsAll)
        ms[:-1] = rng.uniform(-0.1, 0.1, P.mesh.NCellsAll)
        P.aux.set("ZonalStressCell", zs)
        P.aux.set("MeridStressCell", ms)
        P.oracle.aux["ZonalStressCell"][:] = zs
        P.oracle.aux["MeridStressCell"][:] = ms
    return P


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}_K{c[3]}_NT{c[4]}_{i}" for i, c in enumerate(CASES)])
def test_aux_state_compute_all(case):
    """AuxiliaryState::computeAll, array by array (reference-structured kernels)."""
    P = _mk(case)
    # (r6, after test/ocn/TendenciesTest.cpp:159-163) every computed array starts as NaN: an owned element that no kernel
    # writes cannot pass for one whose value happens to be what the allocation held
    # (the zero sentinel row, which no kernel writes and missing neighbours read, stays zero)
    for name in (*AUX_2D, "HTracersEdge", "Del2TracersCell"):
        if name in AUX_2D or case[4] > 0:
            poison = np.full(P.aux._shape(name), np.nan)
            poison[..., -1, :] = 0.0
            P.aux.set(name, poison)
    P.aux.compute_all(P.state, P.tracers)
    oa.device_synchronize()
    P.oracle.compute_all_aux(P.h, P.u, P.tr)
    m = P.mesh
    for name in AUX_2D:
        check(name, P.aux.get(name), P.oracle.aux[name], getattr(m, OWNED[oa.AUX_SHAPES[name]]))
    check("HTracersEdge", P.aux.get("HTracersEdge"), P.oracle.aux["HTracersEdge"], m.NEdgesOwned)
    check("Del2TracersCell", P.aux.get("Del2TracersCell"), P.oracle.aux["Del2TracersCell"], m.NCellsOwned)
    if case[5].get("WindForcingTendencyEnable"):
        got, ref = P.aux.get("NormalStressEdge")[: m.NEdgesOwned], P.oracle.aux["NormalStressEdge"][: m.NEdgesOwned]
        # cos/sin come from different libms (device vs glibc): 4 ulp
        assert np.allclose(got, ref, rtol=1e-15, atol=1e-17)


@pytest.mark.parametrize("fused", [False, True], ids=["unfused", "fused"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}_K{c[3]}_NT{c[4]}_{i}" for i, c in enumerate(CASES)])
def test_compute_all_tendencies(case, fused):
    """Tendencies::computeAllTendencies: reference launch structure and fused RHS."""
    P = _mk(case)
    wind = case[5].get("WindForcingTendencyEnable")
    P.tend.set_fused(fused)
    poison_tendencies(P)          # (r6) TendenciesTest.cpp:159-163: every tendency variable is NaN before the evaluation
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    m = P.mesh
    if wind:
        # NormalStressEdge differs by libm ulps -> compare the wind-touched level loosely
        got = P.tend.get(1)[: m.NEdgesOwned]
        assert np.array_equal(got[:, 1:], uT[: m.NEdgesOwned, 1:])
        assert np.allclose(got[:, 0], uT[: m.NEdgesOwned, 0], rtol=1e-13, atol=0)
    else:
        check("NormalVelocityTend", P.tend.get(1), uT, m.NEdgesOwned)
    check("LayerThicknessTend", P.tend.get(0), hT, m.NCellsOwned)
    check("TracerTend", P.tend.get(2)[: case[4]], trT[: case[4]], m.NCellsOwned)


@pytest.mark.parametrize("case", [(20, 24, 30e3, 80, 6, {}), ("ico4", 0, 0, 80, 6, {}), ("fib1500", 0, 0, 20, 5, {}),
                                  ("hex24x20_coast_strait", 0, 0, 80, 6, {})], ids=lambda c: f"{c[0]}_K{c[3]}_NT{c[4]}")
def test_both_forms_of_the_level3_tracer_loop(case):
    """Option TracerPatch (Tuning.h): the level-3 kernel's tracer loop with the neighbour values through LDS tile
    patches (default from 4 tracers on) and with per-thread gathers give the same bits -- each other's and the oracle's.
    (The rest of the suite runs with the default; this is the only place the gather form of these cases is compared.)"""
    P = _mk(case)
    NT, m = case[4], P.mesh
    _, _, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    got = {}
    P.tend.set_fused(True)
    try:
        for on in (1, 0):
            oa.set_option("TracerPatch", on)
            P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
            oa.device_synchronize()
            got[on] = P.tend.get(2)[:NT].copy()
            check(f"TracerTend (TracerPatch={on})", got[on], trT[:NT], m.NCellsOwned)
    finally:
        oa.set_option("TracerPatch", 1)
    assert np.array_equal(got[0][:, : m.NCellsOwned], got[1][:, : m.NCellsOwned])


def test_sphere_meshes_take_the_fast_paths():
    """Which kernel paths the spherical meshes exercise (HorzMesh table diagnostics): ring-form
    del2 everywhere; cell-centric PV with the pentagons' edges on the edge-centric list."""
    P = _mk(("ico3", 0, 0, 4, 1, {}))
    assert P.mesh.get_int("Del2RingOK") == 1 and P.mesh.get_int("Del2VertOK") == 1
    assert P.mesh.get_int("CellPVOK") == 1 and P.mesh.get_int("CellPVFinalOK") == 1
    assert P.mesh.get_int("NIrregularEdges") == 0          # pentagons are handled by the 5-ring instantiation
    P = _mk(("fib1500", 0, 0, 4, 1, {}))
    assert P.mesh.get_int("MaxEdges") == 7 and P.mesh.get_int("Del2RingOK") == 1
    assert P.mesh.get_int("DomM1") == 1        # hexagons dominate a 7-wide mesh: the full sweeps take valence 6
    assert P.mesh.get_int("CellPVOK") == 1 and P.mesh.get_int("NIrregularEdges") == 0
    # ... on a second, 6-wide set of cell tables, with the heptagons on list launches of the 7-slot kernels
    assert 0 < P.mesh.get_int("NWideCells") < 0.1 * P.mesh.NCellsAll
    assert P.mesh.get_int("NarrowTables") == (1 if oa.get_option("NarrowTables") else 0)
    P = _mk(("ico3pad8", 0, 0, 4, 1, {}))
    assert P.mesh.get_int("MaxEdgesFile") == 8 and P.mesh.get_int("CellPVOK") == 1
    if oa.get_option("KeepMaxEdges") == 1:   # (child run of tests/test_00_multirank_gpu.py)
        assert P.mesh.get_int("MaxEdges") == 8
        assert P.mesh.get_int("NIrregularEdges") == 12 * 5      # 5 < MaxEdges - 2: edge-centric list
    else:       # the mesh keeps its tables at the largest valence present, whatever the file's maxEdges
        assert P.mesh.get_int("MaxEdges") == 6 and P.mesh.get_int("NIrregularEdges") == 0


def test_generic_flags_follow_the_environment():
    """With the option ForceGeneric = 1 (set by tests/test_00_multirank_gpu.py for a child run) every ring-table
    flag is off, so the parity tests of that run exercise the generic kernels; otherwise they are on."""
    P = _mk((16, 16, 30e3, 4, 1, {}))
    want = 0 if oa.get_option("ForceGeneric") == 1 else 1
    for flag in ("PVChainOK", "CellPVOK", "CellPVFinalOK", "Del2RingOK", "Del2VertOK"):
        assert P.mesh.get_int(flag) == want, flag


def test_group_tendencies_fb_path():
    """computeThicknessTendencies / computeTracerTendencies / computeVelocityTendencies
    (the ForwardBackward stepper's calls, Tendencies.cpp:488-575)."""
    P = _mk((20, 16, 30e3, 10, 2, {}))
    m = P.mesh
    P.tend.compute_thickness_tendencies(P.state, P.aux)
    P.tend.compute_tracer_tendencies(P.state, P.aux, P.tracers)
    P.tend.compute_velocity_tendencies(P.state, P.aux)
    oa.device_synchronize()
    check("hTend", P.tend.get(0), P.oracle.compute_thickness_tendencies(P.h, P.u), m.NCellsOwned)
    check("trTend", P.tend.get(2), P.oracle.compute_tracer_tendencies(P.h, P.u, P.tr), m.NCellsOwned)
    check("uTend", P.tend.get(1), P.oracle.compute_velocity_tendencies(P.h, P.u), m.NEdgesOwned)


@pytest.mark.parametrize("kind,okind,fuse", [("RungeKutta4", "rk4", True), ("RungeKutta4", "rk4", False),
                                             ("RungeKutta2", "rk2", None), ("Forward-Backward", "fb", None)])
def test_time_steppers(kind, okind, fuse):
    """doStep x3 against the oracle's restated steppers (state + tracers, bit-exact); RK4 with the
    stage updates folded into the RHS kernels (default) and with the separate update kernels."""
    P = _mk((16, 16, 30e3, 6, 2, {}))
    dt = 600.0
    st = oa.TimeStepper(kind, dt, P.tend, P.aux, P.mesh, None, P.tracers)
    if fuse is not None:
        st.set_option("FuseStageUpdates", fuse)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    m = P.mesh
    # (r6) the NEW time level and the tendencies start as NaN (sentinel rows zero): a scheme must write every element of
    # what it hands back, whatever the buffers held
    nan_h, nan_u, nan_tr = np.full_like(P.h, np.nan), np.full_like(P.u, np.nan), np.full_like(P.tr, np.nan)
    nan_h[-1] = nan_u[-1] = 0.0
    nan_tr[:, -1] = 0.0
    P.state.copy_to_device(nan_h, nan_u, 1)
    P.tracers.copy_to_device(nan_tr, 1)
    poison_tendencies(P)
    for step in range(3):
        st.do_step(P.state)
        oa.device_synchronize()
        P.oracle.step(okind, ost, dt)
        h, u = P.state.copy_to_host(0)
        tr = P.tracers.copy_to_host(0)
        check(f"h step {step}", h, ost["h"][0], m.NCellsOwned)
        check(f"u step {step}", u, ost["u"][0], m.NEdgesOwned)
        check(f"tr step {step}", tr, ost["tr"][0], m.NCellsOwned)


@pytest.mark.parametrize("name", ["hex32x24_perm3", "ico4_perm2", "fib1500_perm5"])
def test_cells_out_of_ring_order_keep_the_fast_paths(name):
    """A mesh file in which a few cells' per-cell lists are not in ring order (two slots swapped): the reference does not
    care, so such cells must not cost the mesh its fast kernels (the flags used to be per mesh).  They are
    served per cell -- generic level-1 / level-2 bodies over the list BadCells, their edges on the irregular-edge list --
    while every ring-table flag stays on; fused RHS, reference-structured RHS and two stage-fused RK4 steps are bit-exact."""
    g = sphere(name)
    P = _mk((name, 0, 0, 8, 2, {}))
    m = P.mesh
    for f in ("CellL1OK", "CellPVOK", "CellPVFinalOK", "Del2RingOK", "Del2VertOK"):
        assert m.get_int(f) == 1, f
    nbad = len(g["permutedCells"])
    assert nbad > 0 and m.get_int("NBadCells") == nbad
    assert m.get_int("NIrregularEdges") >= 3 * nbad // 2       # every edge of a bad cell runs through the edge-centric list
    for fused in (True, False):
        P.tend.set_fused(fused)
        P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
        oa.device_synchronize()
        hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
        check("hTend", P.tend.get(0), hT, m.NCellsOwned)
        check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
        check("trTend", P.tend.get(2), trT, m.NCellsOwned)
    P.tend.set_fused(True)
    dt = 600.0 if not name.startswith("fib") else 5.0
    st = oa.TimeStepper("RungeKutta4", dt, P.tend, P.aux, P.mesh, None, P.tracers)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    for _ in range(2):
        st.do_step(P.state)
        P.oracle.step("rk4", ost, dt)
    oa.device_synchronize()
    h, u = P.state.copy_to_host(0)
    check("h", h, ost["h"][0], m.NCellsOwned)
    check("u", u, ost["u"][0], m.NEdgesOwned)
    check("tr", P.tracers.copy_to_host(0), ost["tr"][0], m.NCellsOwned)


@pytest.mark.parametrize("name", ["ico3", "fib1500", "ico3pad8"])
def test_rk4_on_the_sphere(name):
    """Two RK4 steps (stage updates fused into the RHS kernels) on the spherical meshes: pentagon /
    heptagon ring launches, and with maxEdges = 8 the pentagons' edges through the edge-centric list."""
    P = _mk((name, 0, 0, 8, 2, {}))
    dt = 600.0 if name != "fib1500" else 5.0   # the Fibonacci mesh has a few very short edges
    st = oa.TimeStepper("RungeKutta4", dt, P.tend, P.aux, P.mesh, None, P.tracers)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    m = P.mesh
    for step in range(2):
        st.do_step(P.state)
        oa.device_synchronize()
        P.oracle.step("rk4", ost, dt)
    h, u = P.state.copy_to_host(0)
    tr = P.tracers.copy_to_host(0)
    assert np.isfinite(ost["h"][0]).all() and np.isfinite(ost["u"][0]).all()
    check("h", h, ost["h"][0], m.NCellsOwned)
    check("u", u, ost["u"][0], m.NEdgesOwned)
    check("tr", tr, ost["tr"][0], m.NCellsOwned)


def test_known_answer_through_gpu():
    """The reference's own planar-mesh answers through the HIP path: AuxiliaryVarsTest
    KineticEnergy / VelocityDiv norms (AuxiliaryVarsTest.cpp:34-37) from the GPU arrays."""
    from tests import ka_common as ka
    from tests.test_oracle_known_answers import vecX, vecY, divergence
    K = 16
    P = Problem(planar_hex(48, 48, 1.0 / 48.0), K, 1)
    M = P.omesh
    n = M.NEdgesOwned
    u = np.zeros((M.NEdgesSize, K))
    u[:n] = (np.cos(M.AngleEdge[:n]) * vecX(M.XEdge[:n], M.YEdge[:n])
             + np.sin(M.AngleEdge[:n]) * vecY(M.XEdge[:n], M.YEdge[:n]))[:, None]
    P.state.copy_to_device(P.h, u, 0)
    P.aux.compute_mom_aux(P.state)
    oa.device_synchronize()
    ke = ka.set_scalar(M, K, lambda X, Y: (vecX(X, Y) ** 2 + vecY(X, Y) ** 2) / 2, "Cell")
    ka.check_errors("KineticEnergy", ka.compute_errors(M, P.aux.get("KineticEnergyCell"), ke, "Cell"),
                    (0.00994439065100057897, 0.00703403756741667954), 2e-4)
    ka.check_errors("VelocityDiv", ka.compute_errors(M, P.aux.get("VelocityDivCell"),
                                                     ka.set_scalar(M, K, divergence, "Cell"), "Cell"),
                    (0.00124886886594453264, 0.00124886886590973452), 2e-4)


def _ms_problem(nx, K=1, NT=1, fused=True):
    from tests import manufactured as ms
    g = ms.mesh(nx)
    wx, wy = ms.wavelengths(g)
    P = Problem(g, K, NT)
    h, u, tr = ms.initial_state(P.omesh, K, NT, wx, wy)
    P.h, P.u, P.tr = h, u, tr
    P.state.copy_to_device(h, u, 0)
    P.tracers.copy_to_device(tr, 0)
    P.tend.set_fused(fused)
    P.tend.use_manufactured_solution(P.mesh, wx, wy, ms.ETA0)
    return P, wx, wy


@pytest.mark.parametrize("fused", [False, True], ids=["unfused", "fused"])
def test_manufactured_tendencies_match_the_oracle(fused):
    """Custom (manufactured-solution) tendencies through the Tendencies hooks, both RHS structures.
    sin / cos come from the device libm here and glibc in the oracle: ulp-level differences in the
    source term, so this path is compared at the contractual 1e-12, not bit for bit."""
    from tests import manufactured as ms
    P, wx, wy = _ms_problem(24, K=3, NT=1, fused=fused)
    P.oracle.use_manufactured_solution(wx, wy, ms.ETA0)
    try:
        t = 4321.0
        P.tend.set_time(t)
        P.oracle.set_time(t)
        P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
        oa.device_synchronize()
        hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    finally:
        P.oracle.use_manufactured_solution()
    m = P.mesh
    assert max_rel_diff(P.tend.get(0)[: m.NCellsOwned], hT[: m.NCellsOwned], scale=np.abs(hT).max()) <= RTOL
    assert max_rel_diff(P.tend.get(1)[: m.NEdgesOwned], uT[: m.NEdgesOwned], scale=np.abs(uT).max()) <= RTOL
    # and the source term is really there
    P.tend.clear_custom_tendencies()
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    assert np.abs(P.tend.get(0)[: m.NCellsOwned] - hT[: m.NCellsOwned]).max() > 1e-6


def test_custom_tendency_hooks_see_a_materialised_auxiliary_state():
    """The fused RHS keeps its intermediates private; a custom tendency hook receives the AuxiliaryState and may read it
    (reference: computeAllTendencies fills it before the hooks run, Tendencies.cpp:591, 288-291, 416-419): when a hook is
    installed the library runs AuxiliaryState::computeAll first, so the reference's arrays hold the current state."""
    P = _mk((16, 16, 30e3, 6, 1, {}))
    seen = {}

    def hook(tend, h, u, nall, nsize, k, pitch, t, stream):
        oa.device_synchronize()
        seen["KE"] = P.aux.get("KineticEnergyCell").copy()
        seen["Del2"] = P.aux.get("Del2Edge").copy()

    P.tend.set_custom_tendency(0, hook)
    try:
        P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
        oa.device_synchronize()
    finally:
        P.tend.set_custom_tendency(0, None)
    P.oracle.compute_all_aux(P.h, P.u, P.tr)
    check("KineticEnergyCell in the hook", seen["KE"], P.oracle.aux["KineticEnergyCell"], P.mesh.NCellsOwned)
    check("Del2Edge in the hook", seen["Del2"], P.oracle.aux["Del2Edge"], P.mesh.NEdgesOwned)


def test_manufactured_solution_converges_through_the_gpu_path():
    """RK4 on the GPU with the custom tendencies (stage times from the stepper): second-order
    convergence to the exact solution, and the same numbers as the oracle run to 1e-12."""
    from tests import manufactured as ms
    from tests.test_manufactured_solution import run_oracle
    errs = {}
    for nx in (16, 32):
        P, wx, wy = _ms_problem(nx)
        dt = 600.0 * 16 / nx
        nsteps = int(round(3.0 * 3600 / dt))
        st = oa.TimeStepper("RungeKutta4", dt, P.tend, P.aux, P.mesh, None, P.tracers)
        for _ in range(nsteps):
            st.do_step(P.state)
        oa.device_synchronize()
        assert abs(st.time - nsteps * dt) < 1e-9
        h, _ = P.state.copy_to_host(0)
        errs[nx] = ms.l2_error_h(P.omesh, h, nsteps * dt, wx, wy)
        e_orc, ost = run_oracle(nx, hours=3.0)
        assert abs(errs[nx] - e_orc) <= 1e-9 * e_orc
        assert max_rel_diff(h[: P.mesh.NCellsOwned], ost["h"][0][: P.mesh.NCellsOwned]) <= 1e-11
    rate = np.log2(errs[16] / errs[32])
    assert 1.8 < rate < 2.3, (errs, rate)


def _random_cases(n, seed=1234, tracer_counts=(0, 1, 2, 5)):
    rng = np.random.default_rng(seed)
    flags = ["ThicknessFluxTendencyEnable", "PVTendencyEnable", "KETendencyEnable", "SSHTendencyEnable",
             "VelDiffTendencyEnable", "VelHyperDiffTendencyEnable", "TracerHorzAdvTendencyEnable",
             "TracerDiffTendencyEnable", "TracerHyperDiffTendencyEnable", "FluxThicknessUpwind", "FluxTracerUpwind",
             "BottomDragTendencyEnable"]
    out = []
    for i in range(n):
        K = int(rng.choice([1, 2, 3, 7, 16, 17, 32, 33, 48, 64, 96, 100, 128]))
        NT = int(rng.choice(list(tracer_counts)))
        mesh = rng.choice(["hex", "hex", "ico2", "fib300"])
        cfg = {}
        if i % 3:                       # two thirds of the cases with random switches, one third Default.yml
            for f in flags:
                if rng.random() < 0.35:
                    cfg[f] = int(not O.default_config().__getattribute__(f))
            if rng.random() < 0.5:
                cfg["EddyDiff4"] = float(rng.choice([0.0, 1.0e9]))
        out.append((str(mesh), int(rng.integers(10, 20)), int(rng.integers(5, 9)) * 2, K, NT, cfg))
    return out


# (the second set: tracer counts on both sides of the tracer loops' switches -- tile patches from 4 tracers on, three
# tracers per trip where 3 divides the count, another thread geometry above 8)
@pytest.mark.parametrize("case", _random_cases(64) + _random_cases(32, seed=4321, tracer_counts=(3, 4, 6, 7, 9, 12)),
                         ids=lambda c: f"{c[0]}_K{c[3]}_NT{c[4]}_{len(c[5])}opts")
def test_randomised_configurations(case):
    """Random level counts (aligned / unaligned / odd / single), tracer counts (incl. none), meshes and
    option sets: fused RHS and one RK4 step against the oracle, bit for bit."""
    kind, nx, ny, K, NT, cfg = case
    g = planar_hex(nx, ny, 30e3) if kind == "hex" else sphere(kind)
    P = Problem(g, K, NT, config=cfg)
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    m = P.mesh
    check("hTend", P.tend.get(0), hT, m.NCellsOwned)
    check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
    if NT > 0:
        check("trTend", P.tend.get(2)[:NT], trT[:NT], m.NCellsOwned)
    if kind != "fib300":               # (a few very short edges: not a stable time step at dt = 600 s)
        st = oa.TimeStepper("RungeKutta4", 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
        ost = P.oracle.make_state(P.h, P.u, P.tr)
        st.do_step(P.state)
        oa.device_synchronize()
        P.oracle.step("rk4", ost, 600.0)
        h, u = P.state.copy_to_host(0)
        check("h", h, ost["h"][0], m.NCellsOwned)
        check("u", u, ost["u"][0], m.NEdgesOwned)
        if NT > 0:
            check("tr", P.tracers.copy_to_host(0)[:NT], ost["tr"][0][:NT], m.NCellsOwned)


def test_bench_shaped_problem_against_the_oracle():
    """The bench's own setup at a size the oracle still finishes in a second: Morton-ordered planar mesh,
    80 levels, 6 tracers (tiles of 32 elements, 5 level chunks per tile), fused RHS + RK4 step."""
    from omega_amd.meshgen import reorder_cells_morton
    g = reorder_cells_morton(planar_hex(160, 128, 30.0e3))
    P = Problem(g, 80, 6)
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    m = P.mesh
    check("hTend", P.tend.get(0), hT, m.NCellsOwned)
    check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
    check("trTend", P.tend.get(2), trT, m.NCellsOwned)
    st = oa.TimeStepper("RungeKutta4", 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    st.do_step(P.state)
    oa.device_synchronize()
    P.oracle.step("rk4", ost, 600.0)
    h, u = P.state.copy_to_host(0)
    check("h", h, ost["h"][0], m.NCellsOwned)
    check("u", u, ost["u"][0], m.NEdgesOwned)
    check("tr", P.tracers.copy_to_host(0), ost["tr"][0], m.NCellsOwned)


def test_hip_graph_replay_matches_the_oracle():
    """On a non-default stream the fused RHS and the one-rank stage-fused RK4 step are captured into HIP graphs the
    second time they run with the same arrays and replayed afterwards (GraphCache.h).  Replays must see new DATA in
    the same arrays, both time-level parities of the stepper need their own graph, and everything stays bit-equal
    to the oracle."""
    P = _mk((20, 24, 30e3, 12, 3, {}))
    m, st = P.mesh, oa.Stream()
    P.tend.set_graphs(True)
    for rep in range(4):
        if rep == 3:    # same arrays, new contents
            P.h[:-1] *= 1.25
            P.tr[:, :-1] += 0.5
            P.state.copy_to_device(P.h, P.u, 0)
            P.tracers.copy_to_device(P.tr, 0)
        P.tend.compute_all_tendencies(P.state, P.aux, P.tracers, stream=st)
    st.synchronize()
    gs = P.tend.graph_stats()
    assert gs["captures"] == 1 and gs["replays"] == 3, gs
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    check("hTend", P.tend.get(0), hT, m.NCellsOwned)
    check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
    check("trTend", P.tend.get(2), trT, m.NCellsOwned)
    stepper = oa.TimeStepper("RungeKutta4", 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
    stepper.set_option("UseGraphs", True)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    for step in range(8):
        stepper.do_step(P.state, stream=st)
        P.oracle.step("rk4", ost, 600.0)
    st.synchronize()
    gs = stepper.graph_stats()
    assert gs["captures"] == 2 and gs["replays"] >= 4, gs      # one graph per time-level parity, then replays
    h, u = P.state.copy_to_host(0)
    check("h", h, ost["h"][0], m.NCellsOwned)
    check("u", u, ost["u"][0], m.NEdgesOwned)
    check("tr", P.tracers.copy_to_host(0), ost["tr"][0], m.NCellsOwned)
    # switching replay off mid-run changes nothing
    stepper.set_option("UseGraphs", False)
    stepper.do_step(P.state, stream=st)
    P.oracle.step("rk4", ost, 600.0)
    st.synchronize()
    h, u = P.state.copy_to_host(0)
    check("h", h, ost["h"][0], m.NCellsOwned)
    check("u", u, ost["u"][0], m.NEdgesOwned)


@pytest.mark.parametrize("order", ["curve", "hilbert", "kd"])
@pytest.mark.parametrize("mesh", ["hex", "ico3"])
def test_curve_ordered_local_numbering(mesh, order):
    """Decomp with LocalOrder::Curve / Hilbert / KdTree on a row-major (unordered) input mesh: another local numbering, the same
    results per global id -- fused RHS and an RK4 step against the oracle running on the product's own local arrays."""
    g = planar_hex(40, 24, 30e3) if mesh == "hex" else sphere("ico3")
    P = Problem(g, 12, 2, local_order=order)
    assert not np.array_equal(P.cell_id[:-1], np.arange(1, g["nCells"] + 1))
    m = P.mesh
    assert m.get_int("CellL1OK") == 1 and m.get_int("CellPVFinalOK") == 1
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    check("hTend", P.tend.get(0), hT, m.NCellsOwned)
    check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
    check("trTend", P.tend.get(2), trT, m.NCellsOwned)
    # and the same numbers as the reference numbering, element by element through the global ids
    Q = Problem(g, 12, 2)
    Q.tend.compute_all_tendencies(Q.state, Q.aux, Q.tracers)
    oa.device_synchronize()
    a, b = np.zeros((g["nCells"], 12)), np.zeros((g["nCells"], 12))
    a[P.cell_id[:-1] - 1], b[Q.cell_id[:-1] - 1] = P.tend.get(0)[:-1], Q.tend.get(0)[:-1]
    assert np.array_equal(a, b)
    st = oa.TimeStepper("RungeKutta4", 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    st.do_step(P.state)
    oa.device_synchronize()
    P.oracle.step("rk4", ost, 600.0)
    h, u = P.state.copy_to_host(0)
    check("h", h, ost["h"][0], m.NCellsOwned)
    check("u", u, ost["u"][0], m.NEdgesOwned)


@pytest.mark.parametrize("name,K,NT", [("ico5", 60, 2), ("hex484", 60, 2), ("hex680", 80, 6)],
                         ids=["configs1_QU240_sphere", "configs2_EC30to60_size", "configs3_QU30_size"])
def test_baseline_configurations_at_full_size_against_the_oracle(name, K, NT):
    """BASELINE.json configs[1..3] at their FULL sizes, element by element against the oracle (the property tests of
    tests/test_gpu_properties.py are what remains size-independent; this is the direct comparison): a spherical
    quasi-uniform mesh of 10 242 cells x 60 levels (QU240 itself is a download), 234 256 cells x 60
    levels x 2 tracers, 462 400 cells x 80 levels x 6 tracers -- EXACTLY as bench.py runs them: row-major input, local
    numbering by Decomp in k-d order (bench.py's default --local-order kd; hex680 x 80 x 6 is the headline workload
    `qu30`).  The fused RHS, then one stage-fused RK4 step (the SYPD path: at headline size the stage pair
    CellPVFinalBody + FusedCell3Body is the largest kernel of the step) against the oracle's step."""
    import gc
    if name == "ico5":
        g = sphere(name)
    else:
        n = int(name[3:])
        g = planar_hex(n, n, 30.0e3 if n == 680 else 45.0e3)
    P = Problem(g, K, NT, local_order="kd")
    m = P.mesh
    for f in ("CellL1OK", "CellPVOK", "CellPVFinalOK", "Del2RingOK", "Del2VertOK"):
        assert m.get_int(f) == 1, f
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    check("hTend", P.tend.get(0), hT, m.NCellsOwned)
    check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
    check("trTend", P.tend.get(2), trT, m.NCellsOwned)
    dt = 600.0
    st = oa.TimeStepper("RungeKutta4", dt, P.tend, P.aux, P.mesh, None, P.tracers)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    st.do_step(P.state)
    oa.device_synchronize()
    P.oracle.step("rk4", ost, dt)
    h, u = P.state.copy_to_host(0)
    assert np.isfinite(ost["h"][0]).all()
    check("h", h, ost["h"][0], m.NCellsOwned)
    check("u", u, ost["u"][0], m.NEdgesOwned)
    check("tr", P.tracers.copy_to_host(0), ost["tr"][0], m.NCellsOwned)
    del P
    gc.collect()


@pytest.mark.parametrize("workload", ["ico7", "fib7_coast", "ico8"])
def test_spherical_bench_workloads_at_bench_size_against_the_oracle(workload):
    """The spheres README / DESIGN quote roofline fractions for, at the size and in the order bench.py runs them: `ico7`
    (163 842 cells, 12 pentagons) and `fib7_coast` (relaxed Fibonacci sphere, valences 5 / 6 / 7, 28 % land removed:
    117 746 cells), 80 levels, 6 tracers, local numbering k-d -- the mesh built by bench.py's own workload_mesh().  At
    this size the kernels take paths the small spheres barely reach (k-d tiles of a curved surface, tile patches that do
    not fit their LDS rows and fall back to per-thread gathers, wide-cell lists of thousands of heptagons, tail-split
    tiles): the fused RHS and one stage-fused RK4 step, element by element against the oracle.  `ico8` (655 362 cells: QU30-sized
    ON THE SPHERE, the largest sphere a fraction is quoted for) takes the fused RHS only."""
    import gc
    import bench
    bench.load_library()
    K, NT = bench.WORKLOADS[workload][3:5]
    g = bench.workload_mesh(workload)
    P = Problem(g, K, NT, local_order="kd")
    m = P.mesh
    for f in ("CellL1OK", "CellPVOK", "CellPVFinalOK", "Del2RingOK", "Del2VertOK"):
        assert m.get_int(f) == 1, f
    assert m.get_int("NBadCells") == 0
    # the level-3 tracer loop's tile patches: how many 16-cell tiles need more rows than the LDS patch holds (those tiles
    # take the per-thread gathers inside the same launch).  A curved k-d tile touches ~ 35 rows of the 48: the fallback must
    # stay the exception, or the quoted fractions are those of the fallback path
    tiles, fallback = m.get_int("NPatchTiles16"), m.get_int("NPatchFallback16")
    assert tiles == (m.NCellsAll + 15) // 16 and fallback <= 0.05 * tiles, (tiles, fallback)
    print(f"[{workload}] 16-cell tiles: {tiles}, of which patch fallback: {fallback}")
    if workload == "fib7_coast":
        assert m.get_int("MaxEdges") == 7 and m.get_int("NWideCells") > 500 and m.get_int("NIrregularEdges") > 1000
        assert int(np.asarray(g["boundaryEdge"]).sum()) > 1000
    else:
        assert m.get_int("MaxEdges") == 6
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    check("hTend", P.tend.get(0), hT, m.NCellsOwned)
    check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
    check("trTend", P.tend.get(2), trT, m.NCellsOwned)
    if workload == "ico8":
        del P
        gc.collect()
        return
    dt = 200.0
    st = oa.TimeStepper("RungeKutta4", dt, P.tend, P.aux, P.mesh, None, P.tracers)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    st.do_step(P.state)
    oa.device_synchronize()
    P.oracle.step("rk4", ost, dt)
    h, u = P.state.copy_to_host(0)
    assert np.isfinite(ost["h"][0][: m.NCellsOwned]).all()
    check("h", h, ost["h"][0], m.NCellsOwned)
    check("u", u, ost["u"][0], m.NEdgesOwned)
    check("tr", P.tracers.copy_to_host(0), ost["tr"][0], m.NCellsOwned)
    del P
    gc.collect()
