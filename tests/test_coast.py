"""Land boundaries.  Every mesh BASELINE.json names (QU240, EC30to60, QU30, oRRS18to6) is a CULLED ocean mesh, and the
reference's own RHS tests run on one (test/CMakeLists.txt:52-63 OmegaMesh.nc; test/ocn/TendenciesTest.cpp:172-212).
What only a coast exercises: EdgeMask = 0 where a cell of the edge is missing (src/ocn/HorzMesh.cpp:581-602), missing
neighbours mapped to the zero sentinel row (src/base/Decomp.cpp:553-574), CellsOnVertex / EdgesOnVertex holes in
VorticityAuxVars.h:24-59, zero entries of EdgesOnEdge kept in place (Decomp.cpp:2187-2199).

CPU part: the culled-mesh generator's invariants, the oracle's behaviour on a coast (no flux through it, masks), the
product's host Decomp on culled meshes.  GPU part: every kernel structure against the oracle, bit for bit, on owned
elements that SIT ON the coast; which kernel paths such a mesh takes; steppers; partition lines crossing the coast;
the mesh-file path.  (tests/test_gpu_parity.py holds the culled cases of the big parametrised parity tests, so the
child runs of tests/test_00_multirank_gpu.py with forced kernel structures cover them too.)"""
import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import cull, coast_mask, planar_hex, synthetic_state, zero_boundary_velocity
from oracle import oracle as O
from tests.meshes import named_mesh, COAST_KINDS
from tests.problem import Problem, to_local

MESHES = ["hex24x20", "ico3", "fib700"]


# ------------------------------------------------------------------------------------------------------------------
# CPU: generator invariants
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("raw", [False, True])
@pytest.mark.parametrize("kind", COAST_KINDS)
@pytest.mark.parametrize("base", MESHES)
def test_culled_mesh_is_consistent(base, kind, raw):
    g0 = named_mesh(base)
    g = named_mesh(f"{base}_coast_{kind}" + ("_raw" if raw else ""))
    nC, nE, nV, ME = g["nCells"], g["nEdges"], g["nVertices"], g["maxEdges"]
    assert nC < g0["nCells"] and nE < g0["nEdges"]
    coe, voe, eoc, coc, voc = g["cellsOnEdge"], g["verticesOnEdge"], g["edgesOnCell"], g["cellsOnCell"], g["verticesOnCell"]
    cov, eov = g["cellsOnVertex"], g["edgesOnVertex"]
    assert ((coe >= 0).sum(1) >= 1).all() and (voe >= 0).all()            # no orphan edge; both end vertices survive
    assert ((cov >= 0).sum(1) >= 1).all()                                  # no orphan vertex
    bnd = (coe < 0).any(1)
    assert bnd.any() and np.array_equal(bnd.astype(np.int32), g["boundaryEdge"])
    if not raw:
        assert (coe[:, 0] >= 0).all()                                      # culler convention: surviving cell first
    else:
        assert (coe[bnd, 0] < 0).any()                                     # the other pattern is really there
    for c in range(nC):
        n = g["nEdgesOnCell"][c]
        assert n == g0["nEdgesOnCell"][g["cullCellMap"][c]]
        for j in range(n):
            e = eoc[c, j]
            assert e >= 0 and c in coe[e]                                  # every edge of a surviving cell survives
            other = coe[e, 1] if coe[e, 0] == c else coe[e, 0]
            assert coc[c, j] == other                                      # a removed neighbour is -1 IN PLACE
            v = voc[c, j]
            assert v >= 0 and c in cov[v]
    # geometry rides along unchanged
    assert np.array_equal(g["areaCell"], g0["areaCell"][g["cullCellMap"]])
    # vertices: an edge slot is a hole exactly when the edge lost both its cells
    for v in range(nV):
        for j in range(3):
            e = eov[v, j]
            if e >= 0:
                assert v in voe[e]
    assert (eov < 0).any() and (cov < 0).any()
    # edgesOnEdge: holes in place (entries of removed edges), everything else valid and symmetric in count
    eoe, ne = g["edgesOnEdge"], g["nEdgesOnEdge"]
    live = np.arange(2 * ME)[None, :] < ne[:, None]
    assert (eoe[~live] < 0).all()
    assert (eoe[live] < 0).any()                                           # holes inside the lists
    interior = ~bnd
    assert (eoe[interior][live[interior]] >= 0).all()                      # an interior edge keeps its whole stencil


def test_compact_edges_on_edge_variant():
    g = named_mesh("hex24x20_coast_mixed_compact")
    gh = named_mesh("hex24x20_coast_mixed")
    eoe, ne = g["edgesOnEdge"], g["nEdgesOnEdge"]
    live = np.arange(eoe.shape[1])[None, :] < ne[:, None]
    assert (eoe[live] >= 0).all() and (eoe[~live] < 0).all()
    assert (ne < gh["nEdgesOnEdge"]).any()
    for e in range(0, g["nEdges"], 7):
        a = [(int(x), float(w)) for x, w in zip(gh["edgesOnEdge"][e], gh["weightsOnEdge"][e]) if x >= 0]
        b = [(int(x), float(w)) for x, w in zip(eoe[e, : ne[e]], g["weightsOnEdge"][e, : ne[e]])]
        assert a == b


def _oracle_problem(name, K=3, NT=2, zero_bnd=True, **kw):
    g = named_mesh(name)
    P = Problem(g, K, NT, device=False, **kw)
    if zero_bnd:
        hg, ug, trg = synthetic_state(g, K, NT)
        P.u = to_local(zero_boundary_velocity(g, ug), P.edge_id, P.mesh.NEdgesSize)
    return g, P


@pytest.mark.parametrize("name", ["hex24x20_coast_mixed", "ico3_coast_lakes", "fib700_coast_ragged_raw"])
def test_oracle_on_a_coast_masks_and_conserves(name):
    """EdgeMask (HorzMesh.cpp:581-602): 0 exactly on the boundary edges, and every velocity tendency term carries it
    (TendencyTerms.h:70-340) -> zero velocity tendency there.  With no normal flow through the coast the thickness
    and tracer tendencies are in flux form: area-weighted sums vanish."""
    g, P = _oracle_problem(name)
    M = P.omesh
    eid = P.edge_id[: M.NEdgesOwned] - 1
    bnd = g["boundaryEdge"][eid] != 0
    assert np.array_equal(M.EdgeMask[: M.NEdgesOwned, 0] == 0.0, bnd)
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    assert np.isfinite(hT).all() and np.isfinite(uT).all() and np.isfinite(trT).all()
    assert (uT[: M.NEdgesOwned][bnd] == 0.0).all()
    assert np.abs(uT[: M.NEdgesOwned][~bnd]).max() > 0
    A = M.AreaCell[: M.NCellsOwned, None]
    for name_, T in (("h", hT), ("tr0", trT[0]), ("tr1", trT[1])):
        tot, mag = (A * T[: M.NCellsOwned]).sum(), (A * np.abs(T[: M.NCellsOwned])).sum()
        assert abs(tot) <= 1e-12 * mag, (name_, tot, mag)


@pytest.mark.parametrize("order", ["global", "kd"])
@pytest.mark.parametrize("name", ["hex24x20_coast_mixed", "hex24x20_coast_ragged_raw", "ico3_coast_strait",
                                  "fib700_coast_lakes_compact",
                                  # (r6) the mesh families of the GPU parity suite, not only the coasts: what the GPU tests
                                  # hand the oracle (tests/problem.py: the PRODUCT's localisation) gives the numbers of a
                                  # localisation the product had no part in -- in the reference's numbering and in the k-d one
                                  "hex16x16", "ico3", "fib300", "ico3_pad8", "hex24x20_perm5"])
def test_product_decomp_equals_the_numpy_localisation(name, order):
    """One rank: the product's host Decomp + HorzMesh arrays (missing -> sentinel, per-cell edge compaction, EdgesOnEdge
    holes in place, MaxEdges = largest valence present, optional k-d renumbering) feed the oracle the same numbers as the
    independent numpy localisation of the global mesh (oracle.single_rank_local_arrays), element by element through the
    global ids: the tendencies and one RK4 step."""
    g, P = _oracle_problem(name, zero_bnd=False, local_order=order)
    K, NT = P.K, P.NT
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    hT, uT, trT = hT.copy(), uT.copy(), trT.copy()
    Mg = O.Mesh.single_rank(g, K)
    og = O.Oracle(Mg, NT, O.default_config())
    hg, ug, trg = synthetic_state(g, K, NT)
    pad = lambda x: np.concatenate([x, np.zeros(x.shape[:-2] + (1, x.shape[-1]))], axis=-2)
    hTg, uTg, trTg = og.compute_all_tendencies(pad(hg), pad(ug), pad(trg))
    m = P.mesh
    assert np.array_equal(hT[: m.NCellsOwned], hTg[P.cell_id[: m.NCellsOwned] - 1])
    assert np.array_equal(uT[: m.NEdgesOwned], uTg[P.edge_id[: m.NEdgesOwned] - 1])
    assert np.array_equal(trT[:NT, : m.NCellsOwned], trTg[:NT, P.cell_id[: m.NCellsOwned] - 1])
    dt = 5.0 if name.startswith("fib") else 600.0
    stl, stg = P.oracle.make_state(P.h, P.u, P.tr), og.make_state(pad(hg), pad(ug), pad(trg))
    P.oracle.step("rk4", stl, dt)
    og.step("rk4", stg, dt)
    assert np.array_equal(stl["h"][0][: m.NCellsOwned], stg["h"][0][P.cell_id[: m.NCellsOwned] - 1])
    assert np.array_equal(stl["u"][0][: m.NEdgesOwned], stg["u"][0][P.edge_id[: m.NEdgesOwned] - 1])
    assert np.array_equal(stl["tr"][0][:NT, : m.NCellsOwned], stg["tr"][0][:NT, P.cell_id[: m.NCellsOwned] - 1])


@pytest.mark.parametrize("nparts", [2, 5])
@pytest.mark.parametrize("name", ["hex24x20_coast_mixed", "ico3_coast_ragged"])
def test_decomp_on_a_culled_mesh_owns_every_element_once(name, nparts):
    g = named_mesh(name)
    gm = oa.GlobalMesh(g)
    task = oa.partition_cells(gm, nparts, "graph")[0]
    tot = np.zeros(3, dtype=np.int64)
    for r in range(nparts):
        d = oa.Decomp(gm, nparts, r, 3, cell_task=task)
        for i, (arr, n) in enumerate((("CellID", "NCellsOwned"), ("EdgeID", "NEdgesOwned"), ("VertexID", "NVerticesOwned"))):
            tot[i] += d.get_array(arr)[: d.get_int(n)].astype(np.int64).sum()
    n = np.array([g["nCells"], g["nEdges"], g["nVertices"]], dtype=np.int64)
    assert np.array_equal(tot, n * (n + 1) // 2)


@pytest.mark.parametrize("world,extra", [
    (2, ["--no-del4", "--mesh", "hex24x20_coast_mixed"]),
    (3, ["--halo-width", 5, "--mesh", "hex24x20_coast_strait", "--levels", 3, "--partition", "graph"]),
    (3, ["--no-del4", "--mesh", "ico3_coast_lakes_raw", "--levels", 3, "--partition", "graph", "--local-order", "curve"]),
])
def test_partitioned_oracle_on_a_coast_matches_single_rank(world, extra):
    """Partition lines crossing the coast: host Decomp / Halo lists drive the partitioned oracle over gloo."""
    from tests.test_multirank_cpu import run_ranks
    outs = run_ranks("cpu", world, extra)
    assert all("OK" in o for o in outs)


# ------------------------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@pytest.fixture()
def _gpu():
    assert oa.device_count() > 0, "no HIP device: GPU tests need a real MI355X"
    oa.device_init(0)


def _check(name, got, ref, n):
    assert np.array_equal(got[..., :n, :], ref[..., :n, :]), \
        f"{name}: max abs diff {np.abs(got[..., :n, :] - ref[..., :n, :]).max():.3e}"


GPU_COAST = [f"{b}_coast_{k}" for b in ("hex32x24", "ico4") for k in COAST_KINDS] + \
            ["fib1500_coast_mixed", "fib1500_coast_ragged_raw", "hex32x24_coast_ragged_raw_compact",
             "ico4_pad8_coast_lakes", "hex32x24_pad8_coast_strait_compact"]


@gpu
@pytest.mark.parametrize("name", GPU_COAST)
def test_coastal_meshes_keep_the_fast_kernel_paths(_gpu, name):
    """A coast must not send the mesh to the generic kernels: the ring / cell-centric tables stay valid (the flags are
    per mesh), the boundary edges are the irregular ones (edge-centric list), and the fused RHS is bit-exact on every
    owned element -- the coastal ones included (counted)."""
    g = named_mesh(name)
    P = Problem(g, 6, 2)
    m = P.mesh
    for flag in ("PVChainOK", "CellPVOK", "CellPVFinalOK", "Del2RingOK", "Del2VertOK", "CellL1OK"):
        assert m.get_int(flag) == 1, (name, flag)
    nb = int(g["boundaryEdge"].sum())
    assert m.get_int("NIrregularEdges") == nb > 0
    for fused in (True, False):
        P.tend.set_fused(fused)
        P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
        oa.device_synchronize()
        hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
        _check("hTend", P.tend.get(0), hT, m.NCellsOwned)
        _check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
        _check("trTend", P.tend.get(2)[:2], trT[:2], m.NCellsOwned)
    # the comparison really covered coastal elements: cells with a missing neighbour, edges next to boundary edges
    coc = g["cellsOnCell"]
    live = np.arange(g["maxEdges"])[None, :] < g["nEdgesOnCell"][:, None]
    coastal_cells = ((coc < 0) & live).any(1).sum()
    assert coastal_cells > 0 and np.abs(hT[: m.NCellsOwned]).min() >= 0
    eid = P.edge_id[: m.NEdgesOwned] - 1
    assert (uT[: m.NEdgesOwned][g["boundaryEdge"][eid] != 0] == 0).all()


@gpu
@pytest.mark.parametrize("name", ["hex32x24_coast_mixed", "ico4_coast_lakes", "fib1500_coast_ragged_raw"])
def test_aux_arrays_on_a_coast(_gpu, name):
    """AuxiliaryState::computeAll, all arrays (reference-structured kernels): vertices with one or two missing cells,
    cells with missing neighbours, boundary edges."""
    from tests.test_gpu_parity import AUX_2D, OWNED
    P = Problem(named_mesh(name), 7, 2, config={"FluxThicknessUpwind": 1, "FluxTracerUpwind": 1})
    P.aux.compute_all(P.state, P.tracers)
    oa.device_synchronize()
    P.oracle.compute_all_aux(P.h, P.u, P.tr)
    m = P.mesh
    for nm in AUX_2D:
        _check(nm, P.aux.get(nm), P.oracle.aux[nm], getattr(m, OWNED[oa.AUX_SHAPES[nm]]))
    _check("HTracersEdge", P.aux.get("HTracersEdge"), P.oracle.aux["HTracersEdge"], m.NEdgesOwned)
    _check("Del2TracersCell", P.aux.get("Del2TracersCell"), P.oracle.aux["Del2TracersCell"], m.NCellsOwned)


@gpu
@pytest.mark.parametrize("kind,okind,fuse", [("RungeKutta4", "rk4", True), ("RungeKutta4", "rk4", False),
                                             ("RungeKutta2", "rk2", None), ("Forward-Backward", "fb", None)])
@pytest.mark.parametrize("name", ["hex32x24_coast_mixed", "ico4_coast_mixed", "hex24x20_coast_lakes_raw"])
def test_time_steppers_on_a_coast(_gpu, name, kind, okind, fuse):
    g = named_mesh(name)
    P = Problem(g, 6, 2)
    # an ocean state: no flow through the coast (the masked tendencies then keep it that way)
    hg, ug, trg = synthetic_state(g, 6, 2)
    P.u = to_local(zero_boundary_velocity(g, ug), P.edge_id, P.mesh.NEdgesSize)
    P.state.copy_to_device(P.h, P.u, 0)
    dt = 600.0
    st = oa.TimeStepper(kind, dt, P.tend, P.aux, P.mesh, None, P.tracers)
    if fuse is not None:
        st.set_option("FuseStageUpdates", fuse)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    m = P.mesh
    for step in range(3):
        st.do_step(P.state)
        oa.device_synchronize()
        P.oracle.step(okind, ost, dt)
        h, u = P.state.copy_to_host(0)
        tr = P.tracers.copy_to_host(0)
        assert np.isfinite(ost["h"][0]).all() and np.isfinite(ost["u"][0]).all()
        _check(f"h step {step}", h, ost["h"][0], m.NCellsOwned)
        _check(f"u step {step}", u, ost["u"][0], m.NEdgesOwned)
        _check(f"tr step {step}", tr, ost["tr"][0], m.NCellsOwned)
    eid = P.edge_id[: m.NEdgesOwned] - 1
    assert (u[: m.NEdgesOwned][g["boundaryEdge"][eid] != 0] == 0).all()      # still no flow through the coast


@gpu
def test_fifty_rk4_steps_on_a_coast_stay_on_the_oracles_bits(_gpu):
    """A longer run: 50 RK4 steps (dt = 600 s, 30 km cells, no flow through the coast) on a culled mesh -- the state is
    still bit-identical to the oracle's at the end, and finite."""
    g = named_mesh("hex32x24_coast_mixed")
    K, NT = 6, 2
    P = Problem(g, K, NT)
    hg, ug, trg = synthetic_state(g, K, NT)
    P.u = to_local(zero_boundary_velocity(g, ug), P.edge_id, P.mesh.NEdgesSize)
    P.state.copy_to_device(P.h, P.u, 0)
    st = oa.TimeStepper("RungeKutta4", 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    for _ in range(50):
        st.do_step(P.state)
        P.oracle.step("rk4", ost, 600.0)
    oa.device_synchronize()
    h, u = P.state.copy_to_host(0)
    tr = P.tracers.copy_to_host(0)
    m = P.mesh
    assert np.isfinite(ost["h"][0]).all() and np.isfinite(ost["u"][0]).all() and (ost["h"][0][: m.NCellsOwned] > 0.5).all()
    _check("h", h, ost["h"][0], m.NCellsOwned)
    _check("u", u, ost["u"][0], m.NEdgesOwned)
    _check("tr", tr, ost["tr"][0], m.NCellsOwned)


@gpu
def test_rk4_conserves_volume_and_tracer_content_in_a_closed_basin(_gpu):
    """RK4 on a culled mesh with u = 0 on the coast: total volume and tracer content (device double-double sums)
    are conserved to rounding over 5 steps."""
    g = named_mesh("hex48x40_coast_mixed")
    K, NT = 10, 2
    P = Problem(g, K, NT, oracle=False)
    hg, ug, trg = synthetic_state(g, K, NT)
    P.u = to_local(zero_boundary_velocity(g, ug), P.edge_id, P.mesh.NEdgesSize)
    P.state.copy_to_device(P.h, P.u, 0)
    m = P.mesh
    area = m.local_arrays()["AreaCell"][: m.NCellsOwned, None]
    vol0 = (area * P.h[: m.NCellsOwned]).sum()
    tr0 = (area * P.h[: m.NCellsOwned] * P.tr[:, : m.NCellsOwned]).sum(axis=(1, 2))
    st = oa.TimeStepper("RungeKutta4", 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
    for _ in range(5):
        st.do_step(P.state)
    oa.device_synchronize()
    h, u = P.state.copy_to_host(0)
    tr = P.tracers.copy_to_host(0)
    vol1 = (area * h[: m.NCellsOwned]).sum()
    tr1 = (area * h[: m.NCellsOwned] * tr[:, : m.NCellsOwned]).sum(axis=(1, 2))
    assert abs(vol1 - vol0) <= 1e-13 * abs(vol0)
    assert np.all(np.abs(tr1 - tr0) <= 1e-12 * np.abs(tr0))
    assert np.abs(h[: m.NCellsOwned] - P.h[: m.NCellsOwned]).max() > 1e-6      # and something happened


@gpu
def test_mesh_file_with_a_coast_through_the_reader(_gpu, tmp_path):
    """Culled mesh written in the MPAS file convention (1-based, 0 = missing, holes in place), read by the library
    (MeshIO.cpp), Decomp -> HorzMesh -> fused RHS on the GPU: same bits as the in-memory mesh and as the oracle."""
    from tests.test_mesh_file import write_scipy
    g = named_mesh("hex24x20_coast_mixed")
    path = str(tmp_path / "culled.nc")
    write_scipy(path, g, 2)
    mf = oa.MeshFile(path)
    gm = mf.gm
    K, NT = 5, 1
    d = oa.Decomp(gm, 1, 0, 3)
    mesh = oa.HorzMesh(d, K)
    P = Problem(g, K, NT)
    assert mesh.get_int("NIrregularEdges") == P.mesh.get_int("NIrregularEdges") > 0
    la, lb = mesh.local_arrays(), P.mesh.local_arrays()
    for k in ("CellsOnEdge", "EdgesOnEdge", "CellsOnVertex", "EdgesOnVertex", "CellsOnCell", "WeightsOnEdge"):
        assert np.array_equal(la[k], lb[k]), k
    cfg = oa.default_config()
    state = oa.OceanState(mesh, None, K, 2)
    tracers = oa.Tracers(mesh, None, K, NT, 2)
    aux = oa.AuxiliaryState(mesh, None, K, NT)
    tend = oa.Tendencies(mesh, K, NT, cfg)
    state.copy_to_device(P.h, P.u, 0)
    tracers.copy_to_device(P.tr, 0)
    tend.compute_all_tendencies(state, aux, tracers)
    oa.device_synchronize()
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    _check("hTend", tend.get(0), hT, mesh.NCellsOwned)
    _check("uTend", tend.get(1), uT, mesh.NEdgesOwned)
    _check("trTend", tend.get(2)[:NT], trT[:NT], mesh.NCellsOwned)


def _random_coast_cases(n, seed=77):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        base = str(rng.choice(["hex20x16", "hex28x18", "ico3", "fib700", "hex24x20_pad8"]))
        frac = float(rng.choice([0.03, 0.1, 0.25, 0.45, 0.6]))
        K = int(rng.choice([1, 3, 16, 17, 40, 64]))
        NT = int(rng.choice([0, 1, 3]))
        raw, compact = bool(rng.integers(2)), bool(rng.integers(2))
        cfg = {}
        if i % 2:
            for f in ("FluxThicknessUpwind", "FluxTracerUpwind", "VelHyperDiffTendencyEnable", "PVTendencyEnable",
                      "TracerHyperDiffTendencyEnable", "BottomDragTendencyEnable"):
                if rng.random() < 0.35:
                    cfg[f] = int(not getattr(O.default_config(), f))
        out.append((base, frac, int(rng.integers(1 << 30)), K, NT, raw, compact, cfg))
    return out


@gpu
@pytest.mark.parametrize("case", _random_coast_cases(32), ids=lambda c: f"{c[0]}_land{int(100 * c[1])}_K{c[3]}_NT{c[4]}_{'raw' if c[5] else 'cull'}{'_compact' if c[6] else ''}_{len(c[7])}opts")
def test_random_coasts(_gpu, case):
    """Random land (3 ... 60 % of the cells, each cell independently: isolated ocean cells, one-cell channels, every
    vertex and edge pattern a culler can leave), both boundary-edge conventions, holes kept or compacted, random level /
    tracer counts and option sets: fused RHS and one RK4 step against the oracle, bit for bit on owned elements."""
    base, frac, seed, K, NT, raw, compact, cfg = case
    g0 = named_mesh(base)
    keep = np.random.default_rng(seed).random(g0["nCells"]) >= frac
    g = cull(g0, keep, first_cell_valid=not raw, compact_edges_on_edge=compact)
    P = Problem(g, K, NT, config=cfg)
    m = P.mesh
    assert m.get_int("NIrregularEdges") >= int(g["boundaryEdge"].sum()) > 0
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    _check("hTend", P.tend.get(0), hT, m.NCellsOwned)
    _check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
    if NT:
        _check("trTend", P.tend.get(2)[:NT], trT[:NT], m.NCellsOwned)
    if not base.startswith("fib"):         # (a few very short edges: not a stable step at dt = 600 s)
        hg, ug, trg = synthetic_state(g, K, NT)
        P.u = to_local(zero_boundary_velocity(g, ug), P.edge_id, m.NEdgesSize)
        P.state.copy_to_device(P.h, P.u, 0)
        st = oa.TimeStepper("RungeKutta4", 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
        ost = P.oracle.make_state(P.h, P.u, P.tr)
        st.do_step(P.state)
        oa.device_synchronize()
        P.oracle.step("rk4", ost, 600.0)
        h, u = P.state.copy_to_host(0)
        _check("h", h, ost["h"][0], m.NCellsOwned)
        _check("u", u, ost["u"][0], m.NEdgesOwned)
        if NT:
            _check("tr", P.tracers.copy_to_host(0)[:NT], ost["tr"][0][:NT], m.NCellsOwned)


@gpu
def test_qu30_sized_culled_mesh_at_full_size(_gpu):
    """BASELINE configs[3] as the real mesh is: culled.  800 x 800 hexagons with 28 % land removed ("continents" + one-cell
    islands: 459 955 cells, 14 910 boundary edges -- bench.py's qu30_coast workload), 80 levels, 6 tracers, numbered by
    Decomp along the curve: fused RHS element by element against the oracle, fast paths on."""
    import gc
    g0 = planar_hex(800, 800, 30.0e3)
    g = cull(g0, coast_mask(g0, "continents"))
    del g0
    P = Problem(g, 80, 6, local_order="curve")
    m = P.mesh
    for flag in ("CellL1OK", "CellPVOK", "CellPVFinalOK", "Del2RingOK", "Del2VertOK"):
        assert m.get_int(flag) == 1, flag
    assert m.get_int("NIrregularEdges") == int(g["boundaryEdge"].sum()) > 10000
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
    _check("hTend", P.tend.get(0), hT, m.NCellsOwned)
    _check("uTend", P.tend.get(1), uT, m.NEdgesOwned)
    _check("trTend", P.tend.get(2), trT, m.NCellsOwned)
    del P
    gc.collect()
