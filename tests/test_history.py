"""History output of the state, tracers and auxiliary fields (omega_amd/csrc/History.cpp; reference: the History
IOStream, Default.yml:115-127, field names / groups from auxiliaryVars/*.cpp and OceanState.cpp:190-234), and a
restart that is checked against the ORACLE's state after the same number of steps."""
import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import planar_hex
from tests.problem import Problem

pytestmark = pytest.mark.gpu

AUX = {"KineticEnergyCell": "C", "VelocityDivCell": "C", "FluxLayerThickEdge": "E", "MeanLayerThickEdge": "E", "SshCell": "C",
       "RelVortVertex": "V", "NormRelVortVertex": "V", "NormPlanetVortVertex": "V", "NormRelVortEdge": "E",
       "NormPlanetVortEdge": "E", "Del2Edge": "E", "Del2DivCell": "C", "Del2RelVortVertex": "V"}


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert oa.device_count() > 0
    oa.device_init(0)


def _glob(P, kind, arr_local, n_owned, ids):
    out = np.zeros((P.g[kind],) + arr_local.shape[1:])
    out[ids[:n_owned] - 1] = arr_local[:n_owned]
    return out


def test_history_after_a_fused_step_holds_fresh_auxiliary_fields(tmp_path):
    """After fused RK4 steps most auxiliary arrays on the device are stale (never materialised); the dump must hold
    the fields of the state being written: every variable equals the oracle's AuxiliaryState::computeAll on that
    state, bit for bit, under the reference's names, dimensions and units."""
    K, NT = 20, 2        # 20 levels: device rows padded to 32, file rows compact
    P = Problem(planar_hex(24, 20, 30e3), K, NT, local_order="curve")
    st = oa.TimeStepper("RungeKutta4", 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
    for _ in range(2):
        st.do_step(P.state)
    oa.device_synchronize()
    path = str(tmp_path / "ocn.hist.nc")
    n = oa.write_history(path, P.decomp, P.state, P.tracers, P.aux, "State, Tracers, AuxiliaryState", st.time)
    assert n == 2 + 1 + 18
    h, u = P.state.copy_to_host(0)
    tr = P.tracers.copy_to_host(0)
    P.oracle.compute_all_aux(h, u, tr)
    f = oa.MeshFile(path, mesh=False)
    m = P.mesh
    assert f.dim("NCells") == P.g["nCells"] and f.dim("NVertLayers") == K and f.dim("NTracers") == NT
    assert f.read("SimulationTime")[0] == 1200.0
    own = {"C": (m.NCellsOwned, P.cell_id, "nCells"), "E": (m.NEdgesOwned, P.edge_id, "nEdges"),
           "V": (m.NVerticesOwned, P.vertex_id, "nVertices")}
    got = f.read("LayerThickness").reshape(-1, K)
    assert np.array_equal(got, _glob(P, "nCells", h, m.NCellsOwned, P.cell_id))
    got = f.read("NormalVelocity").reshape(-1, K)
    assert np.array_equal(got, _glob(P, "nEdges", u, m.NEdgesOwned, P.edge_id))
    got = f.read("Tracers").reshape(NT, -1, K)
    for t in range(NT):
        assert np.array_equal(got[t], _glob(P, "nCells", tr[t], m.NCellsOwned, P.cell_id))
    for name, el in AUX.items():
        no, ids, kind = own[el]
        ref = _glob(P, kind, P.oracle.aux[name], no, ids)
        assert np.array_equal(f.read(name).reshape(-1, K), ref), name
    for name, el in (("HTracersEdge", "E"), ("Del2TracersCell", "C")):
        no, ids, kind = own[el]
        got = f.read(name).reshape(NT, -1, K)
        for t in range(NT):
            assert np.array_equal(got[t], _glob(P, kind, P.oracle.aux[name][t], no, ids)), name


def test_history_contents_select_fields_and_reject_unknown_names(tmp_path):
    P = Problem(planar_hex(16, 16, 30e3), 4, 1)
    path = str(tmp_path / "h.nc")
    assert oa.write_history(path, P.decomp, P.state, P.tracers, P.aux, "Tracers,State,SshCell") == 4   # Default.yml History
    f = oa.MeshFile(path, mesh=False)
    ssh = f.read("SshCell").reshape(-1, 4)
    bd = P.mesh.get_array("BottomDepth")
    ref = np.zeros_like(ssh)
    ref[P.cell_id[:-1] - 1] = P.h[:-1] - bd[:-1, None]
    assert np.array_equal(ssh, ref)
    with pytest.raises(KeyError):
        f.read("KineticEnergyCell")
    with pytest.raises(oa.OmegaAmdError, match="no field or field group"):
        oa.write_history(path, P.decomp, P.state, P.tracers, P.aux, "State,NoSuchField")


def test_restart_continues_like_the_oracle(tmp_path):
    """2 steps, restart dump, load into fresh objects under ANOTHER local numbering, 2 more steps: equal to 4 steps
    of the CPU oracle (not merely to the uninterrupted GPU run)."""
    g = planar_hex(20, 16, 30e3)
    K, NT, dt = 6, 2, 600.0
    path = str(tmp_path / "restart.nc")
    P = Problem(g, K, NT)
    st = oa.TimeStepper("RungeKutta4", dt, P.tend, P.aux, P.mesh, None, P.tracers)
    ost = P.oracle.make_state(P.h, P.u, P.tr)
    for i in range(2):
        st.do_step(P.state)
    oa.device_synchronize()
    oa.write_restart(path, P.decomp, P.state, P.tracers, K, NT, st.time, 2)
    Q = Problem(g, K, NT, local_order="curve")
    t, n = oa.read_restart(path, Q.decomp, Q.mesh, Q.state, Q.tracers, K, NT)
    assert (t, n) == (2 * dt, 2)
    st2 = oa.TimeStepper("RungeKutta4", dt, Q.tend, Q.aux, Q.mesh, None, Q.tracers)
    st2.set_start_time(t)
    for i in range(2):
        st2.do_step(Q.state)
    oa.device_synchronize()
    for i in range(4):
        P.oracle.step("rk4", ost, dt, sim_time=i * dt)
    h, u = Q.state.copy_to_host(0)
    tr = Q.tracers.copy_to_host(0)
    # the oracle state lives in P's local numbering, the restarted run in Q's: compare through the global ids
    nc, ne = Q.mesh.NCellsOwned, Q.mesh.NEdgesOwned
    oh = np.zeros((g["nCells"], K)); oh[P.cell_id[:-1] - 1] = ost["h"][0][:-1]
    ou = np.zeros((g["nEdges"], K)); ou[P.edge_id[:-1] - 1] = ost["u"][0][:-1]
    assert np.array_equal(h[:nc], oh[Q.cell_id[:nc] - 1]) and np.array_equal(u[:ne], ou[Q.edge_id[:ne] - 1])
    for l in range(NT):
        ot = np.zeros((g["nCells"], K)); ot[P.cell_id[:-1] - 1] = ost["tr"][0][l, :-1]
        assert np.array_equal(tr[l, :nc], ot[Q.cell_id[:nc] - 1])
