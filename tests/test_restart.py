"""Restart files (omega_amd/csrc/MeshIO.cpp: RestartFile; reference: the RestartWrite / InitialState IOStreams,
Default.yml:91-127): rows written by global id from any partition, read back by any other partition."""
import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import planar_hex

C = oa.C


def _rows(decomp, name, n):
    return np.ascontiguousarray(decomp.get_array(name)[:n], dtype=np.int32)


def test_rows_written_by_three_ranks_are_read_by_two(tmp_path):
    g = planar_hex(10, 8, 1.0)
    gm = oa.GlobalMesh(g)
    K, NT = 5, 2
    path = str(tmp_path / "restart.nc").encode()
    rng = np.random.default_rng(0)
    H, U = rng.random((g["nCells"], K)), rng.random((g["nEdges"], K))
    T = rng.random((NT, g["nCells"], K))
    L = oa.lib()
    oa._chk(L.omg_restart_create(path, C.c_int64(g["nCells"]), C.c_int64(g["nEdges"]), K, NT, C.c_double(1234.5), C.c_int64(7)))
    for r in range(3):
        d = oa.Decomp(gm, 3, r, 2)
        f = C.c_void_p()
        oa._chk(L.omg_restart_open(path, 1, C.byref(f)))
        cid, eid = _rows(d, "CellID", d.get_int("NCellsOwned")), _rows(d, "EdgeID", d.get_int("NEdgesOwned"))
        oa._chk(L.omg_restart_write_rows(f, b"layerThickness", 0, oa._pi(cid), C.c_int64(len(cid)), oa._pd(np.ascontiguousarray(H[cid - 1]))))
        oa._chk(L.omg_restart_write_rows(f, b"normalVelocity", 0, oa._pi(eid), C.c_int64(len(eid)), oa._pd(np.ascontiguousarray(U[eid - 1]))))
        for l in range(NT):
            oa._chk(L.omg_restart_write_rows(f, b"tracers", l, oa._pi(cid), C.c_int64(len(cid)), oa._pd(np.ascontiguousarray(T[l, cid - 1]))))
        L.omg_restart_close(f)
    for r in range(2):
        d = oa.Decomp(gm, 2, r, 3)
        m = oa.HorzMesh(d, K, host_only=True)
        f = C.c_void_p()
        oa._chk(L.omg_restart_open(path, 0, C.byref(f)))
        info = [C.c_int64(), C.c_int64(), C.c_int(), C.c_int(), C.c_double(), C.c_int64()]
        oa._chk(L.omg_restart_info(f, *[C.byref(x) for x in info]))
        assert [x.value for x in info] == [g["nCells"], g["nEdges"], K, NT, 1234.5, 7]
        cid, eid = _rows(d, "CellID", m.NCellsAll), _rows(d, "EdgeID", m.NEdgesAll)   # owned AND halo
        out = np.empty((len(cid), K))
        oa._chk(L.omg_restart_read_rows(f, b"layerThickness", 0, oa._pi(cid), C.c_int64(len(cid)), oa._pd(out)))
        assert np.array_equal(out, H[cid - 1])
        out = np.empty((len(eid), K))
        oa._chk(L.omg_restart_read_rows(f, b"normalVelocity", 0, oa._pi(eid), C.c_int64(len(eid)), oa._pd(out)))
        assert np.array_equal(out, U[eid - 1])
        out = np.empty((len(cid), K))
        oa._chk(L.omg_restart_read_rows(f, b"tracers", 1, oa._pi(cid), C.c_int64(len(cid)), oa._pd(out)))
        assert np.array_equal(out, T[1, cid - 1])
        with pytest.raises(oa.OmegaAmdError):
            oa._chk(L.omg_restart_read_rows(f, b"tracers", NT, oa._pi(cid), C.c_int64(1), oa._pd(out)))
        L.omg_restart_close(f)


@pytest.mark.gpu
def test_restarted_run_continues_bit_for_bit(tmp_path):
    """2 steps + dump + load into fresh objects + 2 steps == 4 steps, on the GPU."""
    from tests.problem import Problem
    assert oa.device_count() > 0
    oa.device_init(0)
    g = planar_hex(20, 16, 30e3)
    K, NT, dt = 6, 2, 600.0
    path = str(tmp_path / "restart.nc")

    def fresh():
        P = Problem(g, K, NT, oracle=False)
        return P, oa.TimeStepper("RungeKutta4", dt, P.tend, P.aux, P.mesh, None, P.tracers)
    P, st = fresh()
    for _ in range(2):
        st.do_step(P.state)
    oa.device_synchronize()
    oa.write_restart(path, P.decomp, P.state, P.tracers, K, NT, st.time, 2)
    for _ in range(2):
        st.do_step(P.state)
    oa.device_synchronize()
    h4, u4 = P.state.copy_to_host(0)
    tr4 = P.tracers.copy_to_host(0)
    Q, st2 = fresh()
    t, n = oa.read_restart(path, Q.decomp, Q.mesh, Q.state, Q.tracers, K, NT)
    assert (t, n) == (2 * dt, 2)
    st2.set_start_time(t)
    for _ in range(2):
        st2.do_step(Q.state)
    oa.device_synchronize()
    h, u = Q.state.copy_to_host(0)
    tr = Q.tracers.copy_to_host(0)
    assert np.array_equal(h, h4) and np.array_equal(u, u4) and np.array_equal(tr, tr4)
    assert st2.time == 4 * dt
