"""The reference's time-stepper known answer (O/test/timeStepping/TimeStepperTest.cpp:375-388): with every
built-in tendency off and the custom velocity tendency du/dt = -0.5 u (DecayVelocityTendency, :49-73), from
h = u = tracers = 1, integrating to T = 1 with dt = 0.2 and 0.1 (adjustTimeStep :252-263), the L-infinity error
of u against exp(-0.5 T) must converge at order 4 (RungeKutta4), 1 (Forward-Backward), 2 (RungeKutta2) +- 0.1 --
independent of the spatial operators.  Checked on the CPU oracle and, through the C ABI's custom-tendency
callback, on the HIP path (whose result must also equal the oracle's bit for bit)."""
import math

import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import planar_hex
from tests.problem import Problem

ALL_OFF = {k: 0 for k in ("ThicknessFluxTendencyEnable", "PVTendencyEnable", "KETendencyEnable", "SSHTendencyEnable",
                          "VelDiffTendencyEnable", "VelHyperDiffTendencyEnable", "TracerHorzAdvTendencyEnable",
                          "TracerDiffTendencyEnable", "TracerHyperDiffTendencyEnable", "WindForcingTendencyEnable",
                          "BottomDragTendencyEnable")}
COEFF, T_END, BASE_DT = 0.5, 1.0, 0.2
EXPECTED = [("RungeKutta4", "rk4", 4.0), ("Forward-Backward", "fb", 1.0), ("RungeKutta2", "rk2", 2.0)]


def _ones(P):
    h, u, tr = np.zeros_like(P.h), np.zeros_like(P.u), np.zeros_like(P.tr)
    h[:-1], u[:-1], tr[:, :-1] = 1.0, 1.0, 1.0
    return h, u, tr


def _steps(dt):
    n = int(math.ceil(T_END / dt))       # adjustTimeStep
    return n, T_END / n


def oracle_run(P, okind, dt):
    h, u, tr = _ones(P)
    st = P.oracle.make_state(h, u, tr)
    n, dt = _steps(dt)
    P.oracle.use_decay_velocity_tendency(COEFF)
    try:
        for i in range(n):
            P.oracle.step(okind, st, dt, sim_time=i * dt)
    finally:
        P.oracle.use_decay_velocity_tendency(None)
    return st


def linf_error(P, u):
    ne = P.mesh.NEdgesOwned
    return np.abs(u[:ne] - math.exp(-COEFF * T_END)).max()


@pytest.mark.parametrize("kind,okind,order", EXPECTED)
def test_oracle_time_steppers_converge_at_their_order(kind, okind, order):
    P = Problem(planar_hex(8, 8, 30e3), 1, 1, device=False, config=ALL_OFF)
    errs = []
    for dt in (BASE_DT, BASE_DT / 2):
        st = oracle_run(P, okind, dt)
        errs.append(linf_error(P, st["u"][0]))
        assert np.array_equal(st["h"][0][:-1], np.ones_like(st["h"][0][:-1]))    # no thickness tendency
        assert np.array_equal(st["tr"][0][:, :-1], np.ones_like(st["tr"][0][:, :-1]))
    rate = math.log2(errs[0] / errs[1])
    assert abs(rate - order) <= 0.1, (kind, errs, rate)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,okind,order", EXPECTED)
def test_hip_time_steppers_converge_at_their_order(kind, okind, order):
    assert oa.device_count() > 0
    oa.device_init(0)
    P = Problem(planar_hex(8, 8, 30e3), 1, 1, config=ALL_OFF)

    def decay(tend, h, u, nall, nsize, k, pitch, t, stream):   # NormalVelTend(IEdge, K) -= Coeff * NormalVelEdge(IEdge, K)
        oa.update_by_tend(tend, tend, u, -COEFF, nall, pitch, stream)     # rows of `pitch` values (padding swept too)
    P.tend.set_custom_tendency(1, decay)
    errs = []
    for dt0 in (BASE_DT, BASE_DT / 2):
        h, u, tr = _ones(P)
        P.state.copy_to_device(h, u, 0)
        P.tracers.copy_to_device(tr, 0)
        n, dt = _steps(dt0)
        st = oa.TimeStepper(kind, dt, P.tend, P.aux, P.mesh, None, P.tracers)
        for _ in range(n):
            st.do_step(P.state)
        oa.device_synchronize()
        hh, uu = P.state.copy_to_host(0)
        errs.append(linf_error(P, uu))
        ost = oracle_run(P, okind, dt0)
        ne, nc = P.mesh.NEdgesOwned, P.mesh.NCellsOwned
        assert np.array_equal(uu[:ne], ost["u"][0][:ne]) and np.array_equal(hh[:nc], ost["h"][0][:nc])
        assert np.array_equal(P.tracers.copy_to_host(0)[:, :nc], ost["tr"][0][:, :nc])
    P.tend.set_custom_tendency(1, None)
    rate = math.log2(errs[0] / errs[1])
    assert abs(rate - order) <= 0.1, (kind, errs, rate)


@pytest.mark.gpu
def test_change_time_step_keeps_the_model_time():
    """TimeStepper::changeTimeStep (TimeStepper.h:141-143), used by the convergence loop of the reference test."""
    assert oa.device_count() > 0
    oa.device_init(0)
    P = Problem(planar_hex(8, 8, 30e3), 1, 1, config=ALL_OFF)
    st = oa.TimeStepper("RungeKutta4", 0.2, P.tend, P.aux, P.mesh, None, P.tracers)
    st.do_step(P.state)
    st.change_time_step(0.1)
    st.do_step(P.state)
    oa.device_synchronize()
    assert abs(st.time - 0.3) < 1e-15
