"""Manufactured-solution custom tendencies on the CPU oracle: the source terms restate
CustomTendencyTerms.cpp:112-208 and, added through the Tendencies hooks (Tendencies.cpp:288-291,
416-419) with the steppers' stage times, make RK4 converge to the exact solution at second order."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import manufactured as ms


def run_oracle(nx, hours=6.0, kind="rk4", config=None):
    g = ms.mesh(nx)
    wx, wy = ms.wavelengths(g)
    K, NT = 1, 1
    M = O.Mesh.single_rank(g, K)
    orc = O.Oracle(M, NT, O.default_config(**(config or {})))
    orc.use_manufactured_solution(wx, wy, ms.ETA0)
    try:
        h, u, tr = ms.initial_state(M, K, NT, wx, wy)
        st = orc.make_state(h, u, tr)
        dt = 600.0 * 16 / nx
        nsteps = int(round(hours * 3600 / dt))
        for n in range(nsteps):
            orc.step(kind, st, dt, sim_time=n * dt)
        return ms.l2_error_h(M, st["h"][0], nsteps * dt, wx, wy), st
    finally:
        orc.use_manufactured_solution()


def test_source_terms_cancel_the_exact_solution_tendency():
    """With the exact state as input, RHS + source must equal d/dt of the exact solution up to the
    spatial truncation error, which shrinks 4x per refinement."""
    errs = []
    for nx in (16, 32):
        g = ms.mesh(nx)
        wx, wy = ms.wavelengths(g)
        M = O.Mesh.single_rank(g, 1)
        orc = O.Oracle(M, 1)
        orc.use_manufactured_solution(wx, wy, ms.ETA0)
        try:
            t = 1234.0
            h, u, tr = ms.initial_state(M, 1, 1, wx, wy, t)
            orc.set_time(t)
            hT, uT, _ = orc.compute_all_tendencies(h, u, tr)
            eps = 1.0
            hp = ms.initial_state(M, 1, 1, wx, wy, t + eps)[0]
            hm = ms.initial_state(M, 1, 1, wx, wy, t - eps)[0]
            dhdt = (hp - hm) / (2 * eps)
            errs.append(np.abs(hT[: M.NCellsAll, 0] - dhdt[: M.NCellsAll, 0]).max())
        finally:
            orc.use_manufactured_solution()
    assert errs[0] / errs[1] > 3.0, errs


def test_rk4_converges_at_second_order():
    e16, _ = run_oracle(16)
    e32, _ = run_oracle(32)
    e64, _ = run_oracle(64)
    r1, r2 = np.log2(e16 / e32), np.log2(e32 / e64)
    assert e64 < e32 < e16 < 0.6 * ms.ETA0 and e64 < 0.03 * ms.ETA0
    assert 1.8 < r1 < 2.3 and 1.8 < r2 < 2.3, (e16, e32, e64, r1, r2)


def test_rk2_and_forward_backward_orders():
    """RK2 converges at second order, Forward-Backward at first order (the orders the reference's
    TimeStepperTest.cpp:375-388 expects of the schemes); both would be off by O(1) with wrong stage times."""
    e = {k: [run_oracle(n, hours=2.0, kind=k)[0] for n in (16, 32)] for k in ("rk2", "fb")}
    assert 1.8 < np.log2(e["rk2"][0] / e["rk2"][1]) < 2.3, e
    assert 0.8 < np.log2(e["fb"][0] / e["fb"][1]) < 1.3, e
    assert e["rk2"][1] < 0.02 * ms.ETA0
