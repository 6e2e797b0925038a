"""Size-independent properties of the HIP path at the full BASELINE size (QU30-sized mesh: 462 400
cells x 80 levels x 6 tracers), where an element-wise oracle comparison would take minutes:

* the thickness tendency and the (thickness-weighted) tracer tendencies are divergences of edge
  fluxes, so their area-weighted global sums vanish on the periodic mesh (TendencyTerms.h:26-66,
  343-492) -- to rounding;
* the tracer tendency is linear in the tracer: scaling the tracers by a power of two scales it
  exactly (bit for bit);
* a time step conserves total volume and total tracer content;
* fused and reference-structured RHS agree bit for bit (both are separately compared with the oracle
  at small sizes in test_gpu_parity.py).
"""
import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import planar_hex, reorder_cells_morton
from tests.problem import Problem

pytestmark = pytest.mark.gpu


# bench.py's workloads "qu30" (BASELINE configs[3] on one GPU) and "orrs18to6_eighth" (the per-GPU share of
# configs[4]: 37 BGC tracers; one tracer array = 11 GB)
@pytest.fixture(scope="module", params=[("qu30", 680, 680, 30.0e3, 80, 6), ("orrs18to6_eighth", 680, 680, 6.0e3, 80, 37)],
                ids=lambda p: p[0])
def P(request):
    import gc
    assert oa.device_count() > 0
    oa.device_init(0)
    _, nx, ny, dc, K, NT = request.param
    g = reorder_cells_morton(planar_hex(nx, ny, dc))
    prob = Problem(g, K, NT, oracle=False)
    prob.dt = 600.0 * dc / 30.0e3   # Default.yml's 10 min at 30 km, scaled with the cell size
    yield prob
    del prob
    gc.collect()


def _sums(P, h_like, tr_like):
    nc = P.mesh.NCellsOwned
    a = P.mesh.get_array("AreaCell")[:nc, None]
    return (a * h_like[:nc]).sum(), (a[None] * tr_like[:, :nc]).sum(axis=(1, 2))


def test_flux_form_tendencies_sum_to_zero(P):
    P.tend.set_fused(True)
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, trT = P.tend.get(0), P.tend.get(2)
    sh, st = _sums(P, hT, trT)
    nh, nt = _sums(P, np.abs(hT), np.abs(trT))
    assert abs(sh) <= 1e-12 * nh
    assert np.all(np.abs(st) <= 1e-12 * nt)
    assert nh > 0 and np.all(nt > 0)


def test_fused_equals_reference_structured_at_full_size(P):
    P.tend.set_fused(True)
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    fused = [P.tend.get(i).copy() for i in range(3)]
    P.tend.set_fused(False)
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    m = P.mesh
    for i, n in ((0, m.NCellsOwned), (1, m.NEdgesOwned), (2, m.NCellsOwned)):
        assert np.array_equal(P.tend.get(i)[..., :n, :], fused[i][..., :n, :])
    P.tend.set_fused(True)


def test_tracer_tendency_is_linear_in_the_tracer(P):
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    t1 = P.tend.get(2).copy()
    P.tracers.copy_to_device(4.0 * P.tr, 0)
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    t4 = P.tend.get(2)
    P.tracers.copy_to_device(P.tr, 0)
    assert np.array_equal(t4[:, : P.mesh.NCellsOwned], 4.0 * t1[:, : P.mesh.NCellsOwned])


def test_rk4_step_conserves_volume_and_tracer_content(P):
    nc = P.mesh.NCellsOwned
    v0, c0 = _sums(P, P.h, P.tr * P.h[None])
    st = oa.TimeStepper("RungeKutta4", P.dt, P.tend, P.aux, P.mesh, None, P.tracers)
    st.do_step(P.state)
    oa.device_synchronize()
    h, _ = P.state.copy_to_host(0)
    tr = P.tracers.copy_to_host(0)
    v1, c1 = _sums(P, h, tr * h[None])
    assert np.isfinite(h[:nc]).all() and not np.array_equal(h[:nc], P.h[:nc])
    assert abs(v1 - v0) <= 1e-13 * abs(v0)
    assert np.all(np.abs(c1 - c0) <= 1e-13 * np.abs(c0))
    # restore the initial state for any test that follows
    P.state.copy_to_device(P.h, P.u, 0)
    P.tracers.copy_to_device(P.tr, 0)


def test_repeated_evaluations_are_bitwise_deterministic(P):
    """No atomics and no order-dependent accumulation anywhere on the path: 30 evaluations of the fused
    RHS and two runs of three RK4 steps from the same state give identical bits (a data race between
    workgroups or between the band / interior launches would show up here as a rare mismatch)."""
    P.tend.set_fused(True)
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    ref = [P.tend.get(i).copy() for i in range(3)]
    for _ in range(30 if P.NT <= 6 else 5):
        P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    for i in range(3):
        assert np.array_equal(P.tend.get(i), ref[i])
    runs = []
    for _ in range(2):
        P.state.copy_to_device(P.h, P.u, 0)
        P.tracers.copy_to_device(P.tr, 0)
        st = oa.TimeStepper("RungeKutta4", P.dt, P.tend, P.aux, P.mesh, None, P.tracers)
        for _ in range(3):
            st.do_step(P.state)
        oa.device_synchronize()
        h, u = P.state.copy_to_host(0)
        runs.append((h, u, P.tracers.copy_to_host(0)))
    for a, b in zip(*runs):
        assert np.array_equal(a, b)
    P.state.copy_to_device(P.h, P.u, 0)
    P.tracers.copy_to_device(P.tr, 0)


def test_arrays_allocated_right_before_their_first_use_on_a_non_blocking_stream(P):
    """The zero fill of a fresh device array is queued on the null stream, and work on a hipStreamNonBlocking stream is
    not ordered after the null stream: objects created right before their first use (here: the tendency arrays and the
    fused RHS's private intermediates, ~ 3 GB at this size; in a model run: the RK4 stepper's provisional state at the
    first step) must not be zeroed after the first kernels have written them.  (Found by a 4-rank bench rehearsal on one
    GPU: NaNs in one of five runs; Device.cpp: DeviceBuffer.)"""
    if P.NT > 8:
        pytest.skip("one size is enough (host copies of 37 tracer tendencies are 11 GB)")
    s = oa.Stream()
    for _ in range(3):
        tend = oa.Tendencies(P.mesh, P.K, P.NT, oa.default_config())
        tend.compute_all_tendencies(P.state, P.aux, P.tracers, stream=s)   # queued while the fills could still be running
        oa.device_synchronize()
        first = [tend.get(0), tend.get(1)]
        tend.compute_all_tendencies(P.state, P.aux, P.tracers, stream=s)
        oa.device_synchronize()
        again = [tend.get(0), tend.get(1)]
        assert np.abs(first[1]).max() > 0
        for a, b in zip(first, again):
            assert np.array_equal(a, b)
        del tend


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "stage-fused", "graphs", "rk2", "fb"])
def test_no_device_resource_is_created_inside_a_step(mode):
    """The reference allocates a stepper's provisional state in finalizeInit (RungeKutta4Stepper.cpp:43-64).  Here the
    constructors / omg_stepper_create create everything a step uses -- the second provisional buffer of the stage-fused
    form, the PV scratch of the fused RHS: the library's resource counter does not move across doStep 1..3 (the overlapped
    multi-rank form is checked in tests/mp_worker.py, where the communication stream, its events and the halo's message
    buffers join the list)."""
    oa.device_init(0)
    P = Problem(planar_hex(24, 20, 30.0e3), 8, 2, oracle=False)
    kind = {"rk2": "RungeKutta2", "fb": "Forward-Backward"}.get(mode, "RungeKutta4")
    st = oa.TimeStepper(kind, 300.0, P.tend, P.aux, P.mesh, None, P.tracers)
    stream = None
    if kind == "RungeKutta4":
        st.set_option("FuseStageUpdates", mode != "plain")
        if mode == "graphs":
            st.set_option("UseGraphs", True)
            stream = oa.Stream()
    n0 = oa.device_resource_count()
    assert n0 > 0
    for _ in range(3):
        st.do_step(P.state, stream=stream)
    oa.device_synchronize()
    assert oa.device_resource_count() == n0
    h, _ = P.state.copy_to_host(0)
    assert np.isfinite(h[: P.mesh.NCellsOwned]).all()
