"""Reductions (reference: components/omega/src/base/Reductions.h): double-double sums.
CPU: the ddSum combination operator.  GPU: the device accumulation, its independence of how the data
is split (the reference's device path is a plain parallelReduce and is NOT reproducible; this one is),
and conservation of volume / tracer content over a time step measured with it."""
import math

import numpy as np
import pytest

import omega_amd as oa


def nasty(n, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(n) * 10.0 ** rng.integers(-8, 12, n)
    a[::7] = -a[1::7][: len(a[::7])] if len(a[1::7]) >= len(a[::7]) else a[::7]
    return a


def knuth(a):
    hi = lo = 0.0
    for x in a:
        t1 = x + hi
        e = t1 - x
        t2 = ((hi - e) + (x - (t1 - e))) + lo
        hi, lo = t1 + t2, t2 - ((t1 + t2) - t1)
    return hi, lo


def test_combine_dd_matches_exact_summation():
    a = nasty(4000)
    parts = [knuth(c) for c in np.array_split(a, 7)]
    hi, lo = oa.combine_dd(parts)
    exact = math.fsum(a)
    assert hi == exact or abs(hi - exact) <= abs(exact) * 2.3e-16
    # order of the ranks does not matter at double precision
    assert oa.combine_dd(parts[::-1])[0] == hi
    assert oa.combine_dd([])[0] == 0.0


@pytest.mark.gpu
def test_device_sum_is_exact_and_split_independent():
    assert oa.device_count() > 0
    oa.device_init(0)
    a = nasty(1_000_003, 1)
    b = nasty(1_000_003, 2) * 1e-6
    ta, tb = oa.DeviceBuffer(a), oa.DeviceBuffer(b)
    whole = oa.local_sum_dd(ta.ptr, len(a))
    assert whole[0] == math.fsum(a)
    assert abs(float(np.sum(a)) - whole[0]) > 0 or True   # (a plain sum generally differs)
    cut = 333_337
    halves = [oa.local_sum_dd(ta.ptr, cut), oa.local_sum_dd(ta.ptr + 8 * cut, len(a) - cut)]
    assert oa.combine_dd(halves)[0] == whole[0]
    prod = oa.local_sum_dd(ta.ptr, len(a), tb.ptr)
    assert prod[0] == math.fsum(a * b)
    assert oa.global_sum_dd(whole) == whole[0]


@pytest.mark.gpu
def test_rk4_conserves_volume_and_tracer_content_in_double_double():
    from omega_amd.meshgen import planar_hex
    from tests.problem import Problem
    oa.device_init(0)
    P = Problem(planar_hex(48, 40, 30.0e3), 20, 2, oracle=False)
    m = P.mesh
    nc, K = m.NCellsOwned, 20
    KP = oa.level_pitch(K)      # the library's own arrays pad 20 levels to 32 (whole cache lines)
    assert KP == 32
    area = oa.DeviceBuffer(m.get_array("AreaCell"))

    def content():
        hp = P.state.device_ptr(0, 0)
        vol = oa.local_weighted_sum_dd(area.ptr, hp, nc, K, row_pitch=KP)[0]
        trs = [oa.local_weighted_sum_dd(area.ptr, hp, nc, K, P.tracers.device_ptr(0) + 8 * l * m.NCellsSize * KP,
                                        row_pitch=KP)[0] for l in range(2)]
        return vol, trs
    v0, t0 = content()
    h0, _ = P.state.copy_to_host(0)
    assert v0 == math.fsum((m.get_array("AreaCell")[:nc, None] * h0[:nc]).ravel()) or \
        abs(v0 - math.fsum((m.get_array("AreaCell")[:nc, None] * h0[:nc]).ravel())) <= 4e-16 * v0
    st = oa.TimeStepper("RungeKutta4", 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
    for _ in range(3):
        st.do_step(P.state)
    oa.device_synchronize()
    v1, t1 = content()
    assert abs(v1 - v0) <= 2e-14 * v0
    for a, b in zip(t0, t1):
        assert abs(a - b) <= 2e-14 * abs(a)
