"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle):
the oracle must keep reproducing them bit for bit (CPU), and the HIP path must equal them on the
same seeded inputs (GPU) -- identity numbering in the fixture, Decomp's local numbering on the device."""
import os

import numpy as np
import pytest

from tests.golden.make_golden import CASES, compute

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_the_golden_vectors(name):
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))
    _, _, _, out = compute(name)
    assert sorted(out) == sorted(ref.files)
    for k in ref.files:
        assert np.array_equal(out[k], ref[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_path_equals_the_golden_vectors(name):
    import omega_amd as oa
    from tests.problem import Problem
    assert oa.device_count() > 0
    oa.device_init(0)
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))
    make, K, NT = CASES[name]
    g = make()
    ci = lambda P: P.cell_id[: P.mesh.NCellsOwned] - 1
    ei = lambda P: P.edge_id[: P.mesh.NEdgesOwned] - 1
    P = Problem(g, K, NT)
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    nc, ne = P.mesh.NCellsOwned, P.mesh.NEdgesOwned
    assert np.array_equal(P.tend.get(0)[:nc], ref["hTend"][ci(P)])
    assert np.array_equal(P.tend.get(1)[:ne], ref["uTend"][ei(P)])
    assert np.array_equal(P.tend.get(2)[:NT, :nc], ref["trTend"][:, ci(P)])
    for kind, key in (("Forward-Backward", "fb"), ("RungeKutta4", "rk4")):
        P = Problem(g, K, NT)
        st = oa.TimeStepper(kind, 600.0, P.tend, P.aux, P.mesh, None, P.tracers)
        st.do_step(P.state)
        oa.device_synchronize()
        h, u = P.state.copy_to_host(0)
        tr = P.tracers.copy_to_host(0)
        assert np.array_equal(h[:nc], ref[key + "_h"][ci(P)]), kind
        assert np.array_equal(u[:ne], ref[key + "_u"][ei(P)]), kind
        assert np.array_equal(tr[:NT, :nc], ref[key + "_tr"][:, ci(P)]), kind
