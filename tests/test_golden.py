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


@pytest.mark.gpu
@pytest.mark.parametrize("name,fmt", [("planar16x16_k4_nt1", "cdf2"), ("ico2_k6_nt2", "cdf5"),
                                      ("ico2_k6_nt2", "cdf2-omega-names")])
def test_hip_path_on_a_file_loaded_mesh_equals_the_golden_vectors(tmp_path, name, fmt):
    """Row (f)2 end to end on the GPU: the mesh and the initial state travel through an MPAS-convention
    NetCDF file (1-based indices, padded connectivity, either name convention), come back through the
    library's reader (omg_mesh_file_*: Decomp.cpp:108-395 readMesh / HorzMesh.cpp:424-523), Decomp and
    HorzMesh are built from the file's arrays, the state from its layerThickness / normalVelocity /
    tracer variables -- and the fused RHS and an RK4 step must equal the committed golden vectors."""
    import omega_amd as oa
    from omega_amd.meshgen import synthetic_state
    from tests.problem import to_local
    from tests.test_mesh_file import write_cdf5, write_scipy
    assert oa.device_count() > 0
    oa.device_init(0)
    ref = np.load(os.path.join(GOLDEN, name + ".npz"))
    make, K, NT = CASES[name]
    g = make()
    hg, ug, trg = synthetic_state(g, K, NT)
    path, state_path = str(tmp_path / f"{name}.nc"), str(tmp_path / f"{name}_init.nc")
    if fmt == "cdf5":
        write_cdf5(path, g)
    else:
        write_scipy(path, g, 2, omega_names=fmt.endswith("names"), K=K)
    # the initial state comes from its own file, as Omega reads mesh and InitialState through separate streams
    # (Default.yml:91-100)
    from scipy.io import netcdf_file
    with netcdf_file(state_path, "w", version=2) as f:
        f.createDimension("nCells", g["nCells"]), f.createDimension("nEdges", g["nEdges"])
        f.createDimension("nVertLevels", K)
        f.createVariable("initThickness", "f8", ("nCells", "nVertLevels"))[:] = hg
        f.createVariable("initVelocity", "f8", ("nEdges", "nVertLevels"))[:] = ug
        for t in range(NT):
            f.createVariable(f"tracer{t}", "f8", ("nCells", "nVertLevels"))[:] = trg[t]
    mf, sf = oa.MeshFile(path), oa.MeshFile(state_path, mesh=False)
    assert sf.dim("nVertLevels") == K
    decomp = oa.Decomp(mf.gm, 1, 0, 3)
    mesh = oa.HorzMesh(decomp, K)
    cid, eid = decomp.get_array("CellID"), decomp.get_array("EdgeID")
    nC, nE = g["nCells"], g["nEdges"]
    h = to_local(sf.read("initThickness").reshape(nC, K), cid, mesh.NCellsSize)
    u = to_local(sf.read("initVelocity").reshape(nE, K), eid, mesh.NEdgesSize)
    tr = to_local(np.stack([sf.read(f"tracer{t}").reshape(nC, K) for t in range(NT)]), cid, mesh.NCellsSize)
    ci, ei = cid[: mesh.NCellsOwned] - 1, eid[: mesh.NEdgesOwned] - 1
    nc, ne = mesh.NCellsOwned, mesh.NEdgesOwned
    for kind, key in ((None, None), ("RungeKutta4", "rk4")):
        state, tracers = oa.OceanState(mesh, None, K, 2), oa.Tracers(mesh, None, K, NT, 2)
        aux, tend = oa.AuxiliaryState(mesh, None, K, NT), oa.Tendencies(mesh, K, NT, oa.default_config())
        state.copy_to_device(h, u, 0)
        tracers.copy_to_device(tr, 0)
        if kind is None:
            tend.compute_all_tendencies(state, aux, tracers)
            oa.device_synchronize()
            assert np.array_equal(tend.get(0)[:nc], ref["hTend"][ci])
            assert np.array_equal(tend.get(1)[:ne], ref["uTend"][ei])
            assert np.array_equal(tend.get(2)[:NT, :nc], ref["trTend"][:, ci])
        else:
            st = oa.TimeStepper(kind, 600.0, tend, aux, mesh, None, tracers)
            st.do_step(state)
            oa.device_synchronize()
            hh, uu = state.copy_to_host(0)
            assert np.array_equal(hh[:nc], ref[key + "_h"][ci])
            assert np.array_equal(uu[:ne], ref[key + "_u"][ei])
            assert np.array_equal(tracers.copy_to_host(0)[:NT, :nc], ref[key + "_tr"][:, ci])
