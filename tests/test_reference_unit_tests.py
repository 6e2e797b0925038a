"""The reference's own unit tests for the containers on the path, restated against this backend (what they check, not
their code): test/ocn/StateTest.cpp:268-411 (time-level rotation with 2, 3 and 4 levels: after N updateTimeLevels() the
data filled at the current / new level is found at level -N / 1-N, wrapped into [-(NTimeLevels-2), 1]),
test/ocn/TracersTest.cpp (the same for the tracer arrays) and test/ocn/HorzMeshTest.cpp:175-690 (counts summed over the
ranks, points on the sphere, lon / lat against x / y / z, area sums, dcEdge / dvEdge against great-circle distances,
angleEdge range, Coriolis parameters, weightsOnEdge bounds, edge signs against CellsOnEdge / VerticesOnEdge order)."""
import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import planar_hex
from tests.meshes import named_mesh


# ---------------------------------------------------------------------------------------------------- HorzMeshTest (host)
@pytest.mark.parametrize("nparts", [1, 3])
@pytest.mark.parametrize("name", ["ico4", "fib700_coast_lakes"])
def test_horz_mesh_like_the_reference_test(name, nparts):
    g = named_mesh(name)
    gm = oa.GlobalMesh(g)
    task = oa.partition_cells(gm, nparts, "graph")[0] if nparts > 1 else None
    R = float(np.sqrt(g["xCell"][0] ** 2 + g["yCell"][0] ** 2 + g["zCell"][0] ** 2))
    owned = np.zeros(3, dtype=np.int64)
    area = np.zeros(3)
    for r in range(nparts):
        d = oa.Decomp(gm, nparts, r, 3, cell_task=task)
        m = oa.HorzMesh(d, 4, host_only=True)
        L = m.local_arrays()
        nco, neo, nvo = m.NCellsOwned, m.NEdgesOwned, m.NVerticesOwned
        nca, nea, nva = m.NCellsAll, m.NEdgesAll, m.NVerticesAll
        owned += (nco, neo, nvo)
        # points on the sphere; lon / lat agree with x / y / z (HorzMeshTest.cpp:196-242, 262-306, 327-372)
        for el, n in (("Cell", nca), ("Edge", nea), ("Vertex", nva)):
            x, y, z = L["X" + el][:n], L["Y" + el][:n], L["Z" + el][:n]
            assert np.allclose(np.sqrt(x * x + y * y + z * z), R, rtol=1e-10)
            lon, lat = L["Lon" + el][:n], L["Lat" + el][:n]
            assert np.allclose(R * np.cos(lat) * np.cos(lon), x, atol=1e-6 * R) and np.allclose(R * np.sin(lat), z, atol=1e-6 * R)
            assert (np.abs(lat) <= np.pi / 2 + 1e-12).all() and (lon >= -1e-12).all() and (lon <= 2 * np.pi + 1e-12).all()
        # area sums over owned elements (:396-455)
        area += (L["AreaCell"][:nco].sum(), L["AreaTriangle"][:nvo].sum(), L["KiteAreasOnVertex"][:nvo].sum())
        # dcEdge / dvEdge = great-circle distances where both ends are local (:460-515)
        def arc(lo1, la1, lo2, la2):
            return 2 * np.arcsin(np.sqrt(np.sin((la2 - la1) / 2) ** 2 + np.cos(la1) * np.cos(la2) * np.sin((lo2 - lo1) / 2) ** 2))
        coe, voe = L["CellsOnEdge"][:neo], L["VerticesOnEdge"][:neo]
        both = (coe < nca).all(axis=1)
        dc = R * arc(L["LonCell"][coe[both, 0]], L["LatCell"][coe[both, 0]], L["LonCell"][coe[both, 1]], L["LatCell"][coe[both, 1]])
        assert np.allclose(dc, L["DcEdge"][:neo][both], rtol=1e-6)
        bothv = (voe < nva).all(axis=1)
        dv = R * arc(L["LonVertex"][voe[bothv, 0]], L["LatVertex"][voe[bothv, 0]], L["LonVertex"][voe[bothv, 1]], L["LatVertex"][voe[bothv, 1]])
        assert np.allclose(dv, L["DvEdge"][:neo][bothv], rtol=1e-6)
        assert (np.abs(L["AngleEdge"][:neo]) <= np.pi + 1e-12).all()                                   # :520-531
        # Coriolis parameters 2 Omega sin(lat) (:536-592) -- the generator's convention is the reference test's
        for el, n, f in (("Cell", nco, "FCell"), ("Vertex", nvo, "FVertex"), ("Edge", neo, "FEdge")):
            assert np.allclose(L[f][:n], 2 * 7.29212e-5 * np.sin(L["Lat" + el][:n]), atol=1e-9)
        assert (np.abs(L["WeightsOnEdge"][:neo]) <= 1.0 + 1e-12).all()                                # :597-611
        # EdgeSignOnCell: -1 where the cell is the edge's first cell, +1 where it is the second (:616-649); EdgeSignOnVertex
        # the same with VerticesOnEdge (:654-686) -- from the derived arrays the kernels' coefficient tables are built from
        es, eoc, n_on = m.get_array("EdgeSignOnCell"), L["EdgesOnCell"], L["NEdgesOnCell"]
        for c in range(0, nco, 7):
            for j in range(n_on[c]):
                e = eoc[c, j]
                assert es[c, j] == (-1.0 if L["CellsOnEdge"][e, 0] == c else 1.0)
        ev, eov = m.get_array("EdgeSignOnVertex"), L["EdgesOnVertex"]
        for v in range(0, nvo, 7):
            for j in range(3):
                e = eov[v, j]
                if e < nea:
                    assert ev[v, j] == (-1.0 if L["VerticesOnEdge"][e, 0] == v else 1.0)
    assert list(owned) == [g["nCells"], g["nEdges"], g["nVertices"]]                                   # :178-192, 245-259, 310-324
    ocean = float(np.sum(g["areaCell"]))
    # cells tile the domain; triangles and kites tile it up to the coast's half-covered rim
    assert abs(area[0] - ocean) <= 1e-9 * ocean
    full = "coast" not in name
    assert abs(area[1] - ocean) / ocean < (1e-9 if full else 0.08) and abs(area[2] - ocean) / ocean < (1e-9 if full else 0.08)
    if full:
        assert abs(ocean - 4 * np.pi * R * R) / (4 * np.pi * R * R) < 1e-3


# ------------------------------------------------------------------------------------ StateTest / TracersTest (time levels)
@pytest.mark.gpu
@pytest.mark.parametrize("ntl", [2, 3, 4])
def test_time_level_rotation_like_the_reference_tests(ntl):
    oa.device_init(0)
    g = planar_hex(12, 12, 30.0e3)
    d = oa.Decomp(oa.GlobalMesh(g), 1, 0, 3)
    m = oa.HorzMesh(d, 5)
    K, NT = 5, 3
    rng = np.random.default_rng(7)
    hdef, udef = rng.random((m.NCellsSize, K)), rng.random((m.NEdgesSize, K))
    trdef = rng.random((NT, m.NCellsSize, K))
    ref, tst = oa.OceanState(m, None, K, ntl), oa.OceanState(m, None, K, ntl)
    rtr, ttr = oa.Tracers(m, None, K, NT, ntl), oa.Tracers(m, None, K, NT, ntl)
    cur, new = 0, 1
    for s, t in ((ref, rtr), (tst, ttr)):                  # current level: the default state; new level: default + 1
        s.copy_to_device(hdef, udef, cur)
        s.copy_to_device(hdef + 1, udef + 1, new)
        t.copy_to_device(trdef, cur)
        t.copy_to_device(trdef + 1, new)

    def same(level_ref, level_tst):
        hr, ur = ref.copy_to_host(level_ref)
        ht, ut = tst.copy_to_host(level_tst)
        return (np.array_equal(hr, ht) and np.array_equal(ur, ut)
                and np.array_equal(rtr.copy_to_host(level_ref), ttr.copy_to_host(level_tst)))
    assert same(cur, cur) and same(new, new) and not same(cur, new)
    for n in range(1, ntl):
        tst.update_time_levels()
        ttr.update_time_levels()
        oa.device_synchronize()
        nmin = -(ntl - 2)                                   # StateTest.cpp:362-377: levels shift one older, wrapping below nmin
        cur_u, new_u = cur - n, new - n
        if cur_u < nmin:
            cur_u += ntl
        if new_u < nmin:
            new_u += ntl
        assert same(cur, cur_u), (ntl, n, "current level")
        assert same(new, new_u), (ntl, n, "new level")
    # a level outside [-(NTimeLevels-2), 1] is an error code, not a wrap (OceanState.cpp:394-407)
    with pytest.raises(oa.OmegaAmdError):
        tst.copy_to_host(2)
    with pytest.raises(oa.OmegaAmdError):
        tst.copy_to_host(-(ntl - 1))


# ---------------------------------------------------------------------------------------------------------- TendenciesTest
@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ["ico5", "fib1500", "ico4_coast_lakes", "hex48x24_coast_mixed"])
def test_tendencies_like_the_reference_test(name, fused):
    """test/ocn/TendenciesTest.cpp:25-44, 150-212: the analytic state h = 2 + cos(lon) cos^4(lat), u from its vector field,
    tracers 2 - cos(lon) cos^4(lat); EVERY tendency variable filled with NaN first; after computeAllTendencies the sums over
    the owned elements are finite and non-zero -- i.e. every owned value was written, by the fused launches and by the
    reference-structured sequence (and, beyond the reference's test, they are the oracle's values)."""
    from tests.problem import Problem, poison_tendencies
    oa.device_init(0)
    g = named_mesh(name)
    K, NT = 60, 3
    P = Problem(g, K, NT)
    m = P.mesh
    L = m.local_arrays()
    R = 6371220.0
    sphere = "zCell" in g and float(np.abs(g["zCell"]).max()) > 0
    lonc, latc = (L["LonCell"], L["LatCell"]) if sphere else (2 * np.pi * L["XCell"] / max(L["XCell"].max(), 1.0), 0.3 * np.ones_like(L["XCell"]))
    lone, late = (L["LonEdge"], L["LatEdge"]) if sphere else (2 * np.pi * L["XEdge"] / max(L["XEdge"].max(), 1.0), 0.3 * np.ones_like(L["XEdge"]))
    h = np.zeros((m.NCellsSize, K))
    h[:] = (2 + np.cos(lonc) * np.cos(latc) ** 4)[:, None]
    ux = -R * np.sin(lone) ** 2 * np.cos(late) ** 3
    uy = -4 * R * np.sin(lone) * np.cos(lone) * np.cos(late) ** 3 * np.sin(late)
    u = np.zeros((m.NEdgesSize, K))
    u[:] = (1.0e-6 * (ux * np.cos(L["AngleEdge"]) + uy * np.sin(L["AngleEdge"])))[:, None]    # (scaled to m/s-sized values)
    tr = np.zeros((NT, m.NCellsSize, K))
    tr[:] = (2 - np.cos(lonc) * np.cos(latc) ** 4)[None, :, None] + 0.1 * np.arange(NT)[:, None, None]
    h[-1] = u[-1] = 0.0
    tr[:, -1] = 0.0
    P.state.copy_to_device(h, u, 0)
    P.tracers.copy_to_device(tr, 0)
    P.tend.set_fused(fused)
    poison_tendencies(P)
    P.tend.compute_all_tendencies(P.state, P.aux, P.tracers)
    oa.device_synchronize()
    hT, uT, trT = P.tend.get(0), P.tend.get(1), P.tend.get(2)
    nc, ne = m.NCellsOwned, m.NEdgesOwned
    for nm, a in (("LayerThickTend", hT[:nc]), ("NormVelTend", uT[:ne]), ("TraceTend", trT[:NT, :nc])):
        sm = float(a.sum())
        assert np.isfinite(a).all() and np.isfinite(sm) and sm != 0.0, nm
    oh, ou, otr = P.oracle.compute_all_tendencies(h, u, tr)
    assert np.array_equal(hT[:nc], oh[:nc]) and np.array_equal(uT[:ne], ou[:ne]) and np.array_equal(trT[:NT, :nc], otr[:NT, :nc])
