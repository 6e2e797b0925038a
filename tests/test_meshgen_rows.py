"""Host-side helpers added in round 4: the per-rank state synthesis bench.py uses at every N, and the generator of meshes
whose per-cell lists are not in ring order."""
import numpy as np

from omega_amd.meshgen import permute_cell_slots, planar_hex, synthetic_state_rows
from tests.meshes import named_mesh


def test_state_rows_depend_on_the_global_id_only():
    """synthetic_state_rows: a row's values are a function of (global id, level) -- whatever subset, order or tracer
    selection it is asked for in -- so an N = 1 and an N = 8 run of bench.py start from the same bits without any rank
    holding a global array."""
    g = planar_hex(24, 20, 30.0e3)
    K, NT = 12, 5
    allc, alle = np.arange(g["nCells"]), np.arange(g["nEdges"])
    h, u, tr = synthetic_state_rows(g, K, NT, allc, alle)
    assert h.shape == (g["nCells"], K) and u.shape == (g["nEdges"], K) and tr.shape == (NT, g["nCells"], K)
    assert h.min() > 1.0 and np.isfinite(u).all()
    rng = np.random.default_rng(0)
    c, e = rng.permutation(allc)[:57], rng.permutation(alle)[:91]
    h2, u2, tr2 = synthetic_state_rows(g, K, NT, c, e)
    assert np.array_equal(h2, h[c]) and np.array_equal(u2, u[e]) and np.array_equal(tr2, tr[:, c])
    one = synthetic_state_rows(g, K, NT, c, e[:0], tracers=[3])[2]
    assert one.shape == (1, len(c), K) and np.array_equal(one[0], tr[3, c])
    # the noise is there (neighbouring levels differ irregularly) and differs between the fields
    assert np.abs(np.diff(h, axis=1)).std() > 0.01 and not np.array_equal(tr[0] - 2.0, tr[1] - 2.1)
    g2 = named_mesh("ico3")
    hs, us, _ = synthetic_state_rows(g2, 4, 0, np.arange(g2["nCells"]), np.arange(g2["nEdges"]), tracers=[])
    assert hs.min() > 1.0 and np.abs(us).max() < 2.0


def test_permuted_cell_slots_stay_consistent_but_leave_the_ring_order():
    g = planar_hex(16, 12, 30.0e3)
    p = permute_cell_slots(g, 0.2)
    bad = p["permutedCells"]
    assert 0 < len(bad) < g["nCells"]
    for c in bad[:20]:
        n = p["nEdgesOnCell"][c]
        # the same sets, and slot j of cellsOnCell is still the cell across slot j of edgesOnCell ...
        assert sorted(p["edgesOnCell"][c, :n]) == sorted(g["edgesOnCell"][c, :n])
        for j in range(n):
            e = p["edgesOnCell"][c, j]
            assert set(p["cellsOnEdge"][e]) == {c, p["cellsOnCell"][c, j]}
        # ... but consecutive slots no longer all share a vertex
        shares = [len(set(p["verticesOnEdge"][p["edgesOnCell"][c, j]]) & set(p["verticesOnEdge"][p["edgesOnCell"][c, (j + 1) % n]]))
                  for j in range(n)]
        assert min(shares) == 0
    good = np.setdiff1d(np.arange(g["nCells"]), bad)
    assert np.array_equal(p["edgesOnCell"][good], g["edgesOnCell"][good])


def test_vectorised_sphere_generator_equals_the_loop_generator_bit_for_bit():
    """omega_amd.meshgen.spherical_voronoi (vectorised in round 5: 96 s -> 12 s for the 163 842-cell Fibonacci sphere)
    against the loop-per-cell generator of rounds 1-4 (tests/meshgen_loops.py): every array identical -- connectivity,
    numbering, and every geometry value to the bit (the golden vectors of tests/golden were made on the latter's mesh).
    Covers Lloyd sweeps, 12-pentagon icosahedra, heptagons, and cells of 8-10 edges (random points)."""
    from omega_amd.meshgen import icosahedral_points, spherical_voronoi
    from tests.meshgen_loops import spherical_voronoi_loops
    rng = np.random.default_rng(3)
    p = rng.standard_normal((300, 3))
    p /= np.linalg.norm(p, axis=1)[:, None]
    for n, kw in ((0, dict(points=icosahedral_points(2), lloyd=2)), (300, dict(lloyd=4)), (1500, dict(lloyd=1, sort=False)),
                  (0, dict(points=p, lloyd=1)), (0, dict(points=icosahedral_points(3), lloyd=0))):
        a, b = spherical_voronoi(n, **kw), spherical_voronoi_loops(n, **kw)
        assert sorted(a) == sorted(b)
        for k in a:
            if isinstance(a[k], np.ndarray):
                assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
            else:
                assert a[k] == b[k], k
    assert a["maxEdges"] == 6 and spherical_voronoi(0, points=p, lloyd=0)["maxEdges"] >= 8
