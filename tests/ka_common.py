"""Test-harness helpers equivalent to the reference's test/ocn/OceanTestCommon.h
(setScalar :72-160, setVectorEdge :166-301 planar branch, computeErrors :399-547,
isApprox :14-23), in numpy, on a single-rank local mesh (oracle.Mesh)."""
import numpy as np

PI = np.pi
LX = 1.0
LY = np.sqrt(3.0) / 2.0

_OWN = {"Cell": "NCellsOwned", "Edge": "NEdgesOwned", "Vertex": "NVerticesOwned"}
_SIZE = {"Cell": "NCellsSize", "Edge": "NEdgesSize", "Vertex": "NVerticesSize"}


def set_scalar(M, K, f, el, rows=None, ntr=None):
    """Fill owned elements with f(X, Y) at every level (halo = none on one rank)."""
    n = M.a[_OWN[el]]
    v = f(M.a["X" + el][:n], M.a["Y" + el][:n])
    rows = M.a[_SIZE[el]] if rows is None else rows
    shape = (rows, K) if ntr is None else (ntr, rows, K)
    out = np.zeros(shape)
    out[..., :n, :] = np.asarray(v)[:, None]
    return out


def set_scalar_1d(M, f, el):
    n = M.a[_OWN[el]]
    out = np.zeros(M.a[_SIZE[el]])
    out[:n] = f(M.a["X" + el][:n], M.a["Y" + el][:n])
    return out


def edge_component(M, fx, fy, comp):
    n = M.NEdgesOwned
    X, Y, ang = M.XEdge[:n], M.YEdge[:n], M.AngleEdge[:n]
    if comp == "Normal":
        return np.cos(ang) * fx(X, Y) + np.sin(ang) * fy(X, Y)
    return -np.sin(ang) * fx(X, Y) + np.cos(ang) * fy(X, Y)


def set_vector_edge(M, K, fx, fy, comp="Normal", rows=None):
    v = edge_component(M, fx, fy, comp)
    rows = M.NEdgesSize if rows is None else rows
    out = np.zeros((rows, K))
    out[: M.NEdgesOwned] = v[:, None]
    return out


def set_vector_edge_1d(M, fx, fy, comp="Normal"):
    out = np.zeros(M.NEdgesSize)
    out[: M.NEdgesOwned] = edge_component(M, fx, fy, comp)
    return out


def compute_errors(M, num, exact, el):
    """(LInf, L2) normalised error measures over owned elements."""
    n = M.a[_OWN[el]]
    if el == "Cell":
        A = M.AreaCell[:n]
    elif el == "Vertex":
        A = M.AreaTriangle[:n]
    else:
        A = M.DcEdge[:n] * M.DvEdge[:n] / 2
    if num.ndim == 1:
        num, exact = num[:n, None], exact[:n, None]
    else:
        num, exact = num[..., :n, :], exact[..., :n, :]
    e = np.abs(num - exact)
    s = np.abs(exact)
    linf = e.max()
    if s.max() > 0:
        linf /= s.max()
    A = A[:, None]
    l2e, l2s = (A * e * e).sum(), (A * s * s).sum()
    l2 = np.sqrt(l2e / l2s) if l2s > 0 else np.sqrt(l2e)
    return float(linf), float(l2)


def is_approx(x, y, rtol, atol=0.0):
    if not (np.isfinite(x) and np.isfinite(y)):
        return False
    return abs(x - y) <= max(atol, rtol * max(abs(x), abs(y)))


def check_errors(name, got, expected, rtol, atol=0.0):
    assert is_approx(got[0], expected[0], rtol, atol), f"{name} LInf: expected {expected[0]!r} got {got[0]!r}"
    assert is_approx(got[1], expected[1], rtol, atol), f"{name} L2: expected {expected[1]!r} got {got[1]!r}"
