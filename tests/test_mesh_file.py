"""MPAS mesh-file reader (omega_amd/csrc/MeshIO.cpp; reference: Decomp.cpp:108-395 readMesh,
HorzMesh.cpp:424-523): files written here in the three NetCDF classic formats -- CDF-1 / CDF-2 through
scipy.io.netcdf_file, CDF-5 through the small writer below -- with MPAS conventions (1-based indices,
0 = none, connectivity padded beyond nEdgesOnCell) under both name conventions must come back as the
mesh they were made from, and a Decomp / HorzMesh built from the file must equal one built from memory."""
import struct

import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import icosahedral_points, planar_hex, spherical_voronoi

INT_VARS = {"cellsOnCell": ("nCells", "maxEdges"), "edgesOnCell": ("nCells", "maxEdges"),
            "verticesOnCell": ("nCells", "maxEdges"), "cellsOnEdge": ("nEdges", "TWO"),
            "verticesOnEdge": ("nEdges", "TWO"), "edgesOnEdge": ("nEdges", "maxEdges2"),
            "cellsOnVertex": ("nVertices", "vertexDegree"), "edgesOnVertex": ("nVertices", "vertexDegree"),
            "nEdgesOnCell": ("nCells",), "nEdgesOnEdge": ("nEdges",)}
REAL_VARS = {"kiteAreasOnVertex": ("nVertices", "vertexDegree"), "weightsOnEdge": ("nEdges", "maxEdges2")}
for _el, _d in (("Cell", "nCells"), ("Edge", "nEdges"), ("Vertex", "nVertices")):
    for _p in ("x", "y", "z", "lon", "lat", "f"):
        REAL_VARS[_p + _el] = (_d,)
REAL_VARS.update({"areaCell": ("nCells",), "areaTriangle": ("nVertices",), "dcEdge": ("nEdges",),
                  "dvEdge": ("nEdges",), "angleEdge": ("nEdges",), "bottomDepth": ("nCells",)})


def mpas_arrays(g):
    """meshgen mesh -> MPAS file conventions: 1-based, 0 = none, padded with the last valid entry."""
    out = {}
    for n in INT_VARS:
        a = np.asarray(g[n]).astype(np.int32)
        if n.startswith("nEdges"):
            out[n] = a
            continue
        b = a + 1
        if n in ("cellsOnCell", "edgesOnCell", "verticesOnCell"):
            for c in range(a.shape[0]):
                k = g["nEdgesOnCell"][c]
                if n != "cellsOnCell":
                    b[c, k:] = b[c, k - 1]      # MPAS files repeat / pad beyond nEdgesOnCell
        out[n] = b
    for n in REAL_VARS:
        out[n] = np.asarray(g[n], dtype=np.float64)
    return out


def omega_name(n):
    return n[0].upper() + n[1:]


def write_scipy(path, g, version, omega_names=False, K=3):
    from scipy.io import netcdf_file
    rn = omega_name if omega_names else (lambda x: x)
    dims = {"nCells": g["nCells"], "nEdges": g["nEdges"], "nVertices": g["nVertices"], "maxEdges": g["maxEdges"],
            "maxEdges2": 2 * g["maxEdges"], "TWO": 2, "vertexDegree": 3, "nVertLevels": K}
    dn = lambda d: d if d in ("TWO", "maxEdges2", "nVertLevels") else rn(d)
    with netcdf_file(path, "w", version=version) as f:
        f.createDimension("Time", None)
        for d, n in dims.items():
            f.createDimension(dn(d), n)
        arr = mpas_arrays(g)
        for n, dd in INT_VARS.items():
            v = f.createVariable(rn(n) if not n.startswith("nEdges") else n, "i4", tuple(dn(d) for d in dd))
            v[:] = arr[n]
        for n, dd in REAL_VARS.items():
            v = f.createVariable(n, "f8", tuple(dn(d) for d in dd))
            v[:] = arr[n]
        rng = np.random.default_rng(1)
        h = f.createVariable("layerThickness", "f8", ("Time", dn("nCells"), "nVertLevels"))
        u = f.createVariable("normalVelocity", "f4", ("Time", dn("nEdges"), "nVertLevels"))
        hv = rng.random((2, g["nCells"], K))
        uv = rng.random((2, g["nEdges"], K)).astype(np.float32)
        h[0], h[1] = hv[0], hv[1]
        u[0], u[1] = uv[0], uv[1]
    return hv, uv


def write_cdf5(path, g):
    """Minimal CDF-5 writer (64-bit sizes everywhere; fixed-size variables only)."""
    arr = mpas_arrays(g)
    dims = [("nCells", g["nCells"]), ("nEdges", g["nEdges"]), ("nVertices", g["nVertices"]),
            ("maxEdges", g["maxEdges"]), ("maxEdges2", 2 * g["maxEdges"]), ("TWO", 2), ("vertexDegree", 3)]
    dimid = {n: i for i, (n, _) in enumerate(dims)}
    i8 = lambda v: struct.pack(">q", v)
    i4 = lambda v: struct.pack(">i", v)

    def name(s):
        b = s.encode()
        return i8(len(b)) + b + b"\0" * ((4 - len(b) % 4) % 4)
    variables = [(n, dd, 4, arr[n].astype(">i4").tobytes()) for n, dd in INT_VARS.items()] + \
                [(n, dd, 6, arr[n].astype(">f8").tobytes()) for n, dd in REAL_VARS.items()]

    def header(begins):
        hdr = b"CDF\x05" + i8(0) + i4(0x0A) + i8(len(dims))
        for n, ln in dims:
            hdr += name(n) + i8(ln)
        hdr += i4(0) + i8(0)                               # no global attributes
        hdr += i4(0x0B) + i8(len(variables))
        for (n, dd, t, data), b in zip(variables, begins):
            hdr += name(n) + i8(len(dd)) + b"".join(i8(dimid[d]) for d in dd)
            hdr += i4(0) + i8(0)                           # no attributes
            hdr += i4(t) + i8((len(data) + 3) // 4 * 4) + i8(b)
        return hdr
    hlen = len(header([0] * len(variables)))
    begins, off = [], hlen
    for _, _, _, data in variables:
        begins.append(off)
        off += (len(data) + 3) // 4 * 4
    with open(path, "wb") as f:
        f.write(header(begins))
        for _, _, _, data in variables:
            f.write(data + b"\0" * ((4 - len(data) % 4) % 4))


@pytest.fixture(scope="module")
def meshes():
    return {"hex": planar_hex(8, 6, 1000.0), "ico": spherical_voronoi(points=icosahedral_points(2), lloyd=1)}


def check_same(read, g):
    for n in ("nCells", "nEdges", "nVertices", "maxEdges", "vertexDegree"):
        assert read[n] == g[n]
    for n in oa.GlobalMeshC._I:
        assert np.array_equal(read[n], np.asarray(g[n]).astype(np.int32)), n
    for n in oa.GlobalMeshC._R:
        assert np.array_equal(read[n], np.asarray(g[n], dtype=np.float64)), n


@pytest.mark.parametrize("fmt", ["cdf1", "cdf2", "cdf2-omega-names", "cdf5"])
@pytest.mark.parametrize("which", ["hex", "ico"])
def test_round_trip(tmp_path, meshes, which, fmt):
    g = meshes[which]
    path = str(tmp_path / f"{which}_{fmt}.nc")
    if fmt == "cdf5":
        write_cdf5(path, g)
    else:
        hv, uv = write_scipy(path, g, 1 if fmt == "cdf1" else 2, omega_names=fmt.endswith("names"))
    mf = oa.MeshFile(path)
    check_same(mf.arrays(), g)
    if fmt != "cdf5":   # record variables of an initial state: double and float, per record and all records
        assert mf.dim("Time") == 2 and mf.dim("nVertLevels") == 3 and mf.dim("nope") == -1
        assert np.array_equal(mf.read("layerThickness", 1), hv[1].ravel())
        assert np.array_equal(mf.read("normalVelocity", 0), uv[0].astype(np.float64).ravel())
        assert np.array_equal(mf.read("layerThickness"), hv.ravel())
        with pytest.raises(KeyError):
            mf.read("temperature")


def test_decomp_from_file_equals_decomp_from_memory(tmp_path, meshes):
    g = meshes["ico"]
    path = str(tmp_path / "ico.nc")
    write_scipy(path, g, 2)
    mf = oa.MeshFile(path)
    for r in range(3):
        d1, d2 = oa.Decomp(oa.GlobalMesh(g), 3, r, 3), oa.Decomp(mf.gm, 3, r, 3)
        for n in ("CellID", "EdgeID", "VertexID", "CellLoc", "EdgeLoc", "NCellsHalo"):
            assert np.array_equal(d1.get_array(n), d2.get_array(n)), n
        m1, m2 = oa.HorzMesh(d1, 4, host_only=True), oa.HorzMesh(d2, 4, host_only=True)
        for n in ("EdgesOnCell", "CellsOnEdge", "EdgesOnEdge", "NEdgesOnCell", "AreaCell", "WeightsOnEdge",
                  "EdgeSignOnCell", "FVertex", "XEdge"):
            assert np.array_equal(m1.get_array(n), m2.get_array(n)), n


def test_partition_file(tmp_path, meshes):
    g = meshes["hex"]
    part = (np.arange(g["nCells"]) * 3 // g["nCells"]).astype(np.int32)
    p = tmp_path / "graph.info.part.3"
    p.write_text("\n".join(str(int(t)) for t in part) + "\n")
    ct = oa.read_partition_file(str(p))
    d = oa.Decomp(oa.GlobalMesh(g), 3, 1, 2, cell_task=ct)
    own = d.get_array("CellID")[: d.get_int("NCellsOwned")] - 1
    assert np.array_equal(np.sort(own), np.nonzero(part == 1)[0])


def test_errors_are_loud(tmp_path):
    p = tmp_path / "junk.nc"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(oa.OmegaAmdError, match="NetCDF-4/HDF5"):
        oa.MeshFile(str(p))
    with pytest.raises(oa.OmegaAmdError, match="cannot open"):
        oa.MeshFile(str(tmp_path / "missing.nc"))
