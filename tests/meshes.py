"""Named test meshes (generated once per process).

    hexNXxNY[dDC]             planar periodic hexagons (dc in km, default 30)
    icoN                      icosahedral Voronoi sphere, level N (12 pentagons)
    fibN                      relaxed Fibonacci sphere with N cells (pentagons, hexagons, heptagons)
    <base>_pad8               the same mesh stored with maxEdges = 8
    <base>_permN              N % of the cells with two slots of their per-cell lists swapped (not in ring order)
    <base>_coast_<kind>[_raw][_compact]
                              culled with omega_amd.meshgen.coast_mask(kind): island | channel | strait | lakes |
                              ragged | mixed; `raw` keeps [missing, cell] on boundary edges whose first cell was
                              removed (default: the culler's convention, surviving cell first); `compact` moves the
                              surviving EdgesOnEdge entries up instead of leaving holes in place
"""
from omega_amd.meshgen import (planar_hex, spherical_voronoi, icosahedral_points, pad_max_edges, cull, coast_mask,
                               permute_cell_slots)

_CACHE = {}

COAST_KINDS = ("island", "channel", "strait", "lakes", "ragged", "mixed")


def named_mesh(name: str) -> dict:
    if name in _CACHE:
        return _CACHE[name]
    if "_coast_" in name:
        base, spec = name.split("_coast_")
        parts = spec.split("_")
        g0 = named_mesh(base)
        g = cull(g0, coast_mask(g0, parts[0]), first_cell_valid="raw" not in parts[1:],
                 compact_edges_on_edge="compact" in parts[1:])
    elif "_perm" in name and name.rsplit("_perm", 1)[1].isdigit():
        base, pct = name.rsplit("_perm", 1)
        g = permute_cell_slots(named_mesh(base), int(pct) / 100.0)
    elif name.endswith("_pad8"):
        g = pad_max_edges(named_mesh(name[:-5]), 8)
    elif name.startswith("hex"):
        dims, _, dc = name[3:].partition("d")
        nx, _, ny = dims.partition("x")
        g = planar_hex(int(nx), int(ny or nx), float(dc or 30) * 1.0e3)
    elif name.startswith("ico"):
        g = spherical_voronoi(points=icosahedral_points(int(name[3:])), lloyd=2)
    elif name.startswith("fib"):
        g = spherical_voronoi(int(name[3:]), lloyd=4)
    else:
        raise ValueError(f"unknown mesh name {name!r}")
    _CACHE[name] = g
    return g
