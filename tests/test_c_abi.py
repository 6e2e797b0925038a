"""The C-ABI library loads (no GPU needed) and exports every symbol include/omega_amd.h
declares; host-only entry points work without a device; device entry points fail loudly
(non-zero + message), never fall back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import planar_hex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "omega_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(omg_[a-z0-9_]+)\s*\(", src))
    names.discard("omg_transport_fn")
    return sorted(names)


def test_every_declared_symbol_is_exported():
    L = oa.lib()
    syms = declared_symbols()
    assert len(syms) >= 70
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, f"declared in include/omega_amd.h but not exported: {missing}"


def test_oracle_is_not_linked_into_the_product():
    """The product library must not depend on the oracle (test infrastructure)."""
    import subprocess
    out = subprocess.run(["ldd", oa.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    nm = subprocess.run(["nm", "-D", oa.LIB_PATH], capture_output=True, text=True).stdout
    assert " orc_" not in nm


def test_host_only_objects_and_loud_device_failures():
    g = planar_hex(8, 8, 1.0)
    gm = oa.GlobalMesh(g)
    d = oa.Decomp(gm, 1, 0, 3)
    m = oa.HorzMesh(d, 4, host_only=True)
    assert m.NCellsAll == 64 and m.NEdgesAll == 192 and m.NVerticesAll == 128
    em = m.get_array("EdgeMask")
    assert em.shape == (193, 4) and np.all(em == 1.0)
    # compute objects need device arrays: must fail with a message, not silently run on the CPU
    with pytest.raises(oa.OmegaAmdError, match="host-only"):
        oa.OceanState(m, None, 4, 2)
    with pytest.raises(oa.OmegaAmdError, match="host-only"):
        oa.Tendencies(m, 4, 1)
    if oa.device_count() == 0:
        with pytest.raises(oa.OmegaAmdError):
            oa.device_init(0)
        with pytest.raises(oa.OmegaAmdError):
            oa.HorzMesh(d, 4)  # device mirrors cannot be allocated without a GPU


def test_mesh_tables_are_kept_at_the_largest_valence_present():
    """A mesh file's maxEdges dimension is often larger than any cell's valence (HorzMesh.cpp: compactMaxEdges):
    HorzMesh keeps its cell-slot tables at the largest valence present, Decomp keeps the file's width."""
    from omega_amd.meshgen import pad_max_edges
    g = pad_max_edges(planar_hex(8, 8, 1.0), 8)
    gm = oa.GlobalMesh(g)
    d = oa.Decomp(gm, 1, 0, 3)
    m = oa.HorzMesh(d, 4, host_only=True)
    assert d.get_int("MaxEdges") == 8 and m.get_int("MaxEdgesFile") == 8 and m.get_int("MaxEdges") == 6
    eoc, eoe = m.get_array("EdgesOnCell"), m.get_array("EdgesOnEdge")
    assert eoc.shape == (65, 6) and eoe.shape[1] == 12
    m0 = oa.HorzMesh(oa.Decomp(oa.GlobalMesh(planar_hex(8, 8, 1.0)), 1, 0, 3), 4, host_only=True)
    assert np.array_equal(eoc, m0.get_array("EdgesOnCell")) and np.array_equal(eoe, m0.get_array("EdgesOnEdge"))
    assert np.array_equal(m.get_array("WeightsOnEdge"), m0.get_array("WeightsOnEdge"))


def test_bad_arguments_return_errors():
    g = planar_hex(8, 8, 1.0)
    gm = oa.GlobalMesh(g)
    with pytest.raises(oa.OmegaAmdError):
        oa.Decomp(gm, 2, 5, 3)          # task out of range
    with pytest.raises(oa.OmegaAmdError):
        oa.Decomp(gm, 1, 0, 0)          # halo width must be >= 1
    d = oa.Decomp(gm, 1, 0, 3)
    with pytest.raises(oa.OmegaAmdError, match="no integer member"):
        d.get_int("NoSuchThing")
    assert oa.coeff_seconds(1. / 3, 600.0) == 200.0  # TimeMgr integer-fraction arithmetic


def test_time_stepper_coefficients_match_oracle():
    from oracle import oracle as O
    for mult in (1. / 6, 1. / 3, 0.5, 1.0):
        for dt in (600.0, 0.2, 0.1, 37.5, 1800.0):
            assert oa.coeff_seconds(mult, dt) == O.coeff_seconds(mult, dt)


def test_library_never_reads_the_environment():
    """No OMEGA_* name is compiled into the product library (default build): kernel structure and tile geometry change
    through omg_set_option only (omega_amd/csrc/Tuning.h)."""
    import subprocess
    import omega_amd as oa
    out = subprocess.run(["strings", oa.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert [l for l in out.splitlines() if "OMEGA_" in l] == []
    assert "getenv" not in subprocess.run(["nm", "-D", "--undefined-only", oa.LIB_PATH], stdout=subprocess.PIPE,
                                          check=True).stdout.decode()


def test_options_api():
    import omega_amd as oa
    import pytest
    # every option the library has (omega_amd/csrc/Tuning.h names, next to each, the mesh class that reaches the
    # structure it forces without the switch); the round-1..5 A/B knobs and the ProbeSlice measurement probe are gone
    survivors = (("MergeL1", 1), ("Pair", 1), ("TracerPatch", 1), ("SendBand", 1), ("BandOnComm", 1), ("ShrinkSweeps", 1),
                 ("ForceGeneric", 0), ("KeepMaxEdges", 0), ("NarrowTables", 1), ("Graphs", -1))
    for gone in ("ProbeSlice", "ProbeBlocks", "EdgeMode", "FuseFinal", "FuseL3", "InlineOther", "FoldLists", "Alternate",
                 "WaveWindow", "ValenceSort", "DomValence", "W", "TX", "TY", "Sweeps", "ChunkSplit", "TailSplit"):
        with pytest.raises(oa.OmegaAmdError):
            oa.set_option(gone, 1)
    for name, default in survivors:
        if not __import__("os").environ.get("OMEGA_AMD_OPTIONS"):
            assert oa.get_option(name) == default, name
        old = oa.get_option(name)
        oa.set_option(name, 7)
        assert oa.get_option(name) == 7
        oa.set_option(name, old)
    with pytest.raises(oa.OmegaAmdError):
        oa.set_option("NoSuchOption", 1)


def test_the_fused_rhs_limit_is_a_decision_not_a_surprise():
    """The fused kernels address an array plane with 32-bit byte offsets: a rank with an array of 4 GiB or more (about
    2.2 M cells x 80 levels) is outside them.  omg_tend_create then FAILS naming the limit (omg_tend_create_reference_structured
    is the caller's explicit acceptance of the 23-launch path) -- checked here on sizes alone, no allocation."""
    import omega_amd as oa
    K = 80
    rows_max = 0xffffff00 // (oa.level_pitch(K) * 8)              # rows of 640 bytes below 4 GiB
    ok, why = oa.fused_limit(rows_max // 3, rows_max, 2 * rows_max // 3, 6, K)
    assert ok and why == ""
    ok, why = oa.fused_limit(rows_max // 3 + 1, rows_max + 1, 2 * rows_max // 3, 6, K)      # one edge row more
    assert not ok and "4 GiB" in why and str(rows_max) in why and "more ranks" in why, why
    ok, why = oa.fused_limit(2_300_000, 6_900_000, 4_600_000, 6, K)                         # "2.3 M cells on one GPU"
    assert not ok and "6900000 rows" in why
    ok, _ = oa.fused_limit(2_300_000, 6_900_000, 4_600_000, 6, 60)                          # the same mesh at 60 (-> 64) levels fits
    assert ok
    ok, why = oa.fused_limit(1000, 3000, 2000, 9, K)
    assert not ok and "MaxEdges = 9" in why
    ok, why = oa.fused_limit(1000, 3000, 2000, 4, K)
    assert not ok and "MaxEdges = 4" in why
    # configs[4] per rank (462 400 owned cells + 4 halo layers, 80 levels) is far inside
    assert oa.fused_limit(480_000, 1_440_000, 960_000, 6, 80)[0]

