"""omega_amd -- Python binding of libomega_amd.so (include/omega_amd.h).

Thin ctypes plumbing used by tests/, bench.py and __graft_entry__.py: the product is the
C++/HIP library under omega_amd/csrc (classes named after Omega's own: Decomp, Halo,
HorzMesh, OceanState, Tracers, AuxiliaryState, Tendencies, TimeStepper).  There is no
Python or CPU implementation of the hot path here: if the shared library is missing,
importing the binding raises, and without a HIP device every device call fails.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OMEGA_AMD_LIB", os.path.join(_HERE, "lib", "libomega_amd.so"))

ON_CELL, ON_EDGE, ON_VERTEX = 0, 1, 2

PD = C.POINTER(C.c_double)
PI = C.POINTER(C.c_int32)


class OmegaAmdError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile libomega_amd.so for gfx950 with hipcc (omega_amd/csrc/Makefile)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8", "-s"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


class GlobalMeshC(C.Structure):
    _I = ("cellsOnCell", "edgesOnCell", "verticesOnCell", "cellsOnEdge", "verticesOnEdge", "edgesOnEdge",
          "cellsOnVertex", "edgesOnVertex")
    _R = ("xCell", "yCell", "zCell", "lonCell", "latCell", "xEdge", "yEdge", "zEdge", "lonEdge", "latEdge",
          "xVertex", "yVertex", "zVertex", "lonVertex", "latVertex", "areaCell", "areaTriangle",
          "kiteAreasOnVertex", "dcEdge", "dvEdge", "angleEdge", "weightsOnEdge", "fCell", "fEdge", "fVertex",
          "bottomDepth")
    _fields_ = ([(n, C.c_int32) for n in ("nCells", "nEdges", "nVertices", "maxEdges", "vertexDegree")]
                + [(n, PI) for n in _I] + [(n, PD) for n in _R])


CONFIG_FLAGS = ("ThicknessFluxTendencyEnable", "PVTendencyEnable", "KETendencyEnable", "SSHTendencyEnable",
                "VelDiffTendencyEnable", "VelHyperDiffTendencyEnable", "WindForcingTendencyEnable",
                "BottomDragTendencyEnable", "TracerHorzAdvTendencyEnable", "TracerDiffTendencyEnable",
                "TracerHyperDiffTendencyEnable", "FluxThicknessUpwind", "FluxTracerUpwind", "WindInterpIsotropic")
CONFIG_REALS = ("ViscDel2", "ViscDel4", "DivFactor", "EddyDiff2", "EddyDiff4", "Density0", "BottomDragCoeff")


class TendConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in CONFIG_FLAGS] + [(n, C.c_double) for n in CONFIG_REALS]


TRANSPORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, PI, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                           C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p)

CUSTOM_TEND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_double, C.c_void_p)

_lib = None


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OmegaAmdError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950). omega_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.omg_last_error.restype = C.c_char_p
        _lib = L
        # test / measurement plumbing (this file, not the library): OMEGA_AMD_OPTIONS="MergeL1=0,Pair=0" is turned into
        # omg_set_option calls, so that a whole pytest or bench.py run can be repeated under another kernel structure
        for item in filter(None, os.environ.get("OMEGA_AMD_OPTIONS", "").split(",")):
            name, _, val = item.partition("=")
            set_option(name.strip(), int(val))
    return _lib


def set_option(name: str, value: int):
    """omg_set_option (omega_amd/csrc/Tuning.h): measurement / test switches; the library never reads the environment."""
    _chk(lib().omg_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_int()
    _chk(lib().omg_get_option(name.encode(), C.byref(v)))
    return v.value


def _chk(rc):
    if rc != 0:
        raise OmegaAmdError(lib().omg_last_error().decode())


def _pd(a):
    if a is None:
        return PD()
    assert a.dtype == np.float64 and a.flags.c_contiguous, "need C-contiguous float64"
    return a.ctypes.data_as(PD)


def _pi(a):
    if a is None:
        return PI()
    assert a.dtype == np.int32 and a.flags.c_contiguous, "need C-contiguous int32"
    return a.ctypes.data_as(PI)


def device_count() -> int:
    n = C.c_int(0)
    _chk(lib().omg_device_count(C.byref(n)))
    return n.value


def device_init(dev: int = 0):
    _chk(lib().omg_device_init(dev))


def device_synchronize():
    _chk(lib().omg_device_synchronize())


class Stream:
    def __init__(self, handle=None):
        self.own = handle is None
        if handle is None:
            h = C.c_void_p()
            _chk(lib().omg_stream_create(C.byref(h)))
            handle = h.value
        self.h = C.c_void_p(handle)

    def synchronize(self):
        _chk(lib().omg_stream_synchronize(self.h))

    def __del__(self):
        try:
            if self.own and self.h:
                lib().omg_stream_destroy(self.h)
        except Exception:
            pass


class Event:
    def __init__(self):
        h = C.c_void_p()
        _chk(lib().omg_event_create(C.byref(h)))
        self.h = h

    def record(self, stream: "Stream | None"):
        _chk(lib().omg_event_record(self.h, stream.h if stream else None))

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float()
        _chk(lib().omg_event_elapsed_ms(self.h, stop.h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib().omg_event_destroy(self.h)
        except Exception:
            pass


def _sh(s):
    return s.h if s is not None else None


class GlobalMesh:
    """Keeps the numpy arrays of a meshgen mesh alive behind an omg_global_mesh."""

    def __init__(self, g: dict):
        self.g = g
        self.keep = {}
        s = GlobalMeshC()
        s.nCells, s.nEdges, s.nVertices = g["nCells"], g["nEdges"], g["nVertices"]
        s.maxEdges, s.vertexDegree = g["maxEdges"], g["vertexDegree"]
        for n in GlobalMeshC._I:
            a = np.ascontiguousarray(g[n], dtype=np.int32)
            self.keep[n] = a
            setattr(s, n, _pi(a))
        for n in GlobalMeshC._R:
            a = np.ascontiguousarray(g[n], dtype=np.float64)
            self.keep[n] = a
            setattr(s, n, _pd(a))
        self.s = s


def write_restart(path, decomp, state, tracers, K, NT, simulation_time, steps_done, rank=0, barrier=None):
    """Restart dump: rank 0 creates the file, every rank writes the rows of its OWNED cells / edges (time
    level 0) at their global positions.  `barrier` (callable) separates the two phases when there are
    several ranks."""
    if rank == 0:
        _chk(lib().omg_restart_create(os.fsencode(path), C.c_int64(decomp.get_int("NCellsGlobal")),
                                      C.c_int64(decomp.get_int("NEdgesGlobal")), K, NT, C.c_double(simulation_time),
                                      C.c_int64(steps_done)))
    if barrier is not None:
        barrier()
    f = C.c_void_p()
    _chk(lib().omg_restart_open(os.fsencode(path), 1, C.byref(f)))
    try:
        nc, ne = decomp.get_int("NCellsOwned"), decomp.get_int("NEdgesOwned")
        cid = np.ascontiguousarray(decomp.get_array("CellID")[:nc], dtype=np.int32)
        eid = np.ascontiguousarray(decomp.get_array("EdgeID")[:ne], dtype=np.int32)
        h, u = state.copy_to_host(0)
        _chk(lib().omg_restart_write_rows(f, b"layerThickness", 0, _pi(cid), C.c_int64(nc), _pd(np.ascontiguousarray(h[:nc]))))
        _chk(lib().omg_restart_write_rows(f, b"normalVelocity", 0, _pi(eid), C.c_int64(ne), _pd(np.ascontiguousarray(u[:ne]))))
        if NT > 0:
            tr = tracers.copy_to_host(0)
            for l in range(NT):
                _chk(lib().omg_restart_write_rows(f, b"tracers", l, _pi(cid), C.c_int64(nc), _pd(np.ascontiguousarray(tr[l, :nc]))))
    finally:
        lib().omg_restart_close(f)
    if barrier is not None:
        barrier()


def write_history(path, decomp, state, tracers, aux, contents="State,Tracers,AuxiliaryState", simulation_time=0.0,
                  time_level=0, create_file=True, stream=None) -> int:
    """One history dump (omg_history_write): field names / groups as in the reference's History stream Contents."""
    n = C.c_int()
    _chk(lib().omg_history_write(os.fsencode(path), decomp.h, state.h, tracers.h if tracers is not None else None, aux.h,
                                 contents.encode(), C.c_double(simulation_time), time_level, int(create_file), _sh(stream),
                                 C.byref(n)))
    return n.value


def read_restart(path, decomp, mesh, state, tracers, K, NT):
    """Restart load: every rank reads the rows of ALL its local cells / edges (owned and halo, by global
    id) into time level 0, so no halo exchange is needed afterwards.  Returns (simulation_time, steps_done)."""
    f = C.c_void_p()
    _chk(lib().omg_restart_open(os.fsencode(path), 0, C.byref(f)))
    try:
        info = [C.c_int64(), C.c_int64(), C.c_int(), C.c_int(), C.c_double(), C.c_int64()]
        _chk(lib().omg_restart_info(f, *[C.byref(x) for x in info]))
        if (info[0].value, info[1].value, info[2].value) != (decomp.get_int("NCellsGlobal"), decomp.get_int("NEdgesGlobal"), K) \
                or info[3].value < max(NT, 1):
            raise OmegaAmdError(f"{path}: restart file does not match this mesh / configuration")
        nc, ne = mesh.NCellsAll, mesh.NEdgesAll
        cid = np.ascontiguousarray(decomp.get_array("CellID")[:nc], dtype=np.int32)
        eid = np.ascontiguousarray(decomp.get_array("EdgeID")[:ne], dtype=np.int32)
        h = np.zeros((mesh.NCellsSize, K))
        u = np.zeros((mesh.NEdgesSize, K))
        rows = np.empty((nc, K))
        _chk(lib().omg_restart_read_rows(f, b"layerThickness", 0, _pi(cid), C.c_int64(nc), _pd(rows)))
        h[:nc] = rows
        rows = np.empty((ne, K))
        _chk(lib().omg_restart_read_rows(f, b"normalVelocity", 0, _pi(eid), C.c_int64(ne), _pd(rows)))
        u[:ne] = rows
        state.copy_to_device(h, u, 0)
        if NT > 0:
            tr = np.zeros((NT, mesh.NCellsSize, K))
            rows = np.empty((nc, K))
            for l in range(NT):
                _chk(lib().omg_restart_read_rows(f, b"tracers", l, _pi(cid), C.c_int64(nc), _pd(rows)))
                tr[l, :nc] = rows
            tracers.copy_to_device(tr, 0)
        return info[4].value, info[5].value
    finally:
        lib().omg_restart_close(f)


class MeshFile:
    """An MPAS mesh / initial-state file (NetCDF classic CDF-1/2/5) opened by the library's own reader;
    `.gm` is the GlobalMesh to build a Decomp from (the arrays live inside the file handle)."""

    def __init__(self, path: str, mesh: bool = True):
        h = C.c_void_p()
        _chk(lib().omg_mesh_file_open(os.fsencode(path), C.byref(h)))
        self.h = h
        self.gm = None
        if not mesh:   # a state-only file (initial conditions, forcing): variables through read()
            return
        gm = GlobalMesh.__new__(GlobalMesh)
        gm.s = GlobalMeshC()
        gm.keep = {"file": self}
        _chk(lib().omg_mesh_file_global_mesh(self.h, C.byref(gm.s)))
        self.gm = gm

    def dim(self, name: str) -> int:
        v = C.c_int64()
        _chk(lib().omg_mesh_file_dim(self.h, name.encode(), C.byref(v)))
        return v.value

    def read(self, name: str, record: int = -1) -> np.ndarray:
        n = C.c_int64()
        _chk(lib().omg_mesh_file_var_size(self.h, name.encode(), C.c_int64(record), C.byref(n)))
        if n.value < 0:
            raise KeyError(name)
        out = np.empty(n.value, dtype=np.float64)
        _chk(lib().omg_mesh_file_read_f64(self.h, name.encode(), C.c_int64(record), _pd(out), C.c_size_t(n.value)))
        return out

    def arrays(self) -> dict:
        """The global mesh as numpy copies, keyed like a meshgen mesh (test use)."""
        s = self.gm.s
        nC, nE, nV, mE, vD = s.nCells, s.nEdges, s.nVertices, s.maxEdges, s.vertexDegree
        shp = {"cellsOnCell": (nC, mE), "edgesOnCell": (nC, mE), "verticesOnCell": (nC, mE), "cellsOnEdge": (nE, 2),
               "verticesOnEdge": (nE, 2), "edgesOnEdge": (nE, 2 * mE), "cellsOnVertex": (nV, vD),
               "edgesOnVertex": (nV, vD), "kiteAreasOnVertex": (nV, vD), "weightsOnEdge": (nE, 2 * mE)}
        out = {"nCells": nC, "nEdges": nE, "nVertices": nV, "maxEdges": mE, "vertexDegree": vD}
        for n in GlobalMeshC._I + GlobalMeshC._R:
            p = getattr(s, n)
            el = n[-4:] == "Cell" and nC or n[-4:] == "Edge" and nE or nV
            if n == "areaTriangle":
                el = nV
            if n == "bottomDepth":
                el = nC
            shape = shp.get(n, (el,))
            out[n] = np.ctypeslib.as_array(p, shape=(int(np.prod(shape)),)).reshape(shape).copy()
        return out

    def __del__(self):
        try:
            lib().omg_mesh_file_close(self.h)
        except Exception:
            pass


class DeviceBuffer:
    """A device copy of a host array (omg_device_malloc / omg_copy_to_device)."""

    def __init__(self, host: np.ndarray):
        self.host = np.ascontiguousarray(host)
        p = C.c_void_p()
        _chk(lib().omg_device_malloc(C.c_size_t(self.host.nbytes), C.byref(p)))
        self.ptr = p.value
        _chk(lib().omg_copy_to_device(C.c_void_p(self.ptr), self.host.ctypes.data_as(C.c_void_p), C.c_size_t(self.host.nbytes)))

    def to_host(self) -> np.ndarray:
        out = np.empty_like(self.host)
        _chk(lib().omg_copy_to_host(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes)))
        return out

    def __del__(self):
        try:
            lib().omg_device_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def copy_to_device(dev_ptr: int, host: np.ndarray):
    """omg_copy_to_device: a contiguous host array to raw device memory (synchronous)"""
    a = np.ascontiguousarray(host)
    _chk(lib().omg_copy_to_device(C.c_void_p(dev_ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)))


def combine_dd(pairs) -> tuple:
    """ddSum (Reductions.h:24-35) over an [n][2] array of (hi, lo) partial sums, in order."""
    p = np.ascontiguousarray(pairs, dtype=np.float64).reshape(-1, 2)
    out = (C.c_double * 2)()
    _chk(lib().omg_combine_dd(_pd(p), p.shape[0], out))
    return out[0], out[1]


def local_sum_dd(a_ptr: int, n: int, b_ptr: int = 0, stream=None) -> tuple:
    """Double-double sum of n device doubles at a_ptr (times those at b_ptr if given)."""
    out = (C.c_double * 2)()
    _chk(lib().omg_local_sum_dd(C.c_void_p(a_ptr), C.c_void_p(b_ptr) if b_ptr else None, C.c_size_t(n), _sh(stream), out))
    return out[0], out[1]


def level_pitch(k: int) -> int:
    """Row pitch (in values) of the library's own level-indexed device arrays (omg_level_pitch)."""
    return lib().omg_level_pitch(k)


def local_weighted_sum_dd(w_ptr: int, a_ptr: int, nrows: int, k: int, b_ptr: int = 0, stream=None, row_pitch: int = 0) -> tuple:
    """row_pitch: pitch of the [rows][k] arrays in values (level_pitch(k) for arrays owned by the library, 0 = compact)"""
    out = (C.c_double * 2)()
    _chk(lib().omg_local_weighted_sum_dd(C.c_void_p(w_ptr), C.c_void_p(a_ptr), C.c_void_p(b_ptr) if b_ptr else None,
                                         nrows, k, row_pitch, _sh(stream), out))
    return out[0], out[1]


def device_resource_count() -> int:
    """omg_device_resource_count: device buffers, streams and events the library has created so far"""
    n = C.c_int64()
    _chk(lib().omg_device_resource_count(C.byref(n)))
    return n.value


def global_sum_dd(local_hi_lo, group=None, halo=None, stream=None) -> float:
    """globalSum (Reductions.h:71-84): all-gather the ranks' (hi, lo) partial sums and combine them with the
    ddSum operator in rank order -- the same value on every rank and for every partition.  With `halo` the gather
    runs inside the library over that Halo's wire (omg_halo_global_sum_dd: RCCL or peer wire); without, over
    torch.distributed (test rigs on the host-staged transport)."""
    if halo is not None:
        return halo.global_sum_dd([local_hi_lo], stream=stream)[0]
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return combine_dd([local_hi_lo])[0]
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor(list(local_hi_lo), dtype=torch.float64, device=dev)
    allp = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(allp, mine, group=group)
    return combine_dd(torch.stack(allp).cpu().numpy())[0]


def read_partition_file(path: str) -> np.ndarray:
    """A METIS partition file (`graph.info.part.N`: one owner task per line, cell order) as the
    cell_task vector of Decomp -- the reference calls METIS itself (Decomp.cpp:868-1000); production
    runs with pre-computed partitions pass them here."""
    return np.loadtxt(path, dtype=np.int32, ndmin=1)


def partition_cells(gm: GlobalMesh, nparts: int, method: str = "graph"):
    """(cell_task[nCells], edge_cut) of the built-in partitioners: "rcb" or "graph" (omg_partition_cells)."""
    out = np.zeros(gm.s.nCells, dtype=np.int32)
    cut = C.c_int64()
    _chk(lib().omg_partition_cells(C.byref(gm.s), nparts, method.encode(), _pi(out), C.byref(cut)))
    return out, cut.value


class Decomp:
    def __init__(self, gm: GlobalMesh, nparts: int = 1, mytask: int = 0, halo_width: int = 3, cell_task=None,
                 local_order: str = "global"):
        """local_order: "global" (the reference's numbering by global id), "curve" (Morton curve through the cell
        centres: spatially compact local numbering whatever the file's order), "hilbert" (Hilbert curve) or "kd" (k-d
        order: compact tiles on the surface -- what spheres want)."""
        self.gm = gm
        h = C.c_void_p()
        ct = None if cell_task is None else np.ascontiguousarray(cell_task, dtype=np.int32)
        _chk(lib().omg_decomp_create_ordered(C.byref(gm.s), nparts, mytask, halo_width, _pi(ct),
                                             {"global": 0, "curve": 1, "hilbert": 2, "kd": 3}[local_order], C.byref(h)))
        self.h = h

    def get_int(self, name: str) -> int:
        v = C.c_int32()
        _chk(lib().omg_decomp_get_int(self.h, name.encode(), C.byref(v)))
        return v.value

    def get_array(self, name: str) -> np.ndarray:
        hw = self.get_int("HaloWidth")
        shapes = {"CellID": (self.get_int("NCellsSize"),), "EdgeID": (self.get_int("NEdgesSize"),),
                  "VertexID": (self.get_int("NVerticesSize"),), "CellLoc": (self.get_int("NCellsSize"), 2),
                  "EdgeLoc": (self.get_int("NEdgesSize"), 2), "VertexLoc": (self.get_int("NVerticesSize"), 2),
                  "NCellsHalo": (hw,), "NEdgesHalo": (hw,), "NVerticesHalo": (hw,),
                  "CellTask": (self.get_int("NCellsGlobal"),)}
        out = np.zeros(shapes[name], dtype=np.int32)
        _chk(lib().omg_decomp_get_array(self.h, name.encode(), _pi(out), C.c_size_t(out.size)))
        return out

    def __del__(self):
        try:
            lib().omg_decomp_destroy(self.h)
        except Exception:
            pass


class RcclComm:
    """RCCL communicator owned by the library (omega_amd/csrc/Rccl.cpp).  `unique_id()` on rank 0, distribute the
    128 bytes by any side channel, then every rank constructs RcclComm(id, nranks, rank) after device_init."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(RcclComm.ID_BYTES)
        _chk(lib().omg_rccl_get_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id: bytes, nranks: int, rank: int):
        assert len(unique_id) == RcclComm.ID_BYTES
        h = C.c_void_p()
        _chk(lib().omg_rccl_create(C.create_string_buffer(unique_id, RcclComm.ID_BYTES), nranks, rank, C.byref(h)))
        self.h = h

    def info(self) -> dict:
        n, r, v, e = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
        _chk(lib().omg_rccl_info(self.h, C.byref(n), C.byref(r), C.byref(v), C.byref(e)))
        return {"nranks": n.value, "rank": r.value, "version": v.value, "exchanges": e.value}

    def abort(self):
        _chk(lib().omg_rccl_abort(self.h))

    def exchange(self, peers, send_ptrs, send_bytes, recv_ptrs, recv_bytes, stream=None):
        n = len(peers)
        _chk(lib().omg_rccl_exchange(self.h, n, (C.c_int * n)(*peers), (C.c_void_p * n)(*send_ptrs),
                                     (C.c_size_t * n)(*send_bytes), (C.c_void_p * n)(*recv_ptrs),
                                     (C.c_size_t * n)(*recv_bytes), _sh(stream)))

    def close(self):
        """omg_rccl_destroy (ncclCommDestroy): explicitly, while the peers are still there -- not left to a destructor at
        interpreter exit"""
        if self.h:
            lib().omg_rccl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PeerWire:
    """Direct peer-copy halo wire owned by the library (omega_amd/csrc/PeerWire.cpp): mailbox + flags exported with
    HIP IPC, exchanges fully stream-ordered.  Create after device_init, all_gather `handle()` over any side channel,
    `connect(list_of_handles_in_rank_order)`, then `Halo.use_peer(wire)`."""

    HANDLE_BYTES = 160

    def __init__(self, nranks: int, rank: int, mailbox_bytes: int):
        h = C.c_void_p()
        _chk(lib().omg_peer_create(nranks, rank, C.c_size_t(mailbox_bytes), C.byref(h)))
        self.h = h
        self.nranks = nranks

    def handle(self) -> bytes:
        buf = C.create_string_buffer(PeerWire.HANDLE_BYTES)
        _chk(lib().omg_peer_local_handle(self.h, buf))
        return buf.raw

    def connect(self, handles):
        blob = b"".join(handles)
        assert len(blob) == self.nranks * PeerWire.HANDLE_BYTES
        _chk(lib().omg_peer_connect(self.h, C.create_string_buffer(blob, len(blob))))

    def info(self) -> dict:
        e, s = C.c_int64(), C.c_int()
        _chk(lib().omg_peer_info(self.h, C.byref(e), C.byref(s)))
        return {"exchanges": e.value, "status": s.value}

    def set_timeout(self, seconds: float):
        _chk(lib().omg_peer_set_timeout(self.h, C.c_double(seconds)))

    def close(self):
        if self.h:
            lib().omg_peer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Halo:
    def __init__(self, decomp: Decomp):
        self.decomp = decomp
        h = C.c_void_p()
        _chk(lib().omg_halo_create(decomp.h, C.byref(h)))
        self.h = h
        self._cb = None
        self._bufs = None

    @property
    def neighbors(self):
        n = C.c_int()
        _chk(lib().omg_halo_num_neighbors(self.h, C.byref(n)))
        out = []
        for i in range(n.value):
            t = C.c_int()
            _chk(lib().omg_halo_neighbor_task(self.h, i, C.byref(t)))
            out.append(t.value)
        return out

    def get_list(self, i: int, elem: int, recv: bool) -> np.ndarray:
        n = C.c_int()
        _chk(lib().omg_halo_list_size(self.h, i, elem, int(recv), C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        _chk(lib().omg_halo_get_list(self.h, i, elem, int(recv), _pi(out)))
        return out[: n.value]

    def required_bytes(self, i: int, per_cell: int, per_edge: int, per_vertex: int = 0) -> int:
        b = C.c_size_t()
        _chk(lib().omg_halo_required_bytes(self.h, i, C.c_size_t(per_cell), C.c_size_t(per_edge), C.c_size_t(per_vertex), C.byref(b)))
        return b.value

    def use_rccl(self, comm: "RcclComm"):
        """Route the exchanges through RCCL send / recv issued inside the library (production wire)."""
        _chk(lib().omg_halo_use_rccl(self.h, comm.h))
        self._comm = comm

    def use_peer(self, wire: "PeerWire"):
        """Route the exchanges through direct peer copies into the neighbours' mailboxes (stream-ordered, no host waits)."""
        _chk(lib().omg_halo_use_peer(self.h, wire.h))
        self._wire = wire

    def global_sum_dd(self, local_pairs, stream=None) -> list:
        """omg_halo_global_sum_dd: [(hi, lo), ...] local partial sums -> the global sums (hi parts), same bits on every rank"""
        p = np.ascontiguousarray(local_pairs, dtype=np.float64).reshape(-1, 2)
        out = np.zeros_like(p)
        _chk(lib().omg_halo_global_sum_dd(self.h, _pd(p), p.shape[0], _pd(out), _sh(stream)))
        return [float(x) for x in out[:, 0]]

    def exchange_state(self, state, tracers=None, time_level: int = 0, tracers_time_level: int | None = None, stream=None):
        """omg_halo_exchange_state: h, u and the tracers as ONE message per neighbour (what the time steppers do after a stage)"""
        _chk(lib().omg_halo_exchange_state(self.h, state.h, time_level, tracers.h if tracers is not None else None,
                                           time_level if tracers_time_level is None else tracers_time_level, _sh(stream)))

    def check(self):
        """omg_halo_check: raises if a peer-wire wait of an earlier exchange gave up (ask after synchronising)"""
        _chk(lib().omg_halo_check(self.h))

    def recv_rows(self, per_cell: int, per_edge: int, per_vertex: int = 0) -> int:
        r = C.c_size_t()
        _chk(lib().omg_halo_recv_rows(self.h, C.c_size_t(per_cell), C.c_size_t(per_edge), C.c_size_t(per_vertex), C.byref(r)))
        return r.value

    def set_transport(self, fn):
        """fn(tasks, send_ptrs, send_bytes, recv_ptrs, recv_bytes, stream_handle) -> int"""
        def _cb(_ctx, n, tasks, sp, sb, rp, rb, stream):
            try:
                return int(fn([tasks[i] for i in range(n)], [sp[i] for i in range(n)], [sb[i] for i in range(n)],
                              [rp[i] for i in range(n)], [rb[i] for i in range(n)], stream) or 0)
            except Exception as e:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._cb = TRANSPORT_FN(_cb)
        _chk(lib().omg_halo_set_transport(self.h, self._cb, None))

    def exchange(self, dev_ptr: int, nt: int, rows_size: int, k: int, elem: int, stream=None, row_pitch: int = 0,
                 elem_bytes: int = 8):
        """Halo::exchangeFullArrayHalo on a raw device array [nt][rows_size][row_pitch or k] of elem_bytes-byte values
        (8: R8 / I8, 4: I4 / R4); rank 1 is nt = 1, k = 1."""
        if elem_bytes == 8:
            _chk(lib().omg_halo_exchange(self.h, C.cast(C.c_void_p(dev_ptr), PD), nt, rows_size, k, row_pitch, elem, _sh(stream)))
        else:
            _chk(lib().omg_halo_exchange_bytes(self.h, C.c_void_p(dev_ptr), elem_bytes, nt, rows_size, k, row_pitch, elem,
                                               _sh(stream)))

    def __del__(self):
        try:
            lib().omg_halo_destroy(self.h)
        except Exception:
            pass


_MESH_I4 = {"CellsOnCell": ("C", "ME"), "EdgesOnCell": ("C", "ME"), "NEdgesOnCell": ("C",),
            "VerticesOnCell": ("C", "ME"), "CellsOnEdge": ("E", 2), "EdgesOnEdge": ("E", "ME2"),
            "NEdgesOnEdge": ("E",), "VerticesOnEdge": ("E", 2), "CellsOnVertex": ("V", "VD"),
            "EdgesOnVertex": ("V", "VD")}
_MESH_R8 = {"AreaCell": ("C",), "AreaTriangle": ("V",), "KiteAreasOnVertex": ("V", "VD"), "DvEdge": ("E",),
            "DcEdge": ("E",), "AngleEdge": ("E",), "WeightsOnEdge": ("E", "ME2"), "FEdge": ("E",), "FCell": ("C",),
            "FVertex": ("V",), "BottomDepth": ("C",), "EdgeSignOnCell": ("C", "ME"), "EdgeSignOnVertex": ("V", "VD"),
            "EdgeMask": ("E", "K"), "MeshScalingDel2": ("E",), "MeshScalingDel4": ("E",)}
for _el, _d in (("Cell", "C"), ("Edge", "E"), ("Vertex", "V")):
    for _p in ("X", "Y", "Z", "Lon", "Lat"):
        _MESH_R8[_p + _el] = (_d,)


class HorzMesh:
    def __init__(self, decomp: Decomp, nvertlayers: int, host_only: bool = False):
        self.decomp = decomp
        h = C.c_void_p()
        _chk(lib().omg_mesh_create(decomp.h, nvertlayers, int(host_only), C.byref(h)))
        self.h = h
        self._dims = None

    def get_int(self, name: str) -> int:
        v = C.c_int32()
        _chk(lib().omg_mesh_get_int(self.h, name.encode(), C.byref(v)))
        return v.value

    def __getattr__(self, name):
        if name.startswith("N") or name in ("MaxEdges", "MaxEdges2", "VertexDegree"):
            try:
                return self.get_int(name)
            except OmegaAmdError:
                pass
        raise AttributeError(name)

    def _shape(self, spec):
        if self._dims is None:
            self._dims = {"C": self.get_int("NCellsSize"), "E": self.get_int("NEdgesSize"),
                          "V": self.get_int("NVerticesSize"), "ME": self.get_int("MaxEdges"),
                          "ME2": self.get_int("MaxEdges2"), "VD": self.get_int("VertexDegree"),
                          "K": self.get_int("NVertLayers")}
        return tuple(self._dims[s] if isinstance(s, str) else s for s in spec)

    def get_array(self, name: str) -> np.ndarray:
        if name in _MESH_I4:
            out = np.zeros(self._shape(_MESH_I4[name]), dtype=np.int32)
            _chk(lib().omg_mesh_get_array_i4(self.h, name.encode(), _pi(out), C.c_size_t(out.size)))
        else:
            out = np.zeros(self._shape(_MESH_R8[name]), dtype=np.float64)
            _chk(lib().omg_mesh_get_array_r8(self.h, name.encode(), _pd(out), C.c_size_t(out.size)))
        return out

    def local_arrays(self) -> dict:
        """All host arrays + sizes, in the dict form oracle.Mesh consumes (test use)."""
        L = {n: self.get_int(n) for n in ("NCellsOwned", "NCellsAll", "NCellsSize", "NEdgesOwned", "NEdgesAll",
                                          "NEdgesSize", "NVerticesOwned", "NVerticesAll", "NVerticesSize",
                                          "MaxEdges", "MaxEdges2", "VertexDegree")}
        for n in list(_MESH_I4) + [k for k in _MESH_R8 if k not in ("EdgeSignOnCell", "EdgeSignOnVertex", "EdgeMask",
                                                                    "MeshScalingDel2", "MeshScalingDel4")]:
            L[n] = self.get_array(n)
        return L

    def set_fvertex(self, values: np.ndarray):
        _chk(lib().omg_mesh_set_fvertex(self.h, _pd(np.ascontiguousarray(values, dtype=np.float64))))

    def __del__(self):
        try:
            lib().omg_mesh_destroy(self.h)
        except Exception:
            pass


class HorzOperators:
    """DivergenceOnCell / GradientOnEdge / CurlOnVertex / TangentialReconOnEdge / InterpCellToEdge
    (HorzOperators.h:9-187) on host arrays staged through device buffers (test / tooling use; the
    C entry points omg_horz_* take raw device pointers)."""

    def __init__(self, mesh: HorzMesh):
        self.mesh = mesh

    def _run(self, fn, x: np.ndarray, rows_out: int, n: int, *extra):
        x = np.ascontiguousarray(x, dtype=np.float64)
        one_d = x.ndim == 1
        k = 1 if one_d else x.shape[1]
        din = DeviceBuffer(x)
        dout = DeviceBuffer(np.zeros((rows_out,) if one_d else (rows_out, k)))
        if one_d:
            _chk(fn(self.mesh.h, C.c_void_p(din.ptr), C.c_void_p(dout.ptr), *extra, n, None))
        else:
            _chk(fn(self.mesh.h, C.c_void_p(din.ptr), C.c_void_p(dout.ptr), k, 0, n, None))
        device_synchronize()
        return dout.to_host()

    def divergence(self, vec_edge, n=-1):
        return self._run(lib().omg_horz_divergence, vec_edge, self.mesh.NCellsSize, n)

    def gradient(self, scalar_cell, n=-1):
        return self._run(lib().omg_horz_gradient, scalar_cell, self.mesh.NEdgesSize, n)

    def curl(self, vec_edge, n=-1):
        return self._run(lib().omg_horz_curl, vec_edge, self.mesh.NVerticesSize, n)

    def tangential_recon(self, vec_edge, n=-1):
        return self._run(lib().omg_horz_tangential_recon, vec_edge, self.mesh.NEdgesSize, n)

    def interp_cell_to_edge(self, array_cell, isotropic: bool, n=-1):
        return self._run(lib().omg_horz_interp_cell_to_edge, array_cell, self.mesh.NEdgesSize, n, int(isotropic))


def default_config(**over) -> TendConfig:
    c = TendConfig()
    lib().omg_tend_config_default(C.byref(c))
    for k, v in over.items():
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    return c


class OceanState:
    def __init__(self, mesh: HorzMesh, halo: Halo | None, nvertlayers: int, ntimelevels: int = 2):
        self.mesh, self.halo, self.K = mesh, halo, nvertlayers
        h = C.c_void_p()
        _chk(lib().omg_state_create(mesh.h, halo.h if halo else None, nvertlayers, ntimelevels, C.byref(h)))
        self.h = h

    def copy_to_device(self, h=None, u=None, time_level: int = 0):
        _chk(lib().omg_state_copy_to_device(self.h, time_level, _pd(h), _pd(u)))

    def copy_to_host(self, time_level: int = 0):
        h = np.zeros((self.mesh.NCellsSize, self.K))
        u = np.zeros((self.mesh.NEdgesSize, self.K))
        _chk(lib().omg_state_copy_to_host(self.h, time_level, _pd(h), _pd(u)))
        return h, u

    def device_ptr(self, which: int, time_level: int = 0) -> int:
        p = PD()
        _chk(lib().omg_state_device_ptr(self.h, time_level, which, C.byref(p)))
        return C.cast(p, C.c_void_p).value

    def exchange_halo(self, time_level: int = 0, stream=None):
        _chk(lib().omg_state_exchange_halo(self.h, time_level, _sh(stream)))

    def update_time_levels(self, stream=None):
        _chk(lib().omg_state_update_time_levels(self.h, _sh(stream)))

    def __del__(self):
        try:
            lib().omg_state_destroy(self.h)
        except Exception:
            pass


class Tracers:
    def __init__(self, mesh: HorzMesh, halo: Halo | None, nvertlayers: int, ntracers: int, ntimelevels: int = 2):
        self.mesh, self.K, self.NT = mesh, nvertlayers, ntracers
        h = C.c_void_p()
        _chk(lib().omg_tracers_create(mesh.h, halo.h if halo else None, nvertlayers, ntracers, ntimelevels, C.byref(h)))
        self.h = h

    def copy_to_device(self, tr, time_level: int = 0):
        _chk(lib().omg_tracers_copy_to_device(self.h, time_level, _pd(tr)))

    def copy_to_host(self, time_level: int = 0):
        tr = np.zeros((max(self.NT, 1), self.mesh.NCellsSize, self.K))
        if self.NT > 0:
            _chk(lib().omg_tracers_copy_to_host(self.h, time_level, _pd(tr)))
        return tr

    def device_ptr(self, time_level: int = 0) -> int:
        p = PD()
        _chk(lib().omg_tracers_device_ptr(self.h, time_level, C.byref(p)))
        return C.cast(p, C.c_void_p).value

    def exchange_halo(self, time_level: int = 0, stream=None):
        _chk(lib().omg_tracers_exchange_halo(self.h, time_level, _sh(stream)))

    def update_time_levels(self, stream=None):
        _chk(lib().omg_tracers_update_time_levels(self.h, _sh(stream)))

    def __del__(self):
        try:
            lib().omg_tracers_destroy(self.h)
        except Exception:
            pass


AUX_SHAPES = {"KineticEnergyCell": "C", "VelocityDivCell": "C", "FluxLayerThickEdge": "E", "MeanLayerThickEdge": "E",
              "SshCell": "C", "RelVortVertex": "V", "NormRelVortVertex": "V", "NormPlanetVortVertex": "V",
              "NormRelVortEdge": "E", "NormPlanetVortEdge": "E", "Del2Edge": "E", "Del2DivCell": "C",
              "Del2RelVortVertex": "V", "HTracersEdge": "TE", "Del2TracersCell": "TC", "NormalStressEdge": "E1",
              "ZonalStressCell": "C1", "MeridStressCell": "C1"}


class AuxiliaryState:
    def __init__(self, mesh: HorzMesh, halo: Halo | None, nvertlayers: int, ntracers: int):
        self.mesh, self.K, self.NT = mesh, nvertlayers, ntracers
        h = C.c_void_p()
        _chk(lib().omg_aux_create(mesh.h, halo.h if halo else None, nvertlayers, ntracers, C.byref(h)))
        self.h = h

    def set_options(self, flux_thickness_upwind=False, flux_tracer_upwind=False, wind_interp_isotropic=True):
        _chk(lib().omg_aux_set_options(self.h, int(flux_thickness_upwind), int(flux_tracer_upwind),
                                       int(wind_interp_isotropic)))

    def compute_mom_aux(self, state: OceanState, thick_tl=0, vel_tl=0, stream=None):
        _chk(lib().omg_aux_compute_mom_aux(self.h, state.h, thick_tl, vel_tl, _sh(stream)))

    def compute_all(self, state: OceanState, tracers: Tracers, tracer_tl=0, thick_tl=0, vel_tl=0, stream=None):
        _chk(lib().omg_aux_compute_all(self.h, state.h, tracers.h, tracer_tl, thick_tl, vel_tl, _sh(stream)))

    def _shape(self, name):
        m, K, nt = self.mesh, self.K, max(self.NT, 1)
        rows = {"C": m.NCellsSize, "E": m.NEdgesSize, "V": m.NVerticesSize}
        s = AUX_SHAPES[name]
        if s in rows:
            return (rows[s], K)
        if s[0] == "T":
            return (nt, rows[s[1]], K)
        return (rows[s[0]],)

    def get(self, name: str) -> np.ndarray:
        out = np.zeros(self._shape(name))
        _chk(lib().omg_aux_copy_to_host(self.h, name.encode(), _pd(out), C.c_size_t(out.size)))
        return out

    def set(self, name: str, values: np.ndarray):
        v = np.ascontiguousarray(values, dtype=np.float64)
        _chk(lib().omg_aux_copy_to_device(self.h, name.encode(), _pd(v), C.c_size_t(v.size)))

    def __del__(self):
        try:
            lib().omg_aux_destroy(self.h)
        except Exception:
            pass


def fused_limit(ncells_size: int, nedges_size: int, nvertices_size: int, max_edges: int, nvertlayers: int):
    """omg_tend_fused_limit: (True, "") if the fused RHS covers arrays of these row counts (sentinel row included), else
    (False, reason).  Sizes only: needs neither a mesh nor a device."""
    ok, why = C.c_int(), C.create_string_buffer(1024)
    _chk(lib().omg_tend_fused_limit(C.c_int64(ncells_size), C.c_int64(nedges_size), C.c_int64(nvertices_size), int(max_edges),
                                    int(nvertlayers), C.byref(ok), why, C.c_size_t(1024)))
    return bool(ok.value), why.value.decode()


class Tendencies:
    def __init__(self, mesh: HorzMesh, nvertlayers: int, ntracers: int, config: TendConfig | None = None,
                 allow_reference_structured: bool = False):
        """Raises for a mesh outside the fused RHS (fused_limit) unless allow_reference_structured: then
        compute_all_tendencies takes the reference-structured 23-launch path."""
        self.mesh, self.K, self.NT = mesh, nvertlayers, ntracers
        self.config = config if config is not None else default_config()
        h = C.c_void_p()
        create = lib().omg_tend_create_reference_structured if allow_reference_structured else lib().omg_tend_create
        _chk(create(mesh.h, nvertlayers, ntracers, C.byref(self.config), C.byref(h)))
        self.h = h

    def set_fused(self, on: bool):
        _chk(lib().omg_tend_set_fused(self.h, int(on)))

    def set_graphs(self, on: bool):
        _chk(lib().omg_tend_set_graphs(self.h, int(on)))

    def graph_stats(self):
        c, r = C.c_int64(), C.c_int64()
        _chk(lib().omg_tend_graph_stats(self.h, C.byref(c), C.byref(r)))
        return {"captures": c.value, "replays": r.value}

    def compute_all_tendencies(self, state, aux, tracers, tracer_tl=0, thick_tl=0, vel_tl=0, stream=None):
        _chk(lib().omg_tend_compute_all(self.h, state.h, aux.h, tracers.h, tracer_tl, thick_tl, vel_tl, _sh(stream)))

    def compute_thickness_tendencies(self, state, aux, thick_tl=0, vel_tl=0, stream=None):
        _chk(lib().omg_tend_compute_thickness(self.h, state.h, aux.h, thick_tl, vel_tl, _sh(stream)))

    def compute_velocity_tendencies(self, state, aux, thick_tl=0, vel_tl=0, stream=None):
        _chk(lib().omg_tend_compute_velocity(self.h, state.h, aux.h, thick_tl, vel_tl, _sh(stream)))

    def compute_tracer_tendencies(self, state, aux, tracers, tracer_tl=0, thick_tl=0, vel_tl=0, stream=None):
        _chk(lib().omg_tend_compute_tracer(self.h, state.h, aux.h, tracers.h, tracer_tl, thick_tl, vel_tl, _sh(stream)))

    def compute_thickness_tendencies_only(self, state, aux, thick_tl=0, vel_tl=0, stream=None):
        _chk(lib().omg_tend_compute_thickness_only(self.h, state.h, aux.h, thick_tl, vel_tl, _sh(stream)))

    def compute_velocity_tendencies_only(self, state, aux, thick_tl=0, vel_tl=0, stream=None):
        _chk(lib().omg_tend_compute_velocity_only(self.h, state.h, aux.h, thick_tl, vel_tl, _sh(stream)))

    def compute_tracer_tendencies_only(self, state, aux, tracers, tracer_tl=0, thick_tl=0, vel_tl=0, stream=None):
        _chk(lib().omg_tend_compute_tracer_only(self.h, state.h, aux.h, tracers.h, tracer_tl, thick_tl, vel_tl,
                                                _sh(stream)))

    def use_manufactured_solution(self, mesh, wavelength_x: float, wavelength_y: float, amplitude: float):
        """Tendencies config UseCustomTendency + ManufacturedSolutionTendency (CustomTendencyTerms.cpp)."""
        _chk(lib().omg_tend_use_manufactured_solution(self.h, mesh.h, C.c_double(wavelength_x), C.c_double(wavelength_y),
                                                      C.c_double(amplitude)))

    def set_custom_tendency(self, which: int, fn):
        """Tendencies::CustomThicknessTend (which 0) / CustomVelocityTend (which 1) as a Python callable
        fn(tend_ptr, h_ptr, u_ptr, n_rows_all, n_rows_size, K, row_pitch, time_seconds, stream_handle); None clears it."""
        if not hasattr(self, "_custom"):
            self._custom = {}
        if fn is None:
            _chk(lib().omg_tend_set_custom_tendency(self.h, which, C.cast(None, CUSTOM_TEND_FN), None))
            self._custom.pop(which, None)
            return

        def _cb(_ctx, tend, h, u, nall, nsize, k, pitch, t, stream):
            try:
                fn(tend, h, u, nall, nsize, k, pitch, t, stream)
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._custom[which] = CUSTOM_TEND_FN(_cb)
        _chk(lib().omg_tend_set_custom_tendency(self.h, which, self._custom[which], None))

    def clear_custom_tendencies(self):
        _chk(lib().omg_tend_clear_custom_tendencies(self.h))

    def set_time(self, seconds: float):
        """model time (s since the reference time) the custom tendencies see in direct compute_* calls"""
        _chk(lib().omg_tend_set_time(self.h, C.c_double(seconds)))

    def kernel_timing(self, on: bool):
        _chk(lib().omg_tend_kernel_timing(self.h, int(on)))

    def collect_kernel_times(self):
        """[(kernel name, mean ms)] over the RHS evaluations recorded since kernel_timing(True)."""
        ms = (C.c_double * 16)()
        nk, ns = C.c_int(), C.c_int()
        _chk(lib().omg_tend_collect_kernel_times(self.h, ms, C.byref(nk), C.byref(ns)))
        L = lib()
        L.omg_tend_kernel_name.restype = C.c_char_p
        if ns.value == 0:
            return []
        out = [(L.omg_tend_kernel_name(i).decode(), ms[i] / ns.value) for i in range(nk.value)]
        return [(k, v) for k, v in out if k]

    def device_ptr(self, which: int):
        """omg_tend_device_ptr: (device address, number of values incl. the row padding) of LayerThicknessTend (0),
        NormalVelocityTend (1), TracerTend (2)"""
        p, n = C.POINTER(C.c_double)(), C.c_size_t()
        _chk(lib().omg_tend_device_ptr(self.h, which, C.byref(p), C.byref(n)))
        return C.cast(p, C.c_void_p).value, n.value

    def get(self, which: int) -> np.ndarray:
        m = self.mesh
        shape = [(m.NCellsSize, self.K), (m.NEdgesSize, self.K), (max(self.NT, 1), m.NCellsSize, self.K)][which]
        out = np.zeros(shape)
        _chk(lib().omg_tend_copy_to_host(self.h, which, _pd(out), C.c_size_t(out.size)))
        return out

    def __del__(self):
        try:
            lib().omg_tend_destroy(self.h)
        except Exception:
            pass


class TimeStepper:
    def __init__(self, kind: str, dt: float, tend: Tendencies, aux: AuxiliaryState, mesh: HorzMesh,
                 halo: Halo | None, tracers: Tracers):
        self.refs = (tend, aux, mesh, halo, tracers)
        h = C.c_void_p()
        _chk(lib().omg_stepper_create(kind.encode(), C.c_double(dt), tend.h, aux.h, mesh.h, halo.h if halo else None,
                                      tracers.h, C.byref(h)))
        self.h = h

    def do_step(self, state: OceanState, stream=None):
        _chk(lib().omg_stepper_do_step(self.h, state.h, _sh(stream)))

    def set_start_time(self, seconds: float):
        _chk(lib().omg_stepper_set_start_time(self.h, C.c_double(seconds)))

    @property
    def time(self) -> float:
        v = C.c_double()
        _chk(lib().omg_stepper_get_time(self.h, C.byref(v)))
        return v.value

    def graph_stats(self):
        c, r = C.c_int64(), C.c_int64()
        _chk(lib().omg_stepper_graph_stats(self.h, C.byref(c), C.byref(r)))
        return {"captures": c.value, "replays": r.value}

    def change_time_step(self, dt: float):
        _chk(lib().omg_stepper_change_time_step(self.h, C.c_double(dt)))

    def set_option(self, name: str, value: bool):
        """RungeKutta4: "FuseStageUpdates" (default on), "StoreStageTendencies" (default off)."""
        _chk(lib().omg_stepper_set_option(self.h, name.encode(), int(value)))

    def __del__(self):
        try:
            lib().omg_stepper_destroy(self.h)
        except Exception:
            pass


def update_by_tend(out_ptr: int, in_ptr: int, tend_ptr: int, coeff: float, n_rows: int, k: int, stream_handle=None):
    """out = in + coeff * tend on raw [n_rows][k] device arrays (TimeStepper::update*ByTend's kernel)."""
    _chk(lib().omg_update_by_tend(C.c_void_p(out_ptr), C.c_void_p(in_ptr), C.c_void_p(tend_ptr), C.c_double(coeff),
                                  n_rows, k, C.c_void_p(stream_handle) if stream_handle else None))


def coeff_seconds(mult: float, dt: float) -> float:
    out = C.c_double()
    _chk(lib().omg_stepper_coeff_seconds(C.c_double(mult), C.c_double(dt), C.byref(out)))
    return out.value
