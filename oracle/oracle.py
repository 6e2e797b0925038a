"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see omega_oracle.h).

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libomega_oracle.so")
_SO_NATIVE = os.path.join(_HERE, "build", "native", "libomega_oracle.so")

PD = C.POINTER(C.c_double)
PI = C.POINTER(C.c_int)


class OrcMesh(C.Structure):
    _fields_ = ([(n, C.c_int) for n in (
        "NCellsOwned", "NCellsAll", "NCellsSize", "NEdgesOwned", "NEdgesAll", "NEdgesSize",
        "NVerticesOwned", "NVerticesAll", "NVerticesSize", "MaxEdges", "MaxEdges2",
        "VertexDegree", "NVertLayers")]
        + [(n, PI) for n in (
            "NEdgesOnCell", "EdgesOnCell", "CellsOnCell", "VerticesOnCell", "CellsOnEdge",
            "VerticesOnEdge", "NEdgesOnEdge", "EdgesOnEdge", "CellsOnVertex", "EdgesOnVertex")]
        + [(n, PD) for n in (
            "AreaCell", "AreaTriangle", "KiteAreasOnVertex", "DcEdge", "DvEdge", "AngleEdge",
            "WeightsOnEdge", "FVertex", "BottomDepth", "EdgeSignOnCell", "EdgeSignOnVertex",
            "EdgeMask", "MeshScalingDel2", "MeshScalingDel4")])


CONFIG_FLAGS = ("ThicknessFluxTendencyEnable", "PVTendencyEnable", "KETendencyEnable",
                "SSHTendencyEnable", "VelDiffTendencyEnable", "VelHyperDiffTendencyEnable",
                "WindForcingTendencyEnable", "BottomDragTendencyEnable",
                "TracerHorzAdvTendencyEnable", "TracerDiffTendencyEnable",
                "TracerHyperDiffTendencyEnable", "FluxThicknessUpwind", "FluxTracerUpwind",
                "WindInterpIsotropic")
CONFIG_REALS = ("ViscDel2", "ViscDel4", "DivFactor", "EddyDiff2", "EddyDiff4", "Density0",
                "BottomDragCoeff")


class OrcManufactured(C.Structure):
    _fields_ = ([(n, C.c_double) for n in ("H0", "Eta0", "Kx", "Ky", "AngFreq", "Grav", "ViscDel2", "ViscDel4")]
                + [(n, C.c_int) for n in ("VelDiffTendencyEnable", "VelHyperDiffTendencyEnable")]
                + [(n, PD) for n in ("XCell", "YCell", "XEdge", "YEdge", "FEdge")])


class OrcConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in CONFIG_FLAGS] + [(n, C.c_double) for n in CONFIG_REALS]


AUX_FIELDS = (  # name, element ('C','E','V'), per-tracer?, has K?
    ("KineticEnergyCell", "C", False, True), ("VelocityDivCell", "C", False, True),
    ("FluxLayerThickEdge", "E", False, True), ("MeanLayerThickEdge", "E", False, True),
    ("SshCell", "C", False, True),
    ("RelVortVertex", "V", False, True), ("NormRelVortVertex", "V", False, True),
    ("NormPlanetVortVertex", "V", False, True),
    ("NormRelVortEdge", "E", False, True), ("NormPlanetVortEdge", "E", False, True),
    ("Del2Edge", "E", False, True), ("Del2DivCell", "C", False, True),
    ("Del2RelVortVertex", "V", False, True),
    ("HTracersEdge", "E", True, True), ("Del2TracersCell", "C", True, True),
    ("NormalStressEdge", "E", False, False), ("ZonalStressCell", "C", False, False),
    ("MeridStressCell", "C", False, False),
)


class OrcAux(C.Structure):
    _fields_ = [(n, PD) for n, *_ in AUX_FIELDS]


class OrcState(C.Structure):
    _fields_ = [("h", PD * 2), ("u", PD * 2), ("tr", PD * 2),
                ("hProvis", PD), ("uProvis", PD), ("trProvis", PD),
                ("hTend", PD), ("uTend", PD), ("trTend", PD)]


def build(force: bool = False, native: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile).  native: the `-O3 -march=native` variant for
    the CPU-baseline timing, always compiled on the machine that runs it (never shipped)."""
    src = os.path.join(_HERE, "omega_oracle.c")
    hdr = os.path.join(_HERE, "omega_oracle.h")
    if native:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
        return _SO_NATIVE
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def host_cores() -> dict:
    """What this process may use of the host: logical CPUs of the machine, CPUs in its affinity mask, and the CPU
    bandwidth quota of its cgroup (v2 cpu.max or v1 cfs quota), if any."""
    out = {"logical_cpus": os.cpu_count() or 1, "affinity": None, "cgroup_quota_cores": None}
    try:
        out["affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()
            if q != "max":
                out["cgroup_quota_cores"] = float(q) / float(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, p = float(fq.read()), float(fp.read())
                if q > 0:
                    out["cgroup_quota_cores"] = q / p
        except (OSError, ValueError):
            pass
    avail = out["affinity"] or out["logical_cpus"]
    if out["cgroup_quota_cores"]:
        avail = max(1, min(avail, int(out["cgroup_quota_cores"] + 0.5)))
    out["cores_available"] = avail
    return out


def default_threads() -> int:
    """OpenMP team of the oracle: the host cores this process may use (affinity mask and cgroup quota), capped at
    OMEGA_ORACLE_THREADS (default 16 = a one-GPU box's CPU share: such a box shows every hardware thread of its host in
    the affinity mask, and a team of hundreds of spinning threads on a 16-core share makes every small parallel region
    take milliseconds).  Tests and bench.py's cpu_baseline use the same number and report both."""
    return max(1, min(host_cores()["cores_available"], int(os.environ.get("OMEGA_ORACLE_THREADS", "16"))))


def use_native_build():
    """bench.py's cpu_baseline: load the -O3 -march=native build (must be called before the first lib())."""
    global _SO
    assert _lib is None, "oracle library already loaded"
    _SO = build(native=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _SO if os.path.exists(_SO) else build()
        L = C.CDLL(path)
        L.orc_coeff_seconds.restype = C.c_double
        L.orc_coeff_seconds.argtypes = [C.c_double, C.c_double]
        L.orc_get_max_threads.restype = C.c_int
        # OpenMP team size: a GPU box exposes every hardware thread of the host but gives the job a
        # 16-core share; a team of hundreds of spinning threads on 16 cores makes each of the many
        # small parallel regions of a time step take milliseconds
        L.orc_set_num_threads(default_threads())
        _lib = L
    return _lib


def _pd(a):
    if a is None:
        return PD()
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(PD)


def _pi(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(PI)


def _pad_rows(a, n_size, fill):
    """Copy `a` (n rows) into an array of n_size rows; extra rows = fill."""
    out = np.full((n_size,) + a.shape[1:], fill, dtype=a.dtype)
    out[: a.shape[0]] = a
    return out


def single_rank_local_arrays(g: dict) -> dict:
    """Local (HorzMesh-form) arrays of a global meshgen mesh on ONE rank: every element
    owned, no halo, one zero sentinel row per index space, missing (-1) -> sentinel index
    (reference: O/src/base/Decomp.cpp:553-574; O/src/ocn/HorzMesh.cpp copies)."""
    nC, nE, nV = g["nCells"], g["nEdges"], g["nVertices"]
    L = {"NCellsOwned": nC, "NCellsAll": nC, "NCellsSize": nC + 1,
         "NEdgesOwned": nE, "NEdgesAll": nE, "NEdgesSize": nE + 1,
         "NVerticesOwned": nV, "NVerticesAll": nV, "NVerticesSize": nV + 1,
         "MaxEdges": g["maxEdges"], "MaxEdges2": 2 * g["maxEdges"],
         "VertexDegree": g["vertexDegree"]}

    def conn(name, n_own, target_n):
        a = g[name].astype(np.int32)
        a = np.where(a < 0, target_n, a).astype(np.int32)
        return _pad_rows(a, n_own + 1, target_n)

    L["NEdgesOnCell"] = _pad_rows(g["nEdgesOnCell"].astype(np.int32), nC + 1, 0)
    L["EdgesOnCell"] = conn("edgesOnCell", nC, nE)
    L["CellsOnCell"] = conn("cellsOnCell", nC, nC)
    L["VerticesOnCell"] = conn("verticesOnCell", nC, nV)
    L["CellsOnEdge"] = conn("cellsOnEdge", nE, nC)
    L["VerticesOnEdge"] = conn("verticesOnEdge", nE, nV)
    L["NEdgesOnEdge"] = _pad_rows(g["nEdgesOnEdge"].astype(np.int32), nE + 1, 0)
    L["EdgesOnEdge"] = conn("edgesOnEdge", nE, nE)
    L["CellsOnVertex"] = conn("cellsOnVertex", nV, nC)
    L["EdgesOnVertex"] = conn("edgesOnVertex", nV, nE)
    for name, src, n in (("AreaCell", "areaCell", nC), ("AreaTriangle", "areaTriangle", nV),
                         ("KiteAreasOnVertex", "kiteAreasOnVertex", nV), ("DcEdge", "dcEdge", nE),
                         ("DvEdge", "dvEdge", nE), ("AngleEdge", "angleEdge", nE),
                         ("WeightsOnEdge", "weightsOnEdge", nE), ("FVertex", "fVertex", nV),
                         ("BottomDepth", "bottomDepth", nC), ("FEdge", "fEdge", nE), ("FCell", "fCell", nC)):
        L[name] = _pad_rows(np.ascontiguousarray(g[src], dtype=np.float64), n + 1, 0.0)
    for el, n in (("Cell", nC), ("Edge", nE), ("Vertex", nV)):
        for pre in ("x", "y", "z", "lon", "lat"):
            L[pre.capitalize() + el] = _pad_rows(np.asarray(g[pre + el], dtype=np.float64), n + 1, 0.0)
    return L


class Mesh:
    """Owns the numpy arrays behind an orc_mesh and its derived arrays."""

    def __init__(self, local: dict, nvertlayers: int):
        self.a = dict(local)
        K = int(nvertlayers)
        self.K = K
        a = self.a
        a["EdgeSignOnCell"] = np.zeros((a["NCellsSize"], a["MaxEdges"]))
        a["EdgeSignOnVertex"] = np.zeros((a["NVerticesSize"], a["VertexDegree"]))
        a["EdgeMask"] = np.zeros((a["NEdgesSize"], K))
        a["MeshScalingDel2"] = np.zeros(a["NEdgesSize"])
        a["MeshScalingDel4"] = np.zeros(a["NEdgesSize"])
        s = OrcMesh()
        for n, t in OrcMesh._fields_:
            if t is C.c_int:
                setattr(s, n, K if n == "NVertLayers" else int(a[n]))
            elif t is PI:
                setattr(s, n, _pi(a[n]))
            else:
                setattr(s, n, _pd(a[n]))
        self.s = s
        lib().orc_mesh_derive(C.byref(s))

    def __getattr__(self, n):
        try:
            return self.__dict__["a"][n]
        except KeyError:
            raise AttributeError(n)

    @classmethod
    def single_rank(cls, g: dict, nvertlayers: int) -> "Mesh":
        return cls(single_rank_local_arrays(g), nvertlayers)

    def rows(self, el: str) -> int:
        return {"C": self.a["NCellsSize"], "E": self.a["NEdgesSize"], "V": self.a["NVerticesSize"]}[el]


def default_config(**over) -> OrcConfig:
    c = OrcConfig()
    lib().orc_config_default(C.byref(c))
    for k, v in over.items():
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    return c


class Aux:
    def __init__(self, mesh: Mesh, ntracers: int):
        self.arr = {}
        s = OrcAux()
        for name, el, per_tr, has_k in AUX_FIELDS:
            shape = (mesh.rows(el),) + ((mesh.K,) if has_k else ())
            if per_tr:
                shape = (max(ntracers, 1),) + shape
            self.arr[name] = np.zeros(shape)
            setattr(s, name, _pd(self.arr[name]))
        self.s = s

    def __getitem__(self, n):
        return self.arr[n]


class Oracle:
    """Convenience object: mesh + config + aux + tendency arrays, methods named after the
    reference entry points."""

    def __init__(self, mesh: Mesh, ntracers: int, config: OrcConfig | None = None):
        self.m = mesh
        self.NT = int(ntracers)
        self.c = config if config is not None else default_config()
        self.aux = Aux(mesh, ntracers)
        K = mesh.K
        self.hTend = np.zeros((mesh.NCellsSize, K))
        self.uTend = np.zeros((mesh.NEdgesSize, K))
        self.trTend = np.zeros((max(self.NT, 1), mesh.NCellsSize, K))
        self.L = lib()

    def _r(self):
        return C.byref(self.m.s), C.byref(self.c), C.byref(self.aux.s)

    def compute_all_aux(self, h, u, tr):
        m, c, a = self._r()
        self.L.orc_aux_compute_all(m, c, a, self.NT, _pd(h), _pd(u), _pd(tr))

    def compute_mom_aux(self, h, u):
        m, c, a = self._r()
        self.L.orc_aux_compute_mom_aux(m, c, a, _pd(h), _pd(u))

    def compute_all_tendencies(self, h, u, tr):
        m, c, a = self._r()
        self.L.orc_tend_compute_all(m, c, a, self.NT, _pd(self.hTend), _pd(self.uTend),
                                    _pd(self.trTend), _pd(h), _pd(u), _pd(tr))
        return self.hTend, self.uTend, self.trTend

    def compute_thickness_tendencies(self, h, u):
        m, c, a = self._r()
        self.L.orc_tend_compute_thickness(m, c, a, _pd(self.hTend), _pd(h), _pd(u))
        return self.hTend

    def compute_velocity_tendencies(self, h, u):
        m, c, a = self._r()
        self.L.orc_tend_compute_velocity(m, c, a, _pd(self.uTend), _pd(h), _pd(u))
        return self.uTend

    def compute_tracer_tendencies(self, h, u, tr):
        m, c, a = self._r()
        self.L.orc_tend_compute_tracer(m, c, a, self.NT, _pd(self.trTend), _pd(h), _pd(u), _pd(tr))
        return self.trTend

    def use_manufactured_solution(self, wavelength_x=None, wavelength_y=None, amplitude=None):
        """Tendencies config UseCustomTendency + ManufacturedSolutionTendency (Tendencies.cpp:41-64):
        from now on the thickness / velocity group functions of the ORACLE LIBRARY add the manufactured
        terms (process-wide switch: call with no arguments to turn it off)."""
        if wavelength_x is None:
            self.L.orc_set_custom_tendency(None)
            self.ms = None
            return None
        ms = OrcManufactured()
        m, c, _ = self._r()
        self.L.orc_manufactured_init(C.byref(ms), m, c, C.c_double(wavelength_x), C.c_double(wavelength_y),
                                     C.c_double(amplitude))
        self._ms_keep = [np.ascontiguousarray(getattr(self.m, n), dtype=np.float64)
                         for n in ("XCell", "YCell", "XEdge", "YEdge", "FEdge")]
        ms.XCell, ms.YCell, ms.XEdge, ms.YEdge, ms.FEdge = [_pd(a) for a in self._ms_keep]
        self.ms = ms
        self.L.orc_set_custom_tendency(C.byref(ms))
        return ms

    def use_decay_velocity_tendency(self, coeff=None):
        """The custom velocity tendency of the reference's TimeStepperTest (du/dt = -coeff u,
        TimeStepperTest.cpp:49-73); process-wide switch, None turns it off."""
        self.L.orc_set_decay_velocity_tendency(0 if coeff is None else 1, C.c_double(0.0 if coeff is None else coeff))

    def set_time(self, t: float):
        """model time (s since the reference time) seen by the custom tendencies in direct tendency calls"""
        self.L.orc_set_time(C.c_double(t))

    def make_state(self, h, u, tr):
        """State with two time levels; level 0 initialised from h,u,tr (copied)."""
        st = {"h": [h.copy(), np.zeros_like(h)], "u": [u.copy(), np.zeros_like(u)],
              "tr": [tr.copy(), np.zeros_like(tr)],
              "hProvis": np.zeros_like(h), "uProvis": np.zeros_like(u), "trProvis": np.zeros_like(tr)}
        return st

    def step(self, kind: str, st: dict, dt: float, exchange=None, sim_time: float = 0.0):
        """One doStep of 'rk4' | 'rk2' | 'fb' starting at model time sim_time; swaps the time levels
        afterwards (OceanState::updateTimeLevels)."""
        self.L.orc_set_sim_time(C.c_double(sim_time))
        s = OrcState()
        for i in range(2):
            s.h[i], s.u[i], s.tr[i] = _pd(st["h"][i]), _pd(st["u"][i]), _pd(st["tr"][i])
        s.hProvis, s.uProvis, s.trProvis = _pd(st["hProvis"]), _pd(st["uProvis"]), _pd(st["trProvis"])
        s.hTend, s.uTend, s.trTend = _pd(self.hTend), _pd(self.uTend), _pd(self.trTend)
        fn = {"rk4": self.L.orc_rk4_step, "rk2": self.L.orc_rk2_step, "fb": self.L.orc_fb_step}[kind]
        m, c, a = self._r()
        XF = C.CFUNCTYPE(None, C.c_void_p, PD, PD, PD)
        if exchange is None:
            cb = C.cast(None, XF)
        else:
            by_addr = {}
            for k in ("h", "u", "tr"):
                for arr in st[k]:
                    by_addr[arr.ctypes.data] = arr
            for k in ("hProvis", "uProvis", "trProvis"):
                by_addr[st[k].ctypes.data] = st[k]

            def _cb(_ctx, ph, pu, ptr):
                exchange(by_addr[C.addressof(ph.contents)], by_addr[C.addressof(pu.contents)],
                         by_addr[C.addressof(ptr.contents)])
            cb = XF(_cb)
        fn(m, c, a, self.NT, C.byref(s), C.c_double(dt), cb, None)
        for k in ("h", "u", "tr"):
            st[k].reverse()
        return st


def coeff_seconds(mult: float, dt: float) -> float:
    return lib().orc_coeff_seconds(mult, dt)
