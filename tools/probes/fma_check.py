import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import omega_amd as oa
from omega_amd.meshgen import planar_hex
from tests.problem import Problem
oa.device_init(0)
P = Problem(planar_hex(40, 32, 30e3), 80, 6)
P.tend.compute_all_tendencies(P.state, P.aux, P.tracers); oa.device_synchronize()
hT, uT, trT = P.oracle.compute_all_tendencies(P.h, P.u, P.tr)
m = P.mesh
for nm, g, r, n in (("h", P.tend.get(0), hT, m.NCellsOwned), ("u", P.tend.get(1), uT, m.NEdgesOwned), ("tr", P.tend.get(2), trT, m.NCellsOwned)):
    g, r = g[..., :n, :], r[..., :n, :]
    d = np.abs(g - r)
    with np.errstate(divide="ignore", invalid="ignore"):
        el = np.nanmax(np.where(np.abs(r) > 0, d / np.abs(r), 0))
    print(f"[fma] {nm}: bit-identical {np.array_equal(g, r)}; max abs diff / field max {d.max() / np.abs(r).max():.3e}; max element-wise rel diff {el:.3e}")
