"""probe: does the RHS time depend on WHERE the arrays were allocated?  Same process, same mesh: the state / aux /
tendency objects are created several times over and the fused RHS is timed on each set."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import omega_amd as oa
from omega_amd.meshgen import planar_hex, synthetic_state

oa.device_init(0)
K, NT = 80, 6
g = planar_hex(680, 680, 30e3)
gm = oa.GlobalMesh(g)
d = oa.Decomp(gm, 1, 0, 3, local_order="curve")
mesh = oa.HorzMesh(d, K)
cid, eid = d.get_array("CellID"), d.get_array("EdgeID")
hg, ug, trg = synthetic_state(g, K, NT)
def loc(a, ids, rows):
    out = np.zeros(a.shape[:-2] + (rows, a.shape[-1])); out[..., :rows-1, :] = a[..., ids[:rows-1]-1, :]; return out
h, u, tr = loc(hg, cid, mesh.NCellsSize), loc(ug, eid, mesh.NEdgesSize), loc(trg, cid, mesh.NCellsSize)
stream = oa.Stream()
keep = []
plans = []   # (a sweep over alignment / skew of the raw allocations was tried here: no relation to the time -- the
             # spread comes with the physical pages an allocation happens to get, not with its virtual address)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    pad = oa.DeviceBuffer(np.zeros(1 + rep * 37 * 1024)) if rep % 2 else None     # perturb the allocator between sets
    state = oa.OceanState(mesh, None, K, 2); tracers = oa.Tracers(mesh, None, K, NT, 2)
    aux = oa.AuxiliaryState(mesh, None, K, NT); tend = oa.Tendencies(mesh, K, NT, oa.default_config())
    state.copy_to_device(h, u, 0); tracers.copy_to_device(tr, 0)
    for _ in range(3):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    oa.device_synchronize()
    res = []
    for trial in range(2):
        tend.kernel_timing(True)
        for _ in range(10):
            tend.compute_all_tendencies(state, aux, tracers, stream=stream)
        oa.device_synchronize(); tend.kernel_timing(False)
        kt = tend.collect_kernel_times()
        res.append([round(v, 3) for _, v in kt])
    print(f"[placement] set {rep} {plans[rep % len(plans)] if plans else ''}: sum {round(sum(res[-1]), 3)} kernels ms (3 trials) {res}", flush=True)
    keep.append((state, tracers, aux, tend, pad))
    if rep % 3 == 2:
        keep.clear()
