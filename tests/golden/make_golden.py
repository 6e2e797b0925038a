"""Generates tests/golden/*.npz: regression vectors of the CPU oracle (oracle/, itself pinned to the
reference's known-answer norms by tests/test_oracle_known_answers.py) for BASELINE.json configs[0]
(planar periodic 16x16 cells, 4 levels, 1 tracer) and a small spherical case.  Inputs are the seeded
synthetic state (omega_amd.meshgen.synthetic_state, seed 20251003); outputs are the three tendencies of
one computeAllTendencies and the state after one Forward-Backward and one RK4 step (dt = 600 s).
The reference itself cannot be built or run here, so these are NOT reference
outputs: they freeze the oracle so that neither it nor the HIP path can drift unnoticed.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from omega_amd.meshgen import icosahedral_points, planar_hex, spherical_voronoi, synthetic_state  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    "planar16x16_k4_nt1": (lambda: planar_hex(16, 16, 30.0e3), 4, 1),
    "ico2_k6_nt2": (lambda: spherical_voronoi(points=icosahedral_points(2), lloyd=2), 6, 2),
}


def pad(x):
    out = np.zeros(x.shape[:-2] + (x.shape[-2] + 1, x.shape[-1]))
    out[..., :-1, :] = x
    return out


def compute(name):
    make, K, NT = CASES[name]
    g = make()
    M = O.Mesh.single_rank(g, K)
    orc = O.Oracle(M, NT)
    h, u, tr = (pad(a) for a in synthetic_state(g, K, NT))
    hT, uT, trT = (a.copy() for a in orc.compute_all_tendencies(h, u, tr))
    out = {"hTend": hT, "uTend": uT, "trTend": trT}
    for kind in ("fb", "rk4"):
        st = orc.make_state(h, u, tr)
        orc.step(kind, st, 600.0)
        out[f"{kind}_h"], out[f"{kind}_u"], out[f"{kind}_tr"] = st["h"][0], st["u"][0], st["tr"][0]
    return g, K, NT, out


if __name__ == "__main__":
    for name in CASES:
        _, _, _, out = compute(name)
        np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
