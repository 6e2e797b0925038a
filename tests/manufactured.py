"""Manufactured-solution test case (reference: components/omega/src/ocn/CustomTendencyTerms.cpp,
Default.yml:143-146; Bishnu et al. 2024): eta = eta0*sin(phase), u = v = eta0*cos(phase),
phase = kx*x + ky*y - omega*t on a doubly periodic planar hex mesh whose periods are the two
wavelengths.  With the manufactured source terms added to the thickness and velocity tendencies the
discrete solution converges to it at second order."""
import numpy as np

from omega_amd.meshgen import planar_hex

LX = 5.0e6          # Default.yml:144 WavelengthX; WavelengthY 4.33013e6 = LX*sqrt(3)/2 (:145)
H0 = 1000.0
ETA0 = 1.0          # Default.yml:146 Amplitude
F0 = 1.0e-4
GRAV = 9.80665


def mesh(nx):
    return planar_hex(nx, nx, LX / nx, f0=F0, bottom_depth=H0)


def wavelengths(g):
    return g["x_period"], g["y_period"]


def exact(x, y, angle, t, wx, wy):
    """(h at cell centres x,y given as the first pair) -- call separately for cells and edges."""
    kx, ky = 2 * np.pi / wx, 2 * np.pi / wy
    om = np.sqrt(H0 * GRAV * (kx * kx + ky * ky))
    ph = kx * x + ky * y - om * t
    h = H0 + ETA0 * np.sin(ph)
    un = (np.cos(angle) + np.sin(angle)) * ETA0 * np.cos(ph) if angle is not None else None
    return h, un


def initial_state(M, K, NT, wx, wy, t=0.0):
    """h, u, tracers (padded with the sentinel row) on an oracle.Mesh M at time t."""
    h = np.zeros((M.NCellsSize, K))
    u = np.zeros((M.NEdgesSize, K))
    nC, nE = M.NCellsAll, M.NEdgesAll
    h[:nC] = exact(M.XCell[:nC], M.YCell[:nC], None, t, wx, wy)[0][:, None]
    u[:nE] = exact(M.XEdge[:nE], M.YEdge[:nE], M.AngleEdge[:nE], t, wx, wy)[1][:, None]
    tr = np.zeros((max(NT, 1), M.NCellsSize, K))
    tr[:, :nC] = 1.0
    return h, u, tr


def l2_error_h(M, h, t, wx, wy):
    nC = M.NCellsOwned
    ex = exact(M.XCell[:nC], M.YCell[:nC], None, t, wx, wy)[0]
    a = M.AreaCell[:nC]
    return float(np.sqrt((a * (h[:nC, 0] - ex) ** 2).sum() / a.sum()))
