"""Boundary proof: two native consumers of the product's boundaries run on the GPU box as child processes --
tests/native/boundary_test.cpp (hipcc, against omega_amd/csrc/*.h: the reference's class / method names and
registries, custom tendency as a std::function with its own HIP kernel, TimeStepperTest's orders 4/1/2) and
tests/native/capi_test.c (gcc -std=c99 against include/omega_amd.h).  Both read the mesh from an MPAS-convention
file and compare the fused RHS and an RK4 step with the committed golden vectors bit for bit.

The file name sorts before the in-process GPU tests on purpose (child programs start before this process touches
the GPU; see tests/test_00_multirank_gpu.py)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")


def build_native():
    subprocess.check_call(["make", "-C", NATIVE, "-s"])
    return os.path.join(NATIVE, "build", "boundary_test"), os.path.join(NATIVE, "build", "capi_test")


def test_native_consumers_compile_against_the_headers():
    """hipcc against omega_amd/csrc/*.h (C++17) and gcc -std=c99 -Wall -Wextra against include/omega_amd.h."""
    cpp, c = build_native()
    assert os.access(cpp, os.X_OK) and os.access(c, os.X_OK)
    out = subprocess.run(["ldd", c], capture_output=True, text=True).stdout
    assert "libomega_amd.so" in out and "oracle" not in out


def _write_case(tmp_path, name):
    from omega_amd.meshgen import synthetic_state
    from tests.golden.make_golden import CASES
    from tests.test_mesh_file import write_scipy
    ref = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    make, K, NT = CASES[name]
    g = make()
    write_scipy(str(tmp_path / "mesh.nc"), g, 2, K=K)
    hg, ug, trg = synthetic_state(g, K, NT)
    for fn, arr in (("h", hg), ("u", ug), ("tr", trg)):
        np.ascontiguousarray(arr, dtype="<f8").tofile(str(tmp_path / f"{fn}.bin"))
    for key in ("hTend", "uTend", "trTend", "rk4_h", "rk4_u", "rk4_tr"):
        a = ref[key]
        a = a[..., :-1, :]          # golden arrays carry the sentinel row
        np.ascontiguousarray(a[:NT] if a.ndim == 3 else a, dtype="<f8").tofile(str(tmp_path / f"{key}.bin"))
    return K, NT


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["planar16x16_k4_nt1", "ico2_k6_nt2"])
def test_cpp_consumer_of_the_class_boundary(tmp_path, name):
    cpp, _ = build_native()
    K, NT = _write_case(tmp_path, name)
    r = subprocess.run([cpp, str(tmp_path / "mesh.nc"), str(tmp_path), str(K), str(NT)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "boundary_test OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    for scheme in ("RungeKutta4", "Forward-Backward", "RungeKutta2"):
        assert scheme in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["planar16x16_k4_nt1", "ico2_k6_nt2"])
def test_c_consumer_of_the_c_abi(tmp_path, name):
    _, c = build_native()
    K, NT = _write_case(tmp_path, name)
    r = subprocess.run([c, str(tmp_path / "mesh.nc"), str(tmp_path), str(K), str(NT)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "capi_test OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
