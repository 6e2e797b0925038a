"""One-off randomised multi-rank runs of tests/mp_worker.py on ONE GPU (not part of the suite): world size, mesh family,
levels, tracers, partitioner, local numbering, wire, overlapped / sequential, scheme and halo width drawn at random.
HaloWidth 4 cases must equal the SINGLE-RANK oracle on owned elements, HaloWidth 3 cases the PARTITIONED oracle on every
local element (--against-partitioned).  One case at a time, at most 5 ranks on the card.
   python tools/fuzz_ranks.py <n cases> <seed>            (through gpurun; prints the failing command lines)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_multirank_cpu import run_ranks  # noqa: E402

# (mesh arguments, largest world size that still leaves every rank interior cells behind its band: the worker asserts that
# the overlapped stages really have a band AND an interior launch)
MESHES = [(["--nx", 48, "--ny", 24], 3), (["--nx", 64, "--ny", 32], 4), (["--nx", 40, "--ny", 40], 3), (["--nx", 96, "--ny", 64], 5),
          (["--mesh", "ico4"], 3), (["--mesh", "fib1500"], 2), (["--mesh", "hex48x24_coast_mixed"], 2), (["--mesh", "ico4_coast_lakes"], 3),
          (["--mesh", "fib1500_coast_ragged"], 2), (["--mesh", "hex48x24_perm5"], 2), (["--mesh", "ico5"], 5)]


def case(rng):
    mesh, max_world = rng.choice(MESHES)
    world = rng.randint(2, max_world)
    halo3 = rng.random() < 0.5
    stepper = rng.choice(["RungeKutta4"] * 4 + ["RungeKutta2", "Forward-Backward"])
    a = [*mesh, "--levels", rng.choice([3, 4, 5, 6, 16, 20, 32]), "--tracers", rng.choice([0, 1, 2, 3, 6]),
         "--steps", rng.choice([1, 2, 3]), "--stepper", stepper, "--partition", rng.choice(["rcb", "graph"]),
         "--local-order", rng.choice(["global", "curve", "hilbert", "kd"]), "--wire", rng.choice(["gloo", "ipc"])]
    if rng.random() < 0.5:
        a += ["--eddy-diff4", 1.0e11]
    if stepper == "RungeKutta4" and rng.random() < 0.4:
        a += ["--no-overlap"]
    if rng.random() < 0.3:
        a += ["--user-stream"]
    if halo3:
        a += ["--halo-width", 3, "--against-partitioned"]
    elif stepper == "Forward-Backward" and rng.random() < 0.5:
        a += ["--halo-width", 3, "--no-del4"]      # one evaluation per exchange, radius 1: partition independent at 3
    else:
        a += ["--halo-width", rng.choice([4, 4, 5])]
    opts = rng.choice(["", "", "SendBand=0,BandOnComm=0,ShrinkSweeps=0", "BandOnComm=0", "MergeL1=0,Pair=0", "TracerPatch=0"])
    return world, a, opts


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed)
    bad = 0
    for i in range(n):
        world, a, opts = case(rng)
        if opts:
            os.environ["OMEGA_AMD_OPTIONS"] = opts
        else:
            os.environ.pop("OMEGA_AMD_OPTIONS", None)
        try:
            outs = run_ranks("gpu", world, a, timeout=600)
            ok = all("OK" in o for o in outs)
        except AssertionError as e:
            ok = False
            print(str(e)[-1500:], flush=True)
        if not ok:
            bad += 1
            print("FAIL", i, world, opts, " ".join(map(str, a)), flush=True)
        if i % 10 == 9:
            print(f"[fuzz_ranks] {i + 1} cases, {bad} failed", flush=True)
    print(f"[fuzz_ranks] done: {n} cases, seed {seed}, {bad} failed", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
