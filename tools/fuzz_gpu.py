"""One-off extension of tests/test_gpu_parity.py::test_randomised_configurations: more seeds / tracer counts than the suite
carries.    python tools/fuzz_gpu.py <n cases> <seed> [tracer counts ...]      (on a GPU box; prints the failing cases)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import omega_amd as oa  # noqa: E402
from tests import test_gpu_parity as T  # noqa: E402


def main():
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    counts = tuple(int(x) for x in sys.argv[3:]) or (0, 1, 2, 3, 4, 5, 6, 7, 9, 12)
    oa.device_init(0)
    bad = 0
    for i, case in enumerate(T._random_cases(n, seed=seed, tracer_counts=counts)):
        try:
            T.test_randomised_configurations.__wrapped__(case) if hasattr(T.test_randomised_configurations, "__wrapped__") \
                else T.test_randomised_configurations(case)
        except AssertionError as e:
            bad += 1
            print("FAIL", i, case, str(e)[:200], flush=True)
        if i % 50 == 49:
            print(f"[fuzz] {i + 1} cases, {bad} failed", flush=True)
    print(f"[fuzz] done: {n} cases, seed {seed}, {bad} failed", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
