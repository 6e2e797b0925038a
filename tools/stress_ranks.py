"""Longer multi-rank runs of tests/mp_worker.py on one GPU (one-off stress, not part of the test suite):
   python tools/stress_ranks.py <world> [worker args ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_multirank_cpu import run_ranks  # noqa: E402

if __name__ == "__main__":
    world = int(sys.argv[1])
    outs = run_ranks("gpu", world, sys.argv[2:], timeout=900)
    for o in outs:
        print(o.strip().splitlines()[-1])
