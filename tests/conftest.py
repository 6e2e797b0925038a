import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import oracle as O
    O.build()


def pytest_collection_modifyitems(config, items):
    """Files whose tests start child processes that use the GPU (multi-rank runs, re-runs of the parity suite under
    other kernel structures, the compiled boundary programs) come first, in file-name order, whatever order or
    selection pytest was given: the test runner's own process then has not touched the GPU while they run."""
    first = ("test_00_multirank_gpu.py", "test_01_boundary_programs.py")
    items.sort(key=lambda it: (0, first.index(it.fspath.basename)) if it.fspath.basename in first else (1, 0))
