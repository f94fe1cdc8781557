import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is torch on the CPU, and half of the GPU suite's wall time is oracle time.  On the GPU box's multi-tenant host torch
    # takes all 128 hardware threads and runs the oracle 4 - 5x SLOWER than on 32 (bench.py's cpu_baseline: 13 - 30 s against 3.4 s per
    # training step): cap the pool.  (Sums change in the last bits with the thread count; every tolerance here is far above that.)
    import torch
    if torch.get_num_threads() > 32:
        torch.set_num_threads(32)


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
