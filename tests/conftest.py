import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))          # (helpers: parity_util, oracle_backend)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The degree-class posttrans route (dgn_amd/ops.py: DC_POSTTRANS) is taken from 16 384 nodes on by default; the parity fixtures and the
# oracle-sized batches are smaller, so the tests lower the threshold: every simple / complex layer test below runs THAT route against
# the oracle (tests/test_dc_hip.py compares it with the folded route as well).
os.environ.setdefault("DGN_DC_MIN_NODES", "0")
# Likewise the block backward of the sweep (csrc/dgn_agg_block.hpp: from 131 072 nodes on by default): the oracle-sized molecule batches
# run it (tests/test_block_backward_gpu.py compares it with the staged scatter; test_shipped_configs_gpu.py has every layer type on the
# default routes).
os.environ.setdefault("DGN_BLK_MIN_NODES", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load
