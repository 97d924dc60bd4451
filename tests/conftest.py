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


# Batches up to 8 192 nodes take the graph-block layer route by default (dgn_amd/ops.py: BLOCK_LAYER_MAX_NODES) -- which is every oracle-
# sized batch of this suite.  tests/test_block_layer_gpu.py runs that route through the layers' default dispatch (and
# test_shipped_configs_gpu.py::test_every_layer_type_on_the_default_routes_vs_oracle at the library's defaults); every other module keeps
# testing the STREAMING kernels it was written for.
@pytest.fixture(autouse=True)
def _streaming_routes_outside_the_block_layer_tests(request, monkeypatch):
    if getattr(request.node, "module", None) is not None and request.node.module.__name__.endswith("test_block_layer_gpu"):
        # (that module tests the route's KERNELS at every shape, also those the dispatch hands to the streaming kernels because their
        #  posttrans is large -- ops.BLOCK_LAYER_MAX_POST, restored to the default in the test of the dispatch itself)
        try:
            import dgn_amd.ops as ops
            monkeypatch.setattr(ops, "BLOCK_LAYER_MAX_POST", 1 << 30)
        except Exception:
            pass
        yield
        return
    try:
        import dgn_amd.ops as ops
    except Exception:
        yield
        return
    monkeypatch.setattr(ops, "BLOCK_LAYER_MAX_NODES", 0)
    yield


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
