import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The oracle's OpenMP loops on the GPU box's 256 hardware threads: test problems are small, and a parallel region per column of a
# 400-row factorisation with 256 spinning threads once stalled a GPU test run for its whole time limit (round 5, first call).  The tests
# do not measure the oracle (bench.py does, with every core): cap its team unless the caller has chosen.
os.environ.setdefault("OMP_NUM_THREADS", "32")
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # idle OpenMP threads sleep: a GPU call after an oracle call keeps its host core


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    from opensfm_amd import _lib

    return _lib.default_context(0)
