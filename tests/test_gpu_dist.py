"""The exchange step of the multi-GPU path on real RCCL: a one-rank process group on the GPU box goes through exactly the calls the
N-rank job makes (page-locked staging, all_gather_into_tensor between device buffers); the N-rank logic itself is covered by the
world-size-2 gloo tests (tests/test_dist_cpu.py).  Runs in a fresh interpreter: torch has to bring up its HIP runtime before the
library is loaded (as in bench.py), which an earlier test of the same pytest process would already have prevented."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_all_gather_match_graph_on_rccl():
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_rccl_check.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL exchange step OK" in r.stdout
