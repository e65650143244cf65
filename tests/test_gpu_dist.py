"""The exchange step of the multi-GPU path on real RCCL.  One-rank test: a one-rank process group on the GPU box goes through exactly
the calls the N-rank job makes (device-resident results, all_gather_into_tensor between device buffers, page-locked staging).  Two-rank
test: runs where the box has two GPUs (skips on the 1-GPU box) -- two processes under torch.distributed.run, one GPU each, shard +
all-gather over RCCL, every rank ends with the single-GPU graph.  The N-rank host logic itself is covered by the world-size-2 gloo
tests (tests/test_dist_cpu.py).  Fresh interpreters: torch has to bring up its HIP runtime before the library is loaded (as in
bench.py), which an earlier test of the same pytest process would already have prevented."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _gpu_count() -> int:
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
    return int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0


def test_all_gather_match_graph_on_rccl():
    r = subprocess.run([sys.executable, os.path.join(HERE, "dist_rccl_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL exchange step OK: world 1" in r.stdout


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(n, script, *args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                           "--master-port", str(_free_port()), script, *args], capture_output=True, text=True, timeout=900, env=env)


def test_two_ranks_shard_and_all_gather_over_rccl():
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs (the N-rank logic runs on gloo in tests/test_dist_cpu.py)")
    r = _torchrun(2, os.path.join(HERE, "dist_rccl_check.py"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL exchange step OK: world 2" in r.stdout


def test_bench_two_gpus_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself and labels the line n_gpus 2"""
    if _gpu_count() < 2:
        pytest.skip("needs two GPUs")
    import json

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--images", "120", "--features", "600", "--steps", "1",
                        "--warmup", "1", "--headline-only", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["pairs_per_rank"] < line["config"]["pairs"]
