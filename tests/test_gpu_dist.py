"""The exchange step of the multi-GPU path on real RCCL: a one-rank process group on the GPU box goes through exactly the calls the
N-rank job makes (page-locked staging, all_gather_into_tensor between device buffers); the N-rank logic itself is covered by the
world-size-2 gloo tests (tests/test_dist_cpu.py)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_all_gather_match_graph_on_rccl(gpu_ctx):
    import torch
    import torch.distributed as dist

    from opensfm_amd import dist as odist
    from opensfm_amd import matching, synthetic

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        sc = synthetic.make_matching_scene(12, 500, seed=3)
        pairs = synthetic.all_pairs(12)
        store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets)
        counts, m = matching.match_pairs(store, pairs)
        for reorder in (False, True):
            for _ in range(2):  # the second round reuses the page-locked staging buffers
                cg, mg = odist.all_gather_match_graph(counts, m, len(pairs), 0, 1, 0, reorder=reorder, force_collective=True)
                assert np.array_equal(cg, counts) and np.array_equal(mg, m)
        assert counts.sum() > 500
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)
        dist.barrier()
        store.close()
    finally:
        dist.destroy_process_group()
