"""Run by tests/test_gpu_dist.py in a fresh interpreter (torch initialises its HIP runtime BEFORE libosfm_mi355.so is loaded, the order
bench.py uses): a one-rank RCCL process group goes through exactly the calls the N-rank job makes in its exchange step."""
import os
import socket
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from opensfm_amd import dist as odist
    from opensfm_amd import matching, synthetic

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
        torch.cuda.synchronize()
        store.close()
    finally:
        dist.destroy_process_group()
    print("RCCL exchange step OK: %d pairs, %d matches" % (len(pairs), int(counts.sum())))


if __name__ == "__main__":
    main()
