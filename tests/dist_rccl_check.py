"""Run by tests/test_gpu_dist.py in a fresh interpreter (torch initialises its HIP runtime BEFORE libosfm_mi355.so is loaded, the order
bench.py uses).  One rank (plain `python dist_rccl_check.py`): a one-rank RCCL process group goes through exactly the calls the N-rank
job makes in its exchange step.  N ranks (`python -m torch.distributed.run --nproc-per-node N dist_rccl_check.py`, one GPU each): every
rank matches its block-cyclic shard of the pair list, the match graph is all-gathered over RCCL from the device-resident results, and
every rank must hold the graph a single-GPU run of the whole list produces (the fan-in of opensfm/matching.py:83-98)."""
import os
import socket
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world == 1:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from opensfm_amd import dist as odist
    from opensfm_amd import matching, synthetic
    from opensfm_amd._lib import default_context

    try:
        sc = synthetic.make_matching_scene(12, 500, seed=3)
        pairs = synthetic.all_pairs(12)
        store = matching.DescriptorStore.from_packed(sc.desc, sc.pts, sc.offsets, default_context(local_rank))
        counts, m = matching.match_pairs(store, pairs)  # the whole list on this GPU: what every rank must end up with
        assert counts.sum() > 500
        block = 8  # several blocks per rank on a 66-pair list
        mine = odist.shard_pairs(pairs, rank, world, block)
        order = odist.gathered_pair_order(len(pairs), world, block)
        off = np.concatenate([[0], np.cumsum(counts)])
        want_rank_major = np.concatenate([m[off[p]: off[p + 1]] for p in order]).reshape(-1, 2)
        for reorder in (False, True):
            for _ in range(2):  # the second round reuses the page-locked staging buffers
                # (a) host variant: D2H of the shard, staged H2D, all-gather, D2H
                c_s, m_s = matching.match_pairs(store, mine)
                cg, mg = odist.all_gather_match_graph(c_s, m_s, len(pairs), rank, world, local_rank, block=block, reorder=reorder,
                                                      force_collective=True)
                # (b) device variant: the shard's rows never leave HBM before the collective
                g = matching.match_pairs(store, mine, keep_device=True)
                assert g.total == int(c_s.sum()) and np.array_equal(g.counts, c_s)
                fc, fm = g.fetch()
                assert np.array_equal(fc, c_s) and np.array_equal(fm, m_s)
                cd, md = odist.all_gather_match_graph_device(g, len(pairs), rank, world, local_rank, block=block, reorder=reorder,
                                                             force_collective=True)
                # (c) the deferred host copy (bench.py's N-rank step): the handle is waited for after another matching call has been queued
                h = odist.all_gather_match_graph_device(g, len(pairs), rank, world, local_rank, block=block, reorder=reorder, force_collective=True,
                                                        defer_host_copy=True)
                g2 = matching.match_pairs(store, mine, keep_device=True)
                ch, mh = h.wait()
                g2.close()
                assert np.array_equal(ch, cd) and np.array_equal(mh, md)
                g.close()
                if reorder:
                    assert np.array_equal(cg, counts) and np.array_equal(mg, m)
                else:
                    assert np.array_equal(cg, counts[order]) and np.array_equal(mg, want_rank_major)
                assert np.array_equal(cd, cg) and np.array_equal(md, mg)
        if world == 1:
            # the one-GPU measurement mode of bench.py (exchange_emulation): E copies of this rank's payload downstream of the collective
            g = matching.match_pairs(store, pairs, keep_device=True)
            for reorder in (False, True):
                tm = {}
                ce, me = odist.all_gather_match_graph_device(g, 3 * len(pairs), 0, 1, local_rank, block=len(pairs), reorder=reorder,
                                                             emulate_world=3, timings=tm)
                assert np.array_equal(ce, np.tile(counts, 3)) and np.array_equal(me, np.tile(m, (3, 1))) and tm["ranks"] == 3
            g.close()
        # an empty shard and an all-empty result go through the device path too
        g = matching.match_pairs(store, pairs[:0], keep_device=True)
        assert g.total == 0 and g.n_pairs == 0
        if world == 1:
            ce, me = odist.all_gather_match_graph_device(g, 0, 0, 1, local_rank, force_collective=True)
            assert len(ce) == 0 and me.shape == (0, 2)
        g.close()
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)
        assert int(t.item()) == world
        dist.barrier()
        torch.cuda.synchronize()
        store.close()
    finally:
        dist.destroy_process_group()
    if rank == 0:
        print("RCCL exchange step OK: world %d, %d pairs, %d matches" % (world, len(pairs), int(counts.sum())))


if __name__ == "__main__":
    main()
