"""world_size-2 gloo test of the sharding + all-gather of the match graph (runs on CPU)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, seed, q, empty_rank=-1):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from opensfm_amd import dist as odist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(seed)
    counts_all = rng.integers(0, 7, n_pairs).astype(np.int32)
    counts_all[rng.random(n_pairs) < 0.5] = 0
    if empty_rank >= 0:  # one rank whose shard has pairs but not a single match
        counts_all[odist.shard_indices(n_pairs, empty_rank, world, block=16)] = 0
    off = np.concatenate([[0], np.cumsum(counts_all)])
    matches_all = rng.integers(0, 2000, (int(off[-1]), 2)).astype(np.int32)
    idx = odist.shard_indices(n_pairs, rank, world, block=16)
    mine = np.concatenate([matches_all[off[i]: off[i + 1]] for i in idx]) if len(idx) else np.zeros((0, 2), np.int32)
    c, m = odist.all_gather_match_graph(counts_all[idx], mine, n_pairs, rank, world, block=16)
    ok = np.array_equal(c, counts_all) and np.array_equal(m, matches_all)
    # rank-major variant: the same graph, pairs in gathered_pair_order
    c2, m2 = odist.all_gather_match_graph(counts_all[idx], mine, n_pairs, rank, world, block=16, reorder=False)
    order = odist.gathered_pair_order(n_pairs, world, block=16)
    want_m = np.concatenate([matches_all[off[i]: off[i + 1]] for i in order]) if n_pairs else np.zeros((0, 2), np.int32)
    ok = ok and np.array_equal(c2, counts_all[order]) and np.array_equal(m2, want_m.reshape(-1, 2))
    ok = ok and np.array_equal(np.sort(order), np.arange(n_pairs))
    # the DEVICE variant -- the function bench.py calls at N > 1 -- through the same process group: its body is backend-agnostic, the
    # shard is handed over as tensors exactly as DeviceMatchGraph hands over its HBM buffers
    import torch

    class HostGraph:
        def __init__(self, c, m):
            self.n_pairs, self.total = len(c), int(c.sum())
            self._c, self._m = torch.from_numpy(np.ascontiguousarray(c, np.int32)), torch.from_numpy(np.ascontiguousarray(m, np.int32).reshape(-1))

        def counts_tensor(self):
            return self._c

        def matches_tensor(self):
            return self._m

    tm = {}
    c3, m3 = odist.all_gather_match_graph_device(HostGraph(counts_all[idx], mine), n_pairs, rank, world, block=16, timings=tm)
    ok = ok and np.array_equal(c3, counts_all) and np.array_equal(m3, matches_all) and tm["ranks"] == world
    c4, m4 = odist.all_gather_match_graph_device(HostGraph(counts_all[idx], mine), n_pairs, rank, world, block=16, reorder=False)
    ok = ok and np.array_equal(c4, counts_all[order]) and np.array_equal(m4, want_m.reshape(-1, 2))
    ok = ok and tm["bytes_to_host"] == 4 * n_pairs + 8 * len(matches_all)
    # the deferred form bench.py's N-rank step uses (round 6): a handle comes back, wait() hands over the same graph
    h = odist.all_gather_match_graph_device(HostGraph(counts_all[idx], mine), n_pairs, rank, world, block=16, reorder=False, defer_host_copy=True)
    c5, m5 = h.wait()
    ok = ok and isinstance(h, odist.PendingGraph) and np.array_equal(c5, c4) and np.array_equal(m5, m4)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [1, 37, 1000])
def test_all_gather_match_graph_world2(n_pairs):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, 3, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.parametrize("n_pairs,empty_rank", [(100, 2), (16 * 8 * 3 + 5, -1)])
def test_all_gather_match_graph_world8(n_pairs, empty_rank):
    """the shape of the driver's 8-GPU run on gloo: 100 pairs in blocks of 16 = seven ranks with pairs (the last block short) and one
    rank with NONE, one of the others without a single match; 389 pairs = every rank busy, an uneven last block.  Both result orders,
    host and device variants (the worker checks all four)"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, n_pairs, 5, q, empty_rank)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(8)]


def test_shards_partition_the_pair_list():
    from opensfm_amd import dist as odist

    for world in (1, 2, 4, 8):
        idx = np.concatenate([odist.shard_indices(100000, r, world) for r in range(world)])
        assert np.array_equal(np.sort(idx), np.arange(100000))
        sizes = [len(odist.shard_indices(499500, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= odist.BLOCK


def _emulated_worker(port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from opensfm_amd import dist as odist

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    rng = np.random.default_rng(11)
    n, E = 53, 4
    c = rng.integers(0, 5, n).astype(np.int32)
    m = rng.integers(0, 999, (int(c.sum()), 2)).astype(np.int32)

    class HostGraph:
        n_pairs, total = n, int(c.sum())

        def counts_tensor(self):
            return torch.from_numpy(c)

        def matches_tensor(self):
            return torch.from_numpy(m.reshape(-1))

    ok = True
    for reorder in (False, True):
        # block = shard length: E contiguous shards, each a copy of this rank's payload
        cg, mg = odist.all_gather_match_graph_device(HostGraph(), E * n, 0, 1, block=n, reorder=reorder, emulate_world=E)
        ok = ok and np.array_equal(cg, np.tile(c, E)) and np.array_equal(mg, np.tile(m, (E, 1)))
    q.put(bool(ok))
    dist.destroy_process_group()


def test_emulated_world_fills_the_layout_of_an_n_rank_job():
    """bench.py's one-GPU measurement of the exchange step (`exchange_emulation`): one rank, the receive buffers hold E copies of its
    payload, everything after the collective runs at the E-rank size"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_emulated_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=120) is True
    p.join(timeout=60)
