"""Multi-GPU sharding of the pair-matching path (one process per GPU, RCCL over xGMI).

The reference's only parallelism on this path is a thread pool over pairs
(``opensfm/matching.py:83`` -> ``context.parallel_map``, ``context.py:47-67``): pairs are independent
units.  Here the pair list is dealt block-cyclically to the ranks, every rank holds the whole
descriptor store (10k images x 2k x 128 B = 2.56 GB, nothing next to 288 GB of HBM), and the only
collective is the exchange step at the end: an all-gather of per-pair match counts followed by an
all-gather of the concatenated (i, j) lists, after which every rank holds the identical global
match graph (what ``match_images_with_pairs`` returns).  ``backend="nccl"`` is RCCL on ROCm; the
same code runs over gloo on CPU tensors for the world-size-2 tests.
"""
from __future__ import annotations

from functools import lru_cache
from typing import List, Optional, Tuple

import numpy as np

BLOCK = 4096  # pairs per dealt block: consecutive pairs share their first image (L2 locality)


def shard_indices(n_pairs: int, rank: int, world: int, block: int = BLOCK) -> np.ndarray:
    idx = np.arange(n_pairs, dtype=np.int64)
    if world == 1:
        return idx
    return idx[(idx // block) % world == rank]


@lru_cache(maxsize=8)
def _all_shards(n_pairs: int, world: int, block: int) -> List[np.ndarray]:
    return [shard_indices(n_pairs, r, world, block) for r in range(world)]


def gathered_pair_order(n_pairs: int, world: int, block: int = BLOCK) -> np.ndarray:
    """Order of the pairs in the graph ``all_gather_match_graph(..., reorder=False)`` returns: rank 0's
    shard, then rank 1's, ... (``pairs[gathered_pair_order(...)]`` is the matching pair list)."""
    return np.concatenate(_all_shards(n_pairs, world, block)) if world > 1 else np.arange(n_pairs, dtype=np.int64)


def shard_pairs(pairs: np.ndarray, rank: int, world: int, block: int = BLOCK) -> np.ndarray:
    return np.ascontiguousarray(pairs[shard_indices(len(pairs), rank, world, block)])


_PINNED = {}  # (tag, dtype) -> pinned host tensor, grown on demand: page-locked staging for the exchange step


def _pinned(tag: str, n: int, dtype):
    import torch

    t = _PINNED.get((tag, dtype))
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1) + max(n, 1) // 4, dtype=dtype, pin_memory=True)
        _PINNED[(tag, dtype)] = t
    return t[:n]


def _all_gather_padded(local: np.ndarray, maxlen: int, world: int, dev, on_gpu: bool, tag: str) -> np.ndarray:
    """all-gather of one int32 vector per rank, padded to ``maxlen``; returns a (world, maxlen) host array.
    RCCL: one ``all_gather_into_tensor`` between device buffers, page-locked staging on both sides (one H2D of the local shard,
    one D2H of the gathered graph); gloo (CPU tests): the list form on host tensors."""
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local, np.int32).reshape(-1)
    if on_gpu:
        stage = _pinned(tag + "_in", maxlen, torch.int32)
        stage[: local.size].copy_(torch.from_numpy(local))
        if local.size < maxlen:
            stage[local.size:].zero_()
        send = torch.empty(maxlen, dtype=torch.int32, device=dev)
        send.copy_(stage, non_blocking=True)
        recv = torch.empty(world * maxlen, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(recv, send)
        out = _pinned(tag + "_out", world * maxlen, torch.int32)
        out.copy_(recv, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        return out.numpy().reshape(world, maxlen)
    send = torch.zeros(maxlen, dtype=torch.int32)
    send[: local.size] = torch.from_numpy(local)
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send)
    return np.stack([t.numpy() for t in parts])


def _assemble(allc: np.ndarray, allm: np.ndarray, idx, totals, n_pairs: int, world: int, reorder: bool):
    """(world, maxn) counts + (world, 2 maxm) match rows -> the global graph, rank-major or in the original pair order"""
    if not reorder:
        counts_g = np.concatenate([allc[r][: len(idx[r])] for r in range(world)])
        matches_g = np.concatenate([allm[r][: 2 * totals[r]] for r in range(world)]).reshape(-1, 2)
        return counts_g, matches_g
    counts_g = np.zeros(n_pairs, np.int32)
    for r in range(world):
        counts_g[idx[r]] = allc[r][: len(idx[r])]
    goff = np.concatenate([[0], np.cumsum(counts_g, dtype=np.int64)])
    matches_g = np.zeros((int(goff[-1]), 2), np.int32)
    for r in range(world):
        cr = allc[r][: len(idx[r])].astype(np.int64)
        tot = int(cr.sum())
        if tot == 0:
            continue
        roff = np.concatenate([[0], np.cumsum(cr)])[:-1]
        dest = np.repeat(goff[idx[r]] - roff, cr) + np.arange(tot)
        matches_g[dest] = allm[r][: 2 * tot].reshape(-1, 2)
    return counts_g, matches_g


@lru_cache(maxsize=8)
def _gather_plan(n_pairs: int, world: int, block: int, dev_str: str):
    """For the original-pair-order result: ``slot[p]`` = position of pair p in the gathered (world, maxn) count buffer, as a tensor on
    the exchange device (built once per pair-list shape: n_pairs int64)."""
    import torch

    idx = _all_shards(n_pairs, world, block)
    maxn = max(1, max(len(i) for i in idx))
    slot = np.empty(n_pairs, np.int64)
    for r in range(world):
        slot[idx[r]] = r * maxn + np.arange(len(idx[r]), dtype=np.int64)
    return torch.from_numpy(slot).to(torch.device(dev_str))


def _host_result(n: int, dtype, on_gpu: bool):
    """a fresh host tensor for a result of the exchange step: page-locked on a GPU box (torch's caching host allocator hands the block of
    the previous step back once its numpy view is gone, so steady-state steps do not pay for hipHostMalloc); never shared between calls"""
    import torch

    return torch.empty(max(n, 1), dtype=dtype, pin_memory=on_gpu)


class PendingGraph:
    """The global match graph of an exchange step whose copy to the host is still in flight (``all_gather_match_graph_device(...,
    defer_host_copy=True)``): the collectives and the device-side layout are done, the two D2H copies into page-locked memory were queued on
    a side stream behind them, and the caller goes on -- to the matching of the next chunk or step, whose kernels run on the library's own
    stream while the copy engine drains.  ``wait()`` blocks until the copies have landed and returns (counts, matches) as numpy views."""

    def __init__(self, counts_h, matches_h, n_pairs: int, total: int, event=None, keep=()):
        self._c, self._m, self._n, self._t, self._ev, self._keep = counts_h, matches_h, n_pairs, total, event, keep

    def wait(self) -> Tuple[np.ndarray, np.ndarray]:
        if self._ev is not None:
            self._ev.synchronize()
            self._ev, self._keep = None, ()
        return self._c.numpy()[: self._n], self._m.numpy()[: 2 * self._t].reshape(-1, 2)


_COPY_STREAMS = {}  # device -> the side stream of the deferred host copies


def all_gather_match_graph_device(graph, n_pairs: int, rank: int, world: int, local_rank: Optional[int] = None, block: int = BLOCK,
                                  reorder: bool = True, force_collective: bool = False, emulate_world: int = 0,
                                  timings: Optional[dict] = None, defer_host_copy: bool = False):
    """The exchange step straight from HBM: ``graph`` is the ``DeviceMatchGraph`` of this rank's shard
    (``matching.match_pairs(..., keep_device=True)``).  The counts and the match rows are all-gathered
    (``all_gather_into_tensor``: RCCL over xGMI between the buffers the kernels wrote -- no D2H / H2D hop of the shard), the global
    graph is put into its FINAL layout on the device (rank-major: a compaction of the padded rows; original pair order: one gather
    through a cached permutation) and comes to the host with one copy per array into page-locked memory -- the host does no
    per-row work.  Same return value as ``all_gather_match_graph``.

    The body does not depend on the backend: under gloo (the world-size-2 CPU test) ``graph`` hands out CPU tensors and the same
    lines run.  ``emulate_world = E`` (one rank only, measurement): after the real one-rank collective the receive buffers are filled
    with E copies of this rank's payload, so everything downstream of the collective -- compaction, D2H -- runs at the size an
    E-rank job has; ``n_pairs`` is then the E-rank list's length and ``graph`` holds rank 0's shard of it.  ``timings`` receives
    ``collective_ms``, ``layout_ms``, ``d2h_ms`` and ``bytes_to_host``.

    ``defer_host_copy`` (round 6): the step is not serial any more.  The D2H of the gathered graph -- 222 MB at eight ranks of configs[1], 3.9 of
    the exchange step's 4.5 ms -- is queued on a side stream and a ``PendingGraph`` comes back at once; the caller matches its next chunk
    (or step) and calls ``wait()`` when it needs the rows.  What stays exposed is the two collectives and the compaction (0.6 ms)."""
    if world == 1 and not force_collective and not emulate_world:
        if defer_host_copy:
            import torch

            c, m = graph.fetch()
            return PendingGraph(torch.from_numpy(np.ascontiguousarray(c)), torch.from_numpy(np.ascontiguousarray(m).reshape(-1)), len(c), len(m))
        return graph.fetch()
    import time

    import torch
    import torch.distributed as dist

    assert not emulate_world or world == 1, "emulate_world is a one-rank measurement"
    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", local_rank if local_rank is not None else rank) if on_gpu else torch.device("cpu")
    W = emulate_world or world  # ranks the layout is built for

    def sync():
        if on_gpu:
            torch.cuda.current_stream(dev).synchronize()

    t0 = time.perf_counter()
    idx = _all_shards(n_pairs, W, block) if W > 1 else [np.arange(n_pairs, dtype=np.int64)]
    lens = [len(i) for i in idx]
    maxn = max(1, max(lens))
    assert graph.n_pairs == lens[rank], (graph.n_pairs, lens[rank])
    # counts: shards differ by at most one block, so the send buffer is the device counts padded with zeros
    send = torch.zeros(maxn, dtype=torch.int32, device=dev)
    send[: graph.n_pairs].copy_(graph.counts_tensor())
    recv = torch.empty(world * maxn, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send)
    if emulate_world:
        recv = recv.repeat(W)
    recv2 = recv.view(W, maxn)
    # per-rank totals (rows past a shard's length are padding): the one value the host needs before it can size the second collective
    valid = torch.arange(maxn, device=dev)[None, :] < torch.tensor(lens, device=dev)[:, None]
    totals = [int(v) for v in (recv2 * valid).sum(1, dtype=torch.int64).tolist()]
    assert totals[rank] == graph.total, (totals[rank], graph.total)
    maxm = max(1, max(totals))
    sendm = torch.empty(2 * maxm, dtype=torch.int32, device=dev)
    sendm[: 2 * graph.total].copy_(graph.matches_tensor())  # D2D; the tail is never read
    recvm = torch.empty(world * 2 * maxm, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recvm, sendm)
    if emulate_world:
        recvm = recvm.repeat(W)
    sync()
    t1 = time.perf_counter()
    # ---- final layout on the device ----
    total = int(sum(totals))
    rows = recvm.view(W, maxm, 2)
    if not reorder:
        counts_d = torch.cat([recv2[r, : lens[r]] for r in range(W)]) if W > 1 else recv2[0, : lens[0]]
        matches_d = torch.cat([rows[r, : totals[r]] for r in range(W)]) if W > 1 else rows[0, : totals[0]]
    else:
        slot = _gather_plan(n_pairs, W, block, str(dev)) if W > 1 else torch.arange(n_pairs, device=dev)
        counts_d = recv[slot]  # counts in the original pair order
        c64 = counts_d.to(torch.int64)
        # first row of every pair inside its rank's padded block: the exclusive scan of the rank's own counts
        rscan = torch.cumsum(recv2.to(torch.int64), 1) - recv2
        src0 = (rscan + (torch.arange(W, device=dev) * maxm)[:, None]).view(-1)[slot]
        goff = torch.cumsum(c64, 0) - c64
        # row d of the result belongs to pair p(d); it is row src0[p] + (d - goff[p]) of the padded buffer
        pair_of_row = torch.repeat_interleave(torch.arange(n_pairs, device=dev), c64, output_size=total)
        src = (src0 - goff)[pair_of_row] + torch.arange(total, device=dev)
        matches_d = rows.view(-1, 2)[src]
    sync()
    t2 = time.perf_counter()
    # ---- one D2H per array, into page-locked memory ----
    counts_h = _host_result(n_pairs, torch.int32, on_gpu)
    matches_h = _host_result(2 * total, torch.int32, on_gpu)
    if defer_host_copy and on_gpu:
        # the copies go to a side stream behind the layout; the device tensors stay alive in the handle (and are marked as used by that stream)
        side = _COPY_STREAMS.get(str(dev))
        if side is None:
            side = _COPY_STREAMS[str(dev)] = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        cflat, mflat = counts_d.reshape(-1), matches_d.reshape(-1)
        with torch.cuda.stream(side):
            counts_h[:n_pairs].copy_(cflat, non_blocking=True)
            matches_h[: 2 * total].copy_(mflat, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        cflat.record_stream(side)
        mflat.record_stream(side)
        if timings is not None:
            timings.update(collective_ms=1e3 * (t1 - t0), layout_ms=1e3 * (t2 - t1), d2h_ms=0.0, bytes_to_host=4 * n_pairs + 8 * total, ranks=W, deferred=True)
        return PendingGraph(counts_h, matches_h, n_pairs, total, ev, (cflat, mflat))
    counts_h[:n_pairs].copy_(counts_d.reshape(-1), non_blocking=on_gpu)
    matches_h[: 2 * total].copy_(matches_d.reshape(-1), non_blocking=on_gpu)
    sync()
    t3 = time.perf_counter()
    if timings is not None:
        timings.update(collective_ms=1e3 * (t1 - t0), layout_ms=1e3 * (t2 - t1), d2h_ms=1e3 * (t3 - t2),
                       bytes_to_host=4 * n_pairs + 8 * total, ranks=W)
    if defer_host_copy:  # (gloo: the copies above were synchronous)
        return PendingGraph(counts_h, matches_h, n_pairs, total)
    return counts_h.numpy()[:n_pairs], matches_h.numpy()[: 2 * total].reshape(-1, 2)


def all_gather_match_graph(counts: np.ndarray, matches: np.ndarray, n_pairs: int, rank: int, world: int,
                           local_rank: Optional[int] = None, block: int = BLOCK, reorder: bool = True,
                           force_collective: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Every rank contributes (counts, matches) of its shard; every rank returns the global
    (counts[n_pairs], matches[total, 2]): in the order of the original pair list (``reorder=True``), or
    rank-major -- shard after shard, the order ``gathered_pair_order`` describes -- which needs no
    scatter of the tens of millions of match rows on the host (``reorder=False``).
    ``force_collective``: go through the collectives even with one rank (exercises the RCCL calls on a single GPU)."""
    if world == 1 and not force_collective:
        return counts, matches
    import torch
    import torch.distributed as dist

    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", local_rank if local_rank is not None else rank) if on_gpu else torch.device("cpu")
    idx = _all_shards(n_pairs, world, block) if world > 1 else [np.arange(n_pairs, dtype=np.int64)]
    maxn = max(1, max(len(i) for i in idx))
    allc = _all_gather_padded(counts, maxn, world, dev, on_gpu, "counts").copy()  # the staging buffer is reused below
    totals = [int(allc[r][: len(idx[r])].sum()) for r in range(world)]
    maxm = max(1, max(totals))
    allm = _all_gather_padded(matches, 2 * maxm, world, dev, on_gpu, "matches")
    return _assemble(allc, allm, idx, totals, n_pairs, world, reorder)
