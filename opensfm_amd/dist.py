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


def all_gather_match_graph_device(graph, n_pairs: int, rank: int, world: int, local_rank: Optional[int] = None, block: int = BLOCK,
                                  reorder: bool = True, force_collective: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """The exchange step straight from HBM: ``graph`` is the ``DeviceMatchGraph`` of this rank's shard
    (``matching.match_pairs(..., keep_device=True)``).  The counts and the match rows are all-gathered (RCCL,
    ``all_gather_into_tensor``) from the buffers the kernels wrote -- no D2H / H2D hop of the shard -- and the gathered graph comes
    to the host once, into page-locked memory.  Same return value as ``all_gather_match_graph``.  RCCL only (the gloo tests go
    through the host variant)."""
    if world == 1 and not force_collective:
        return graph.fetch()
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", local_rank if local_rank is not None else rank)
    idx = _all_shards(n_pairs, world, block) if world > 1 else [np.arange(n_pairs, dtype=np.int64)]
    maxn = max(1, max(len(i) for i in idx))
    assert graph.n_pairs == len(idx[rank]), (graph.n_pairs, len(idx[rank]))
    # counts: shards differ by at most one block, so the send buffer is the device counts padded with zeros
    send = torch.zeros(maxn, dtype=torch.int32, device=dev)
    send[: graph.n_pairs].copy_(graph.counts_tensor())
    recv = torch.empty(world * maxn, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send)
    allc = _pinned("counts_out", world * maxn, torch.int32)
    allc.copy_(recv, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    allc = allc.numpy().reshape(world, maxn).copy()
    totals = [int(allc[r][: len(idx[r])].sum()) for r in range(world)]
    assert totals[rank] == graph.total, (totals[rank], graph.total)
    maxm = max(1, max(totals))
    sendm = torch.empty(2 * maxm, dtype=torch.int32, device=dev)
    sendm[: 2 * graph.total].copy_(graph.matches_tensor())  # D2D; the tail is never read
    recvm = torch.empty(world * 2 * maxm, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recvm, sendm)
    allm = _pinned("matches_out", world * 2 * maxm, torch.int32)
    allm.copy_(recvm, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    return _assemble(allc, allm.numpy().reshape(world, 2 * maxm), idx, totals, n_pairs, world, reorder)


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
