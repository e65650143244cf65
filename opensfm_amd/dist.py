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


def all_gather_match_graph(counts: np.ndarray, matches: np.ndarray, n_pairs: int, rank: int, world: int,
                           local_rank: Optional[int] = None, block: int = BLOCK, reorder: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """Every rank contributes (counts, matches) of its shard; every rank returns the global
    (counts[n_pairs], matches[total, 2]): in the order of the original pair list (``reorder=True``), or
    rank-major -- shard after shard, the order ``gathered_pair_order`` describes -- which needs no
    scatter of the tens of millions of match rows on the host (``reorder=False``)."""
    if world == 1:
        return counts, matches
    import torch
    import torch.distributed as dist

    on_gpu = dist.get_backend() == "nccl"
    dev = torch.device("cuda", local_rank if local_rank is not None else rank) if on_gpu else torch.device("cpu")
    idx = _all_shards(n_pairs, world, block)
    maxn = max(len(i) for i in idx)
    c = torch.zeros(maxn, dtype=torch.int32)
    c[: len(counts)] = torch.from_numpy(np.ascontiguousarray(counts, np.int32))
    c = c.to(dev)
    allc = [torch.empty_like(c) for _ in range(world)]
    dist.all_gather(allc, c)
    allc = [t.cpu().numpy() for t in allc]
    totals = [int(allc[r][: len(idx[r])].sum()) for r in range(world)]
    maxm = max(1, max(totals))
    m = torch.zeros(maxm * 2, dtype=torch.int32)
    flat = np.ascontiguousarray(matches, np.int32).reshape(-1)
    m[: flat.size] = torch.from_numpy(flat)
    m = m.to(dev)
    allm = [torch.empty_like(m) for _ in range(world)]
    dist.all_gather(allm, m)
    if not reorder:
        counts_g = np.concatenate([allc[r][: len(idx[r])] for r in range(world)])
        hm = [allm[r].cpu().numpy() for r in range(world)]
        matches_g = np.concatenate([hm[r][: 2 * totals[r]] for r in range(world)]).reshape(-1, 2)
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
        matches_g[dest] = allm[r].cpu().numpy()[: 2 * tot].reshape(-1, 2)
    return counts_g, matches_g
