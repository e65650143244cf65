#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X matching + bundle-adjustment hot path.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON
line on rank 0.  A "step" is one pass of the pair-matching hot path (descriptor distances, top-2,
Lowe ratio, mutual check, F-matrix RANSAC, gates) over the rank's shard of an exhaustive pair
list, followed by the all-gather of the match graph.

  metric  : BASELINE.json's "image-pairs matched/sec (+ BA LM-iters/sec)"
  value   : image pairs matched per second, whole job (all ranks), inputs resident in HBM
  N = 1   : BASELINE.json configs[1] -- 1 000 images x 2 000 x 128-D, all 499 500 pairs
  N > 1   : the image count grows so that every rank keeps ~499 500 pairs (weak scaling);
            descriptors are replicated, the pair list is dealt block-cyclically, the match graph
            is all-gathered over RCCL at the end of every step
  "ba"    : the second half of the metric (LM iterations / s of global BA), 1 GPU, rank 0 only
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_I8_TOPS = 5000.0  # dense int8 MFMA peak of MI355X (2x the 2.5 PFLOP/s bf16 dense peak); ubench ceiling 4404
FLOP_PER_PAIR = 2.0 * 2000 * 2000 * 128  # one distance matrix serves both directions


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=1000, help="images at N=1 (configs[1]: 1000)")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-pairs", type=int, default=3072)  # ~15 s of host work on 128 threads
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--no-tracks", action="store_true")
    ap.add_argument("--ba-shots", type=int, default=5000)
    ap.add_argument("--ba-points", type=int, default=500000)
    ap.add_argument("--ba-track", type=int, default=10)
    ap.add_argument("--ba-iters", type=int, default=10)
    ap.add_argument("--no-robust", action="store_true", help="descriptor stage only (debug)")
    return ap.parse_args()


def main():
    args = parse()
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)

    from opensfm_amd import matching, synthetic
    from opensfm_amd import dist as odist
    from opensfm_amd._lib import MatchTimings, default_context

    ctx = default_context(local_rank)
    # ---- workload: exhaustive pairs over n_images(N) images so that pairs ~= N * pairs(images) ----
    p1 = args.images * (args.images - 1) // 2
    n_images = args.images if world == 1 else int(math.ceil((1 + math.sqrt(1 + 8.0 * world * p1)) / 2))
    t0 = time.time()
    scene = synthetic.make_matching_scene(n_images, args.features, seed=args.seed)
    pairs_all = synthetic.all_pairs(n_images)
    pairs_all = pairs_all[: world * p1]
    my_pairs = odist.shard_pairs(pairs_all, rank, world)
    pairs_gathered = pairs_all[odist.gathered_pair_order(len(pairs_all), world)]  # pair list of the gathered graph
    store = matching.DescriptorStore.from_packed(scene.desc, scene.pts, scene.offsets, ctx)
    t_setup = time.time() - t0
    robust = not args.no_robust

    def step(tm=None):
        counts, m = matching.match_pairs(store, my_pairs, robust=robust, timings=tm)
        # rank-major gathered order: no host-side scatter of the match rows inside the timed region
        return odist.all_gather_match_graph(counts, m, len(pairs_all), rank, world, local_rank, reorder=False)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    tms = []
    barrier()
    t0 = time.perf_counter()
    graph = None
    for _ in range(args.steps):
        tm = MatchTimings()
        graph = step(tm)
        tms.append(tm)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_pairs = len(pairs_all) * args.steps
    value = total_pairs / elapsed

    # ---- roofline of the dominant kernel (fused distance/top-2 kernel), HIP events on its stream ----
    launches = sum(int(t.match_launches) for t in tms)
    ms_kernel = sum(float(t.ms_match_kernel) for t in tms)
    avg_ms = ms_kernel / max(1, launches)
    pairs_per_launch = len(my_pairs) * args.steps / max(1, launches)
    n_avg = float(np.mean(np.diff(scene.offsets)))
    flop_pair = 2.0 * n_avg * n_avg * 128
    achieved = flop_pair * pairs_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    roofline = {
        "bound": "mfma",
        "kernel": "match_fused4_kernel",
        "achieved": round(achieved, 2),
        "peak": PEAK_I8_TOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_I8_TOPS, 4),
        "traffic": None,  # PMC passes cannot run inside this process: see profiles/r01_final_match_pmc.txt (36.7 GB read / launch)
        "avg_launch_ms": round(avg_ms, 3),
        "launches": launches,
        "algorithmic_flop_per_pair": flop_pair,
    }

    out = {
        "metric": "image-pairs matched/sec (+ BA LM-iters/sec, 5k cams / 500k pts)",
        "value": round(value, 1),
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i8 (exact int32 accumulate) + f64 RANSAC",
        "data": "synthetic",
        "config": {
            "workload": f"{n_images} images x {args.features} x 128-D uint8-valued descriptors, exhaustive "
                        f"{len(pairs_all)} pairs, BRUTEFORCE symmetric + Lowe 0.8 + F-RANSAC(0.004, 0.9999), min 20",
            "images": n_images,
            "features_per_image": args.features,
            "pairs": int(len(pairs_all)),
            "pairs_per_rank": int(len(my_pairs)),
            "robust": robust,
            "parallelism": f"pair-sharded x{world}, descriptors replicated, all-gather of match graph",
        },
        "roofline": roofline,
        "stage_ms_per_step": {
            "match_kernel": round(ms_kernel / args.steps, 3),
            "ransac_kernel": round(sum(float(t.ms_ransac_kernel) for t in tms) / args.steps, 3),
            "call_total": round(sum(float(t.ms_total) for t in tms) / args.steps, 3),
        },
        "result": {
            "pairs_with_matches": int((graph[0] > 0).sum()) if graph is not None else 0,
            "total_inlier_matches": int(graph[0].sum()) if graph is not None else 0,
            "pairs_exact_path": int(sum(int(t.pairs_exact_path) for t in tms) / args.steps),
        },
        "setup_s": round(t_setup, 2),
    }

    if rank == 0:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, pairs_gathered, args.cpu_sample_pairs, graph)
        if not args.no_tracks and graph is not None:
            out["tracks"] = tracks_bench(ctx, scene, pairs_gathered, graph, not args.no_cpu_baseline)
        if not args.no_ba:
            try:
                import bench_ba as ba_bench

                out["ba"] = ba_bench.run(ctx, args.ba_shots, args.ba_points, args.ba_track, args.ba_iters,
                                         cpu_baseline=not args.no_cpu_baseline)
            except ImportError:
                out["ba"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def tracks_bench(ctx, scene, pairs_all, graph, with_cpu):
    """Next row after the hot path: link the all-gathered match graph into tracks
    (tracking.create_tracks_manager's union-find + _good_track) on the GPU, next to the CPU oracle."""
    from opensfm_amd import tracking

    counts, matches = graph
    ea, eb = tracking.edges_from_match_graph(pairs_all, counts, matches, scene.offsets)
    tracking.create_tracks_arrays(ea[:1000], eb[:1000], scene.offsets, 2, ctx)  # warm-up (hipCUB kernels)
    tm = {}
    t0 = time.perf_counter()
    nt, ot, oi, of = tracking.create_tracks_arrays(ea, eb, scene.offsets, 2, ctx, timings=tm)
    wall = time.perf_counter() - t0
    out = {
        "metric": "matches linked into tracks / s",
        "workload": f"{len(ea)} matches over {len(scene.offsets) - 1} images x {int(scene.offsets[-1])} features, min_track_length 2",
        "value": round(len(ea) / (tm["ms_device"] * 1e-3), 1),
        "unit": "matches/s",
        "device_ms": round(tm["ms_device"], 3),
        "call_ms_incl_h2d_d2h": round(1e3 * wall, 3),
        "tracks": int(nt),
        "observations": int(len(ot)),
    }
    if with_cpu:
        import oracle

        t0 = time.perf_counter()
        nt_o, ot_o, oi_o, of_o = oracle.tracks(ea, eb, scene.offsets, 2)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": round(len(ea) / dt, 1), "unit": "matches/s", "cores": 1, "kind": "port",
            "sample": f"the same {len(ea)} matches, {dt * 1e3:.1f} ms, sequential union-find in C (the reference is pure Python)",
            "parity": bool(nt_o == nt and np.array_equal(ot_o, ot) and np.array_equal(oi_o, oi) and np.array_equal(of_o, of)),
        }
    return out


def cpu_baseline(scene, pairs_all, n_sample, graph):
    """The CPU oracle (a port of the reference's cv2 path, see oracle/*.c) timed on this box's host
    cores on a bounded, strided sample of the same pair list; also re-checks parity on the sample."""
    import oracle

    n_sample = min(n_sample, len(pairs_all))
    sel = np.linspace(0, len(pairs_all) - 1, n_sample).astype(np.int64)
    sample = pairs_all[sel]
    desc = scene.desc.astype(np.float32)
    t0 = time.perf_counter()
    res = oracle.match_pairs(desc, scene.pts, scene.offsets, sample)
    dt = time.perf_counter() - t0
    ok = None
    if graph is not None:
        counts, matches = graph
        off = np.concatenate([[0], np.cumsum(counts)])
        ok = all(np.array_equal(matches[off[p]: off[p + 1]], r) for p, r in zip(sel, res))
    return {
        "value": round(n_sample / dt, 3),
        "unit": "pairs/s",
        "cores": oracle.num_threads(),
        "kind": "port",
        "sample": f"{n_sample} pairs strided over the same pair list, {dt:.1f} s, OpenMP over pairs",
        "parity_on_sample": ok,
    }


if __name__ == "__main__":
    main()
