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

# the CPU legs' OpenMP threads sleep between parallel regions instead of spinning: a GPU leg timed right after an oracle leg otherwise shares
# the host cores with a few hundred busy-waiting threads (measured: 8 ms of solver run became 95 ms).  Read when the OpenMP runtime starts.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_I8_TOPS = 5000.0  # dense int8 MFMA peak of MI355X (2x the 2.5 PFLOP/s bf16 dense peak); ubench ceiling 4404
FLOP_PER_PAIR = 2.0 * 2000 * 2000 * 128  # one distance matrix serves both directions
PEAK_F64_VALU_TFLOPS = 78.6  # fp64 vector peak (half the 157.3 TFLOP/s fp32 rate of MI355X_MICROARCH.md)
RANSAC_FLOP_PER_MODEL_POINT = 40.0  # symmetric epipolar error of one correspondence under one F: two 3x3 products, two norms, compare
RELPOSE_FLOP_PER_MODEL_POINT = 150.0  # RelativePose::Evaluate: rotate, midpoint triangulation, two reprojection cosines
PMC_FILE = os.path.join(ROOT, "profiles", "r06_match_pmc.json")  # HBM bytes per launch from the rocprofv3 --pmc passes of this command (tools/pmc_passes.sh)
PMC_KERNEL = "match_fused_kernel"  # the kernel the roofline block is about: a counter file taken on another kernel is refused, not quoted


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
    ap.add_argument("--ba-iters", type=int, default=20)
    ap.add_argument("--no-robust", action="store_true", help="descriptor stage only (debug)")
    ap.add_argument("--strong", action="store_true", help="fixed total work for every N (configs[3]: --images 10000 --strong)")
    ap.add_argument("--no-overlap", action="store_true", help="skip the neighbour-preselected workload")
    ap.add_argument("--overlap-neighbors", type=int, default=16)
    ap.add_argument("--no-calibrated", action="store_true", help="skip the calibrated (essential-matrix) branch legs")
    ap.add_argument("--no-float", action="store_true", help="skip the float-descriptor (root-SIFT) workload")
    ap.add_argument("--no-guided", action="store_true", help="skip the guided-matching workload")
    ap.add_argument("--no-hahog", action="store_true", help="skip the HAHOG extraction workload")
    ap.add_argument("--full-parity", action="store_true", help="check EVERY pair with matches + 5000 empties against the oracle (~1.5 min)")
    ap.add_argument("--headline-only", action="store_true", help="only the headline workload + its cpu_baseline (configs[3] runs)")
    ap.add_argument("--all-sections", action="store_true", help="N > 1: also run the one-GPU secondary workloads and CPU baselines on rank 0")
    ap.add_argument("--emulate-world", type=int, default=8, help="N = 1: size of the emulated exchange step (exchange_emulation section; 0 = skip)")
    a = ap.parse_args()
    if a.headline_only:
        a.no_ba = a.no_tracks = a.no_overlap = a.no_calibrated = a.no_float = a.no_guided = a.no_hahog = True
    return a


def headline_only_for_ranks(args, world: int) -> bool:
    """N > 1: the line is the sharded headline workload only.  The secondary workloads and the CPU baselines are one-GPU / host
    measurements (reported at N = 1, as the bench contract asks for cpu_baseline); repeating them on rank 0 would keep the other
    ranks in the closing barrier for minutes.  Returns True when it switched them off."""
    if world <= 1 or getattr(args, "all_sections", False):
        return False
    for flag in ("no_overlap", "no_calibrated", "no_float", "no_guided", "no_cpu_baseline", "no_tracks", "no_hahog", "no_ba"):
        setattr(args, flag, True)
    return True


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, RCCL rendezvous on 127.0.0.1),
    exactly as the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` does; rank 0 of
    the children prints the JSON line."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # dmabuf IPC only on this driver: RCCL needs it in EVERY rank, also when the driver's torch.distributed.run started them
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # a run can never be labelled with a GPU count it did not use
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a launcher)"
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
    n_images = args.images if (world == 1 or args.strong) else int(math.ceil((1 + math.sqrt(1 + 8.0 * world * p1)) / 2))
    t0 = time.time()
    scene = synthetic.make_matching_scene(n_images, args.features, seed=args.seed)
    pairs_all = synthetic.all_pairs(n_images)
    pairs_all = pairs_all[: (p1 if args.strong else world * p1)]
    my_pairs = odist.shard_pairs(pairs_all, rank, world)
    pairs_gathered = pairs_all[odist.gathered_pair_order(len(pairs_all), world)]  # pair list of the gathered graph
    store = matching.DescriptorStore.from_packed(scene.desc, scene.pts, scene.offsets, ctx)
    t_setup = time.time() - t0
    robust = not args.no_robust

    class Done:  # (N = 1: the result is there when the call returns)
        def __init__(self, res):
            self.res = res

        def wait(self):
            return self.res

    def step(tm=None):
        if world == 1:
            return Done(matching.match_pairs(store, my_pairs, robust=robust, timings=tm))
        # N ranks: the shard's match rows stay in HBM and the exchange step all-gathers from there (RCCL over xGMI); rank-major
        # gathered order: no host-side scatter of the match rows inside the timed region.  The copy of the gathered graph to the host
        # (most of the exchange step's time) is queued on a side stream and lands under the NEXT step's matching (round 6): the handle
        # is waited for one step later; the collectives stay in program order on this thread
        g = matching.match_pairs(store, my_pairs, robust=robust, timings=tm, keep_device=True)
        try:
            xt = {}
            res = odist.all_gather_match_graph_device(g, len(pairs_all), rank, world, local_rank, reorder=False, timings=xt, defer_host_copy=True)
            if tm is not None:
                exchange_tms.append(xt)
            return res
        finally:
            g.close()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    exchange_tms = []
    for _ in range(args.warmup):
        step().wait()
    tms = []
    barrier()
    t0 = time.perf_counter()
    graph, pending = None, None
    for _ in range(args.steps):
        tm = MatchTimings()
        h = step(tm)
        if pending is not None:
            graph = pending.wait()  # the previous step's graph: its host copy ran under this step's matching
        pending = h
        tms.append(tm)
    graph = pending.wait() if pending is not None else None  # every step's result is on the host inside the timed region
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
        "kernel": "match_fused_kernel",
        "achieved": round(achieved, 2),
        "peak": PEAK_I8_TOPS,
        "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_I8_TOPS, 4),
        "traffic": None,
        "avg_launch_ms": round(avg_ms, 3),
        "launches": launches,
        "algorithmic_flop_per_pair": flop_pair,
    }

    # HBM traffic per launch: PMC passes cannot run inside this process, so the committed counters of the same command
    # (tools/r03_final_profiles.sh -> profiles/r03_match_pmc.json: FETCH_SIZE doubled as the gfx950 guide prescribes, + WRITE_SIZE) are quoted,
    # scaled to this run's pairs per launch
    if os.path.exists(PMC_FILE):
        pmc = json.load(open(PMC_FILE))
        assert pmc.get("kernel") == PMC_KERNEL, f"{PMC_FILE} holds counters of {pmc.get('kernel')!r}, the roofline block is about {PMC_KERNEL!r}: re-run tools/pmc_passes.sh"
        try:
            k = pairs_per_launch / float(pmc["pairs_per_launch"])
            roofline["traffic"] = {"hbm_read_bytes": pmc["fetch_bytes_corrected"] * k, "hbm_write_bytes": pmc["write_bytes"] * k,
                                   "algorithmic_bytes": 2 * n_avg * 128 * pairs_per_launch, "source": pmc["source"]}
        except (KeyError, ValueError):
            pass
    rs_ms = sum(float(t.ms_ransac_kernel) for t in tms)
    rs_work = sum(int(t.ransac_model_points) for t in tms)
    ransac_line = {
        "bound": "valu-f64", "kernel": "fransac_draw / solve / decide / rest kernels", "unit": "TFLOP/s", "peak": PEAK_F64_VALU_TFLOPS,
        "model_points_per_s": round(rs_work / (rs_ms * 1e-3), 1) if rs_ms > 0 else 0.0,
        "achieved": round(rs_work * RANSAC_FLOP_PER_MODEL_POINT / (rs_ms * 1e-3) / 1e12, 4) if rs_ms > 0 else 0.0,
        "flop_per_model_point": RANSAC_FLOP_PER_MODEL_POINT,
        "note": "hypotheses x correspondences actually scored (lazy: only the models the sequential loop reaches) per second of the four "
                "kernels' own time on the matcher's stream, HIP events around them",
    }
    ransac_line["frac"] = round(ransac_line["achieved"] / PEAK_F64_VALU_TFLOPS, 5)

    out = {
        "metric": "image-pairs matched/sec (+ BA LM-iters/sec, 5k cams / 500k pts)",
        "value": round(value, 1),
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
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
        "roofline_ransac": ransac_line,
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
    if exchange_tms:  # N > 1: this rank's exchange step, inside the timed region
        ex = {k: round(float(np.mean([x[k] for x in exchange_tms])), 3) for k in ("collective_ms", "layout_ms", "d2h_ms")}
        ex["ms_per_step"] = round(sum(ex.values()), 3)
        ex["share_of_step"] = round(ex["ms_per_step"] / (1e3 * elapsed / args.steps), 4)
        ex["bytes_to_host"] = int(exchange_tms[-1]["bytes_to_host"])
        out["exchange"] = ex
    if dist is not None:  # N > 1: every rank's own clocks, so that a poor scaling curve can be read from one run
        mine = {"rank": rank, "pairs": int(len(my_pairs)),
                "match_call_ms": round(sum(float(t.ms_total) for t in tms) / args.steps, 3),
                "match_kernel_ms": round(ms_kernel / args.steps, 3),
                "ransac_kernel_ms": round(sum(float(t.ms_ransac_kernel) for t in tms) / args.steps, 3),
                "exchange_ms": round(float(np.mean([x["collective_ms"] + x["layout_ms"] + x["d2h_ms"] for x in exchange_tms])), 3) if exchange_tms else None,
                "exchange_collective_ms": round(float(np.mean([x["collective_ms"] for x in exchange_tms])), 3) if exchange_tms else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        out["per_rank"] = per_rank

    if rank == 0:
        def section(name, fn, *a):
            """the secondary workloads must not take the headline line down with them: a failure is recorded in place of the numbers"""
            try:
                out[name] = fn(*a)
            except Exception as exc:  # noqa: BLE001
                import traceback

                out[name] = {"error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-1500:]}

        if headline_only_for_ranks(args, world):
            out["note"] = "secondary workloads and cpu_baseline are measured at N = 1 (python bench.py); --all-sections repeats them on rank 0"
        if world == 1 and args.emulate_world > 1 and not args.headline_only:
            section("exchange_emulation", exchange_emulation, args, store, my_pairs, local_rank, robust, out["ms_per_step"])
        if not args.no_overlap:
            section("overlap_workload", overlap_bench, args, ctx, store, scene, n_images, not args.no_cpu_baseline)
        if not args.no_calibrated:
            section("calibrated", calibrated_bench, args, ctx, store, scene, n_images, not args.no_cpu_baseline)
        if not args.no_float:
            section("float_descriptors", float_bench, args, ctx, scene, pairs_all if world == 1 else pairs_all[:p1], n_images, not args.no_cpu_baseline)
        if not args.no_guided:
            section("guided", guided_bench, args, ctx, store, scene, n_images, not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            section("cpu_baseline", cpu_baseline, scene, pairs_gathered, args.cpu_sample_pairs, graph, args.full_parity)
        if not args.no_tracks and graph is not None:
            section("tracks", tracks_bench, ctx, scene, pairs_gathered, graph, not args.no_cpu_baseline)
        if not args.no_hahog:
            section("hahog", hahog_bench, ctx, not args.no_cpu_baseline)
        if not args.no_ba:
            def ba_section():
                import bench_ba as ba_bench

                return ba_bench.run(ctx, args.ba_shots, args.ba_points, args.ba_track, args.ba_iters, cpu_baseline=not args.no_cpu_baseline)

            section("ba", ba_section)
        # the JSON line is the LAST line of stdout: RCCL's version banner (written through C stdio when the exchange emulation makes its
        # one-rank communicator) would otherwise be flushed behind it at exit
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def exchange_emulation(args, store, my_pairs, local_rank, robust, ms_per_step_n1, reps: int = 5):
    """What the exchange step of an E-rank job costs each rank beyond the collective's wire time, measured on ONE GPU: a one-rank RCCL
    group runs the very function the N-rank step calls (dist.all_gather_match_graph_device), with the receive buffers filled with E
    copies of this rank's payload (emulate_world), so the device-side layout and the D2H of the gathered graph run at the E-rank size.
    The xGMI transfer itself is replaced by D2D copies; its size is reported next to the link rate."""
    import socket

    import torch
    import torch.distributed as dist

    from opensfm_amd import dist as odist
    from opensfm_amd import matching

    E = int(args.emulate_world)
    own_group = not dist.is_initialized()
    if own_group:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    try:
        n = len(my_pairs)
        t0 = time.perf_counter()
        g = matching.match_pairs(store, my_pairs, robust=robust, keep_device=True)
        match_ms = 1e3 * (time.perf_counter() - t0)
        res = {}
        try:
            for reorder in (False, True):
                odist.all_gather_match_graph_device(g, E * n, 0, 1, local_rank, block=n, reorder=reorder, emulate_world=E)  # warm-up: allocator caches
                acc = []
                for _ in range(reps):
                    xt = {}
                    t0 = time.perf_counter()
                    c, m = odist.all_gather_match_graph_device(g, E * n, 0, 1, local_rank, block=n, reorder=reorder, emulate_world=E, timings=xt)
                    xt["call_ms"] = 1e3 * (time.perf_counter() - t0)
                    acc.append(xt)
                    ok = len(c) == E * n and len(m) == E * g.total and np.array_equal(c[:n], g.counts) and np.array_equal(c[-n:], g.counts)
                    del c, m
                r = {k: round(float(np.mean([x[k] for x in acc])), 3) for k in ("call_ms", "collective_ms", "layout_ms", "d2h_ms")}
                r["share_of_step_serial"] = round(r["call_ms"] / (match_ms + r["call_ms"]), 4)
                # what bench.py's N-rank step does since round 6: the host copy deferred to a side stream, waited for after the NEXT matching
                exposed, hidden_ok = [], True
                for _ in range(reps):
                    t0 = time.perf_counter()
                    h = odist.all_gather_match_graph_device(g, E * n, 0, 1, local_rank, block=n, reorder=reorder, emulate_world=E, defer_host_copy=True)
                    exposed.append(1e3 * (time.perf_counter() - t0))
                    g2 = matching.match_pairs(store, my_pairs, robust=robust, keep_device=True)  # the next step's matching
                    t1 = time.perf_counter()
                    c, m = h.wait()
                    waited = 1e3 * (time.perf_counter() - t1)
                    g2.close()
                    hidden_ok = hidden_ok and waited < 0.5 and len(c) == E * n and np.array_equal(c[:n], g.counts) and np.array_equal(c[-n:], g.counts)
                    del c, m, h
                r["exposed_ms_deferred_copy"] = round(float(np.mean(exposed)), 3)
                r["share_of_step"] = round(r["exposed_ms_deferred_copy"] / (match_ms + r["exposed_ms_deferred_copy"]), 4)
                r["host_copy_hidden_under_next_matching"] = bool(hidden_ok)
                r["layout_checked"] = bool(ok)
                res["rank_major" if not reorder else "original_pair_order"] = r
        finally:
            g.close()
        to_host = 4 * E * n + 8 * E * g.total
        return {"emulated_ranks": E, "pairs_per_rank": int(n), "match_rows_per_rank": int(g.total), "bytes_to_host_per_rank": int(to_host),
                "bytes_over_xgmi_per_rank": int(to_host * (E - 1) // E), "xgmi_link_GBps": 153.0,
                "est_wire_ms_one_link": round(to_host * (E - 1) / E / 153e9 * 1e3, 3),
                "match_ms_keep_device": round(match_ms, 3), "ms_per_step_n1": ms_per_step_n1, **res,
                "note": "bench.py's N-rank step uses the rank-major layout with the host copy of the gathered graph deferred under the next step's matching "
                        "(share_of_step = exposed exchange / (shard matching + exposed exchange); share_of_step_serial = round 5's serial step); the "
                        "collective's wire time (est_wire_ms_one_link: ring all-gather bound by one 153 GB/s link) is NOT in the measured figure"}
    finally:
        if own_group:
            dist.destroy_process_group()


def hahog_bench(ctx, with_cpu, rows: int = 1536, cols: int = 2048, target: int = 10000, reps: int = 5):
    """HAHOG extraction at OpenSfM's processing size (feature_process_size 2048, config.py:33) with its default
    thresholds and feature_min_frames-like count: images per second of osfm_hahog_extract (upload of the grey image, every kernel,
    download of keypoints and descriptors), next to the REFERENCE's features::hahog compiled from /root/reference (one thread: vlfeat
    is built without OpenMP, as the reference's own CMake builds it) on the same image."""
    from opensfm_amd import features

    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    im = np.zeros((rows, cols), np.float32)
    for _ in range(1500):
        cx, cy, sg = rng.uniform(0, cols), rng.uniform(0, rows), rng.uniform(1.5, 24)
        x0, x1, y0, y1 = int(max(0, cx - 4 * sg)), int(min(cols, cx + 4 * sg)), int(max(0, cy - 4 * sg)), int(min(rows, cy + 4 * sg))
        im[y0:y1, x0:x1] += rng.uniform(-1, 1) * np.exp(-((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2) / (2 * sg * sg))
    im += 0.05 * rng.standard_normal((rows, cols)).astype(np.float32)
    im = np.ascontiguousarray((im - im.min()) / (im.max() - im.min()), np.float32)
    features.hahog(im, 1e-5, 10.0, target, ctx=ctx)  # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        pts, desc = features.hahog(im, 1e-5, 10.0, target, ctx=ctx)
    ms = (time.perf_counter() - t0) / reps * 1e3
    octaves = int(np.floor(np.log2((min(rows, cols) - 1) / 15.0))) + 1
    px = sum((rows >> o) * (cols >> o) for o in range(octaves))
    # algorithmic bytes: per level two separable passes (read + write each), the response (read + write), the extremum search reads the
    # response once: 5 levels x (16 + 8 + 4) bytes per pixel of the pyramid; the per-feature patches are negligible beside it
    alg_bytes = 5 * 28 * px
    out = {"workload": f"{rows} x {cols} grey image, peak 1e-5, edge 10, {target} features", "value": round(1e3 / ms, 2), "unit": "images/s",
           "ms_per_image": round(ms, 3), "features": int(len(pts)), "octaves": octaves,
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "achieved": round(alg_bytes / (ms * 1e-3) / 1e9, 1),
                        "frac": round(alg_bytes / (ms * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_bytes": alg_bytes,
                        "note": "whole call (H2D of the image, ~50 launches, two host round trips for the feature counts, D2H of the results) "
                                "against the streaming bytes of the pyramid"}}
    # what the pipeline hands over: a decoded grey image, uint8 (features.extract_features_hahog, opensfm/features.py:516-534) -- the bytes go to the
    # device as they are and level / 255 is formed there (OSFM_HAHOG_IMAGE_U8): no host-side float pass, a quarter of the PCIe bytes
    try:
        im8 = np.ascontiguousarray(np.round(255 * im), np.uint8)
        cfg = {"feature_root": True, "hahog_normalize_to_uchar": True, "hahog_peak_threshold": 1e-5, "hahog_edge_threshold": 10.0}
        features.extract_features_hahog(im8, cfg, target, ctx=ctx)
        t0 = time.perf_counter()
        for _ in range(reps):
            p8, d8 = features.extract_features_hahog(im8, cfg, target, ctx=ctx)
        ms8 = (time.perf_counter() - t0) / reps * 1e3
        t0 = time.perf_counter()
        for _ in range(reps):
            pf, df = features.extract_features_hahog(im8.astype(np.float32), cfg, target, ctx=ctx)  # (a float image takes the host-side division)
        msf = (time.perf_counter() - t0) / reps * 1e3
        out["uint8_image"] = {"value": round(1e3 / ms8, 2), "unit": "images/s", "ms_per_image": round(ms8, 3), "features": int(len(p8)),
                              "float_path_images_per_s": round(1e3 / msf, 2), "identical_to_float_path": bool(np.array_equal(p8, pf) and np.array_equal(d8, df)),
                              "note": "features.extract_features_hahog on the decoded uint8 image (root + uchar descriptors), image by image"}
        nb = 32
        for conc in (8,):
            # (warm-up = the timed call itself: the streams, the cached device blocks AND the host pages of a 32-image call; a warm-up of 8
            #  images left the first 32-image call at about half its steady rate -- tools/r06_hahog_batch_matrix.py)
            for _ in range(2):  # (twice: this is the first batch call of the process -- eight streams, sixteen cached device blocks)
                features.hahog_batch([im8] * nb, 1e-5, 10.0, target, flags=features.HAHOG_ROOT | features.HAHOG_UCHAR, concurrency=conc, ctx=ctx)
            t0 = time.perf_counter()
            for _ in range(2):
                res8 = features.hahog_batch([im8] * nb, 1e-5, 10.0, target, flags=features.HAHOG_ROOT | features.HAHOG_UCHAR, concurrency=conc, ctx=ctx)
            dt = (time.perf_counter() - t0) / 2
            out["uint8_image"][f"batch_host_images_x{conc}"] = {"value": round(nb / dt, 1), "unit": "images/s", "images": nb, "concurrency": conc,
                                                               "identical_to_single": bool(all(np.array_equal(p, p8) and np.array_equal(dd, d8) for p, dd in res8))}
    except Exception as e:  # noqa: BLE001
        out["uint8_image"] = {"error": f"{type(e).__name__}: {e}"}
    # a data set's worth of images in one call (osfm_hahog_extract_batch): several in flight on separate streams / host threads
    try:
        nb = 32
        for conc in (4, 8):
            features.hahog_batch([im] * nb, 1e-5, 10.0, target, concurrency=conc, ctx=ctx)  # warm-up: streams, block cache, host pages
            t0 = time.perf_counter()
            for _ in range(2):
                res = features.hahog_batch([im] * nb, 1e-5, 10.0, target, concurrency=conc, ctx=ctx)
            dt = (time.perf_counter() - t0) / 2
            out[f"batch_host_images_x{conc}"] = {"value": round(nb / dt, 1), "unit": "images/s", "images": nb, "concurrency": conc,
                                                 "identical_to_single": bool(all(np.array_equal(p, pts) and np.array_equal(dd, desc) for p, dd in res))}
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")  # a resident copy of the image without going through torch
        dptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(im.nbytes)) == 0
        try:
            assert hip.hipMemcpy(dptr, ctypes.c_void_p(im.ctypes.data), ctypes.c_size_t(im.nbytes), 1) == 0
            for conc in (4, 8, 12):
                features.hahog_batch([dptr.value] * nb, 1e-5, 10.0, target, concurrency=conc, shapes=[im.shape] * nb, ctx=ctx)
                t0 = time.perf_counter()
                for _ in range(2):
                    res = features.hahog_batch([dptr.value] * nb, 1e-5, 10.0, target, concurrency=conc, shapes=[im.shape] * nb, ctx=ctx)
                dt = (time.perf_counter() - t0) / 2
                out[f"batch_resident_images_x{conc}"] = {
                    "value": round(nb / dt, 1), "unit": "images/s", "images": nb, "concurrency": conc,
                    "hbm_frac": round(alg_bytes * nb / dt / 1e9 / 8000.0, 4),
                    "identical_to_single": bool(all(np.array_equal(p, pts) and np.array_equal(dd, desc) for p, dd in res)),
                    "note": "images already in HBM (OSFM_HAHOG_IMAGE_ON_DEVICE); keypoints and descriptors still come back to the host"}
        finally:
            hip.hipFree(dptr)
    except Exception as exc:  # noqa: BLE001
        out["batch_error"] = f"{type(exc).__name__}: {exc}"
    if with_cpu:
        import oracle

        t0 = time.perf_counter()
        ref = oracle.hahog_ref(im, 1e-5, 10.0, target)
        dt = time.perf_counter() - t0
        if ref is not None:
            same = ref[0].shape == pts.shape and np.array_equal(ref[0][:, :3], pts[:, :3]) and np.array_equal(ref[1], desc)
            out["cpu_baseline"] = {"value": round(1.0 / dt, 3), "unit": "images/s", "cores": 1, "kind": "reference",
                                   "sample": f"the same image once ({dt:.2f} s; oracle/_ref/libhahog_ref.so = hahog.cc + vlfeat compiled from /root/reference)",
                                   "identical_keypoints_and_descriptors": bool(same)}
    return out


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


def neighbour_pairs(n_images: int, k: int) -> np.ndarray:
    """(i, j), i < j <= i + k: what pairs_selection emits with matching_order_neighbors = k (opensfm/pairs_selection.py:347-368) on a
    sequence -- every pair sees common scene points, unlike the exhaustive list where 98 % of the pairs are empty."""
    i = np.repeat(np.arange(n_images), k)
    j = i + np.tile(np.arange(1, k + 1), n_images)
    keep = j < n_images
    return np.stack([i[keep], j[keep]], 1).astype(np.int32)


def overlap_bench(args, ctx, store, scene, n_images, with_cpu):
    """Second timed workload on the same store: the neighbour-preselected pair list (the normal OpenSfM use).  Every pair has work
    for the ratio / mutual re-examination and for the RANSAC stage."""
    from opensfm_amd import matching
    from opensfm_amd._lib import MatchTimings

    pairs = neighbour_pairs(n_images, args.overlap_neighbors)
    matching.match_pairs(store, pairs)  # untimed: the context's chunk buffers grow to this list's size here, not inside the timed calls
    tms = []
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        tm = MatchTimings()
        counts, m = matching.match_pairs(store, pairs, timings=tm)
        tms.append(tm)
    dt = (time.perf_counter() - t0) / max(1, args.steps)
    c0, _ = matching.match_pairs(store, pairs, robust=False)
    ms_k = float(np.mean([t.ms_match_kernel for t in tms]))
    ms_r = float(np.mean([t.ms_ransac_kernel for t in tms]))
    n_avg = float(np.mean(np.diff(scene.offsets)))
    flop = 2.0 * n_avg * n_avg * 128 * len(pairs)
    out = {
        "workload": f"{len(pairs)} pairs (i, j), j - i <= {args.overlap_neighbors}, of the same {n_images} x {args.features} store",
        "value": round(len(pairs) / dt, 1), "unit": "pairs/s",
        "descriptor_stage_pairs_per_s": round(len(pairs) / (ms_k * 1e-3), 1),
        "match_kernel_ms": round(ms_k, 3), "ransac_kernel_ms": round(ms_r, 3), "call_ms": round(1e3 * dt, 3),
        "ransac_share_of_stream_time": round(ms_r / max(ms_k + ms_r, 1e-9), 3),
        "pairs_reaching_ransac": int((c0 >= 20).sum()), "pairs_with_matches": int((counts > 0).sum()),
        "descriptor_matches_per_pair": round(float(c0.mean()), 1), "inlier_matches_per_pair": round(float(counts.mean()), 1),
        "roofline": {"bound": "mfma", "kernel": "match_fused_kernel", "unit": "TFLOP/s", "peak": PEAK_I8_TOPS,
                     "achieved": round(flop / (ms_k * 1e-3) / 1e12, 2), "frac": round(flop / (ms_k * 1e-3) / 1e12 / PEAK_I8_TOPS, 4)},
        "roofline_ransac": {"bound": "valu-f64", "kernel": "fransac_draw / solve / decide / rest kernels", "unit": "TFLOP/s", "peak": PEAK_F64_VALU_TFLOPS,
                            "model_points_per_s": round(float(np.mean([t.ransac_model_points for t in tms])) / (ms_r * 1e-3), 1),
                            "achieved": round(float(np.mean([t.ransac_model_points for t in tms])) * RANSAC_FLOP_PER_MODEL_POINT / (ms_r * 1e-3) / 1e12, 4)},
    }
    out["roofline_ransac"]["frac"] = round(out["roofline_ransac"]["achieved"] / PEAK_F64_VALU_TFLOPS, 5)
    if with_cpu:
        import oracle

        sel = np.linspace(0, len(pairs) - 1, 768).astype(np.int64)
        t0 = time.perf_counter()
        res = oracle.match_pairs(scene.desc.astype(np.float32), scene.pts, scene.offsets, pairs[sel])
        dtc = time.perf_counter() - t0
        off = np.concatenate([[0], np.cumsum(counts)])
        out["cpu_baseline"] = {"value": round(len(sel) / dtc, 3), "unit": "pairs/s", "cores": oracle.num_threads(), "kind": "port",
                               "sample": f"{len(sel)} pairs strided over the list, {dtc:.1f} s, OpenMP over pairs",
                               "parity_on_sample": bool(all(np.array_equal(m[off[p]: off[p + 1]], r) for p, r in zip(sel, res)))}
    return out


def root_features(desc_u8: np.ndarray) -> np.ndarray:
    """features.root_feature (opensfm/features.py:292-298): L1-normalise, square root -- what feature_root = True (the default) makes
    of SIFT descriptors: float32 values that are not integers"""
    d = desc_u8.astype(np.float32)
    d /= np.maximum(d.sum(1, keepdims=True), 1e-7)
    return np.sqrt(d).astype(np.float32)


def float_bench(args, ctx, scene, pairs_all, n_images, with_cpu):
    """configs[1] again on root (non-integer float32) descriptors: the fused kernel in FQ mode -- int8 quantisation with rigorous bounds on
    the matrix pipe, float32 evaluation of the undecided queries -- same pair list, same semantics, results identical to cv2's float
    arithmetic as the oracle restates it."""
    from opensfm_amd import matching
    from opensfm_amd._lib import MatchTimings

    desc = root_features(scene.desc)
    store = matching.DescriptorStore.from_packed(desc, scene.pts, scene.offsets, ctx)
    try:
        near = neighbour_pairs(n_images, args.overlap_neighbors)
        matching.match_pairs(store, near[:512])
        res = {}
        for name, pl in (("exhaustive", pairs_all), ("neighbour", near)):
            tm = MatchTimings()
            t0 = time.perf_counter()
            counts, m = matching.match_pairs(store, pl, timings=tm)
            dt = time.perf_counter() - t0
            n_avg = float(np.mean(np.diff(scene.offsets)))
            flop = 2.0 * n_avg * n_avg * 128 * len(pl)
            res[name] = {"pairs": int(len(pl)), "value": round(len(pl) / dt, 1), "unit": "pairs/s",
                         "descriptor_stage_pairs_per_s": round(len(pl) / (tm.ms_match_kernel * 1e-3), 1),
                         "match_kernel_ms": round(float(tm.ms_match_kernel), 3), "call_ms": round(1e3 * dt, 3),
                         "pairs_with_float_evaluation": int(tm.pairs_exact_path), "pairs_with_matches": int((counts > 0).sum()),
                         "total_inlier_matches": int(counts.sum()),
                         "roofline": {"bound": "mfma", "kernel": "match_fused_kernel<FQ>", "unit": "TFLOP/s", "peak": PEAK_I8_TOPS,
                                      "achieved": round(flop / (tm.ms_match_kernel * 1e-3) / 1e12, 2),
                                      "frac": round(flop / (tm.ms_match_kernel * 1e-3) / 1e12 / PEAK_I8_TOPS, 4)}}
        out = {"workload": f"{n_images} images x {args.features} x 128-D root descriptors (float32, L2-normalised; features.py:292-298), the same pair lists",
               "dtype": "i8 quantised candidates (exact int32) + f32 evaluation in cv2's accumulation order + f64 RANSAC", **res}
        if with_cpu:
            import oracle

            sel = np.linspace(0, len(near) - 1, 96).astype(np.int64)
            t0 = time.perf_counter()
            want = oracle.match_pairs(desc, scene.pts, scene.offsets, near[sel])
            dtc = time.perf_counter() - t0
            c, m = matching.match_pairs(store, near[sel])
            got = matching.split_matches(c, m)
            out["cpu_baseline"] = {"value": round(len(sel) / dtc, 3), "unit": "pairs/s", "cores": oracle.num_threads(), "kind": "port",
                                   "sample": f"{len(sel)} neighbour pairs, {dtc:.1f} s, OpenMP over pairs (float32 brute force)",
                                   "parity_on_sample": bool(all(np.array_equal(g, w) for g, w in zip(got, want)))}
        return out
    finally:
        store.close()


def guided_bench(args, ctx, store, scene, n_images, with_cpu):
    """Guided matching (match_images_with_pairs with poses, matching.py:204-207,260-337) on the neighbour list of the first images of the
    same store, with the views' true relative poses: epipolar mask (guided_matching_threshold 0.006 rad) + masked symmetric matcher +
    gates + fundamental-matrix RANSAC, one osfm_match_pairs_guided call."""
    from opensfm_amd import matching
    from opensfm_amd._lib import MatchTimings

    n_g = min(n_images, 300)
    pairs = neighbour_pairs(n_g, args.overlap_neighbors)
    focal = 0.85
    b = np.c_[scene.pts / focal, np.ones(len(scene.pts))]
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    bears = [b[scene.offsets[i]: scene.offsets[i + 1]].astype(np.float32) for i in range(n_images)]
    rels = []
    for a, c in pairs:
        Ra, Rc, oa, oc = scene.cam_R[a], scene.cam_R[c], scene.cam_o[a], scene.cam_o[c]
        rels.append(np.concatenate([(Rc @ Ra.T).T.reshape(9), Ra @ (oc - oa)]))  # pose_c.relative_to(pose_a): R cam-to-world, origin
    cfg = {"guided_matching_threshold": 0.006}
    matching.match_pairs_guided(store, pairs[:256], bears, rels[:256], cfg)
    tm = MatchTimings()
    t0 = time.perf_counter()
    counts, m = matching.match_pairs_guided(store, pairs, bears, rels, cfg, robust=True, timings=tm)
    dt = time.perf_counter() - t0
    out = {"workload": f"{len(pairs)} neighbour pairs of the first {n_g} images (2000 x 2000 features), true relative poses, threshold 0.006 rad",
           "value": round(len(pairs) / dt, 1), "unit": "pairs/s",
           "descriptor_stage_pairs_per_s": round(len(pairs) / (tm.ms_match_kernel * 1e-3), 1),
           "descriptor_stage_ms": round(float(tm.ms_match_kernel), 3), "call_ms": round(1e3 * dt, 3),
           "pairs_with_matches": int((counts > 0).sum()), "inlier_matches_per_pair": round(float(counts.mean()), 1),
           "roofline": {"bound": "valu-f64", "kernel": "guided_pairs_match_kernel", "unit": "TFLOP/s", "peak": PEAK_F64_VALU_TFLOPS,
                        "algorithmic_flop": "13 per (query, target) and direction: the epipolar predicate on all n1 x n2 combinations",
                        "achieved": round(13.0 * 2 * sum(float(len(bears[a])) * len(bears[c]) for a, c in pairs) / (tm.ms_match_kernel * 1e-3) / 1e12, 3)}}
    out["roofline"]["frac"] = round(out["roofline"]["achieved"] / PEAK_F64_VALU_TFLOPS, 4)
    if with_cpu:
        import oracle

        sel = np.linspace(0, len(pairs) - 1, 6).astype(np.int64)
        t0 = time.perf_counter()
        ok = True
        off = np.concatenate([[0], np.cumsum(counts)])
        d32 = scene.desc.astype(np.float32)
        for p in sel:
            a, c = pairs[p]
            da, dc = d32[scene.offsets[a]: scene.offsets[a + 1]], d32[scene.offsets[c]: scene.offsets[c + 1]]
            pa, pc = scene.pts[scene.offsets[a]: scene.offsets[a + 1]], scene.pts[scene.offsets[c]: scene.offsets[c + 1]]
            emask, _ = oracle.epipolar_mask(bears[a], bears[c], rels[p][:9].reshape(3, 3), rels[p][9:], 0.006)
            mm = oracle.match_brute_force_masked(da, dc, emask, 0.8, symmetric=True)
            want = np.zeros((0, 2), np.int32)
            if len(mm) >= 20:
                F, mask, _ = oracle.find_fundamental_ransac(pa[mm[:, 0]], pc[mm[:, 1]], 0.004, 0.9999)
                if F is not None and F[2, 2] != 0.0 and mask.sum() >= 20:
                    want = mm[mask]
            ok = ok and np.array_equal(m[off[p]: off[p + 1]], want)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(len(sel) / dtc, 3), "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"{len(sel)} pairs strided over the list, {dtc:.1f} s, one thread (mask + masked matcher + RANSAC)",
                               "parity_on_sample": bool(ok)}
    return out


def calibrated_bench(args, ctx, store, scene, n_images, with_cpu):
    """The essential-matrix branch of robust_match (every camera that is not an undistorted perspective one, matching.py:906-929):
    (a) the geometric stage alone on synthetic bearing sets (osfm_relpose_pairs, 300 correspondences per pair, 40 % outliers), with
    its fp64-VALU roofline line; (b) matching.match end to end (osfm_match_pairs_calibrated, device-resident) on the neighbour list
    of the same store under a slightly distorted camera."""
    from types import SimpleNamespace

    from opensfm_amd import matching
    from opensfm_amd._lib import MatchTimings

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from relpose_bench import two_view_bearings

    rng = np.random.default_rng(1)
    n_pairs, n = 16384, 300
    base = [two_view_bearings(rng, n, 0.4) for _ in range(256)]  # 256 distinct problems, tiled: the rounds do not care
    b1 = np.concatenate([base[k % 256][0] for k in range(n_pairs)])
    b2 = np.concatenate([base[k % 256][1] for k in range(n_pairs)])
    off = np.arange(n_pairs + 1, dtype=np.int64) * n
    matching.relpose_pairs(b1[: 64 * n], b2[: 64 * n], off[:65], 0.004, "match", ctx=ctx)
    res, mask, ms = matching.relpose_pairs(b1, b2, off, 0.004, "match", ctx=ctx)
    iters = np.array([r["iterations"] for r in res])
    flops = RELPOSE_FLOP_PER_MODEL_POINT * 6.0 * float(iters.sum()) * n
    out = {"geometric_stage": {
        "workload": f"{n_pairs} pairs x {n} correspondences, 40 % outliers, threshold 0.004 rad, LO-RANSAC + 3 refinement rounds",
        "value": round(n_pairs / (ms * 1e-3), 1), "unit": "pairs/s", "device_ms": round(ms, 2),
        "mean_ransac_iterations": round(float(iters.mean()), 1),
        "roofline": {"bound": "valu-f64", "kernel": "rp_walk_kernel (+ rp_solve5a/b, rp_solveN, rp_pose, rp_finish)", "unit": "TFLOP/s",
                     "peak": PEAK_F64_VALU_TFLOPS, "achieved": round(flops / (ms * 1e-3) / 1e12, 3),
                     "frac": round(flops / (ms * 1e-3) / 1e12 / PEAK_F64_VALU_TFLOPS, 4),
                     "algorithmic_flop": "150 per (model, correspondence) x ~6 models per iteration: the scoring only; the solvers are not counted"}}}
    if with_cpu:
        import oracle

        t0 = time.perf_counter()
        same = 0
        for p in range(24):
            sl = slice(off[p], off[p + 1])
            w = oracle.robust_match_calibrated_bearings(b1[sl], b2[sl], 0.004, 1000, 0.99, True, 10, 10)
            same += bool(np.array_equal(w["mask"], mask[sl]))
        dtc = time.perf_counter() - t0
        out["geometric_stage"]["cpu_baseline"] = {"value": round(24 / dtc, 2), "unit": "pairs/s", "cores": 1, "kind": "port",
                                                   "sample": f"24 pairs, {dtc:.1f} s, one thread", "parity_on_sample": f"{same}/24 identical inlier sets"}
    pairs = neighbour_pairs(n_images, args.overlap_neighbors)
    cam = SimpleNamespace(projection_type="perspective", k1=1e-3, k2=0.0, focal=0.85)
    cams = [cam] * n_images
    tm0 = MatchTimings()
    t0 = time.perf_counter()
    matching.match_pairs_calibrated(store, pairs, cams, timings=tm0)  # first call on this context: allocator cache and ShouldStop tables cold
    dt0 = time.perf_counter() - t0
    tm = MatchTimings()
    t0 = time.perf_counter()
    counts, m = matching.match_pairs_calibrated(store, pairs, cams, timings=tm)
    dt = time.perf_counter() - t0
    out["match_end_to_end"] = {
        "first_call_ms": round(1e3 * dt0, 2), "first_call_geometric_stage_ms": round(float(tm0.ms_ransac_kernel), 2),
        "workload": f"{len(pairs)} neighbour pairs of the {n_images} x {args.features} store, camera perspective k1 = 1e-3 (calibrated branch)",
        "value": round(len(pairs) / dt, 1), "unit": "pairs/s", "call_ms": round(1e3 * dt, 2), "match_kernel_ms": round(float(tm.ms_match_kernel), 2),
        "geometric_stage_ms": round(float(tm.ms_ransac_kernel), 2), "pairs_reaching_geometric_stage": int(tm.pairs_ransac),
        "pairs_with_matches": int((counts > 0).sum()), "inlier_matches_per_pair": round(float(counts.mean()), 1)}
    return out


def cpu_baseline(scene, pairs_all, n_sample, graph, full_parity=False):
    """The CPU oracle (a port of the reference's cv2 path, see oracle/*.c) timed on this box's host
    cores on a bounded, strided sample of the same pair list; also re-checks parity on the sample.
    --full-parity: additionally EVERY pair of the list that produced matches and 5000 random empty ones."""
    import oracle

    n_sample = min(n_sample, len(pairs_all))
    n_img = len(scene.offsets) - 1
    sample_note = "strided over the same pair list"
    extra = np.zeros(0, np.int64)  # pairs checked for parity but NOT part of the timed sample
    if n_img <= 2000 or graph is None:
        sel = np.linspace(0, len(pairs_all) - 1, n_sample).astype(np.int64)
    else:
        # large stores (configs[3]: 10 000 images = 10 GB as float32): runs of 64 consecutive pairs at strided anchors, so the
        # sample touches ~n_sample / 64 * 65 images, of which only those are converted.  The TIMED sample is representative of the
        # list (0.2 % of an exhaustive list has matches); pairs that produced matches are checked for parity in a second, separately
        # timed call so they do not bias the rate
        anchors = np.linspace(0, len(pairs_all) - 65, max(1, n_sample // 64)).astype(np.int64)
        sel = np.unique((anchors[:, None] + np.arange(64)[None, :]).reshape(-1))
        hit = np.flatnonzero(graph[0] > 0)
        n_hit = min(len(hit), n_sample // 4)
        extra = np.setdiff1d(hit[np.linspace(0, len(hit) - 1, n_hit).astype(np.int64)], sel) if n_hit else extra
        n_sample = len(sel)
        sample_note = f"{len(anchors)} runs of 64 consecutive pairs at strided anchors of the same pair list"

    def run(which):
        sample = pairs_all[which]
        used, inv = np.unique(sample.reshape(-1), return_inverse=True)  # only the images the sample touches, renumbered
        sample_c = inv.reshape(-1, 2).astype(np.int32)
        rows = np.concatenate([np.arange(scene.offsets[i], scene.offsets[i + 1]) for i in used])
        offs_c = np.concatenate([[0], np.cumsum([scene.offsets[i + 1] - scene.offsets[i] for i in used])]).astype(np.int64)
        desc = scene.desc[rows].astype(np.float32)
        pts_c = scene.pts[rows]
        t0 = time.perf_counter()
        res = oracle.match_pairs(desc, pts_c, offs_c, sample_c)
        return res, time.perf_counter() - t0

    res, dt = run(sel)
    ok = None
    n_with = int(sum(len(r) > 0 for r in res))
    if graph is not None:
        counts, matches = graph
        off = np.concatenate([[0], np.cumsum(counts, dtype=np.int64)])
        ok = all(np.array_equal(matches[off[p]: off[p + 1]], r) for p, r in zip(sel, res))
    out = {
        "value": round(n_sample / dt, 3),
        "unit": "pairs/s",
        "cores": oracle.num_threads(),
        "kind": "port",
        "sample": f"{n_sample} pairs {sample_note}, {dt:.1f} s, OpenMP over pairs",
        "parity_on_sample": ok,
        "sample_pairs_with_matches": n_with,
    }
    # a second CPU figure: the path someone would write for uchar descriptors on AVX hardware -- the distances in GEMM form (exact on
    # integer-valued levels), one blocked 2000 x 2000 x 128 product per pair serving both directions (oracle/match_oracle.c); descriptor
    # stage only, so it is set beside the direct form's descriptor stage on the same pairs.  Neither is cv2.
    try:
        sample = pairs_all[sel]
        used, inv = np.unique(sample.reshape(-1), return_inverse=True)
        rows = np.concatenate([np.arange(scene.offsets[i], scene.offsets[i + 1]) for i in used])
        offs_c = np.concatenate([[0], np.cumsum([scene.offsets[i + 1] - scene.offsets[i] for i in used])]).astype(np.int64)
        desc = scene.desc[rows].astype(np.float32)
        sample_c = inv.reshape(-1, 2).astype(np.int32)
        t0 = time.perf_counter()
        rg = oracle.match_pairs_gemm(desc, offs_c, sample_c)
        dtg = time.perf_counter() - t0
        t0 = time.perf_counter()
        rd = oracle.match_pairs(desc, scene.pts[rows], offs_c, sample_c, stage=0)
        dtd = time.perf_counter() - t0
        out["gemm_form"] = {"value": round(n_sample / dtg, 3), "unit": "pairs/s (descriptor stage)", "cores": oracle.num_threads(), "kind": "port",
                            "direct_form_descriptor_stage_pairs_per_s": round(n_sample / dtd, 3),
                            "identical_to_direct_form": bool(all(np.array_equal(a, b) for a, b in zip(rg, rd))),
                            "gflops_per_core": round(n_sample / dtg * 2.0 * float(np.mean(np.diff(scene.offsets))) ** 2 * 128 / 1e9 / oracle.num_threads(), 2),
                            "note": "4 x 64 register-blocked fp32 micro-kernel (GCC vector extensions, -march=native), top-2 with a squared-distance pre-test"}
    except Exception as exc:  # noqa: BLE001
        out["gemm_form"] = {"error": f"{type(exc).__name__}: {exc}"}
    if len(extra) and graph is not None:
        res_x, dt_x = run(extra)
        ok_x = all(np.array_equal(matches[off[p]: off[p + 1]], r) for p, r in zip(extra, res_x))
        out["parity_on_pairs_with_matches"] = {"pairs": int(len(extra)), "identical": bool(ok_x), "seconds": round(dt_x, 1),
                                               "note": "pairs that produced matches, strided over them; not part of the timed sample"}
    if full_parity and graph is not None:
        counts, matches = graph
        off = np.concatenate([[0], np.cumsum(counts)])
        hit = np.flatnonzero(counts > 0)
        rng = np.random.default_rng(5)
        empty = rng.choice(np.flatnonzero(counts == 0), min(5000, int((counts == 0).sum())), replace=False)
        chk = np.concatenate([hit, empty])
        t0 = time.perf_counter()
        bad = 0
        desc = scene.desc.astype(np.float32)
        for lo in range(0, len(chk), 2048):
            part = chk[lo: lo + 2048]
            res = oracle.match_pairs(desc, scene.pts, scene.offsets, pairs_all[part])
            bad += sum(not np.array_equal(matches[off[p]: off[p + 1]], r) for p, r in zip(part, res))
        out["full_parity"] = {"pairs_with_matches_checked": int(len(hit)), "empty_pairs_checked": int(len(empty)), "mismatches": int(bad),
                              "seconds": round(time.perf_counter() - t0, 1)}
    return out


if __name__ == "__main__":
    main()
