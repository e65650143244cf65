"""Host-side logic that needs no GPU: the neighbourhood selection and sub-problem extraction of the
local bundle adjustment (ba_helpers.cc:36-115,117-222) against a set-based transcription, and the
flattening of a match graph into the reference's union order."""
import numpy as np

from opensfm_amd import bundle, synthetic, tracking


def _direct_neighbors_sets(shots_of_point, points_of_shot, inside, min_common, max_neighbors):
    # BAHelpers::DirectShotNeighbors, ba_helpers.cc:68-115 (ties broken by lower shot id, as the adapter documents)
    points = set()
    for s in inside:
        points |= points_of_shot[s]
    common = {}
    for p in points:
        for s in shots_of_point[p]:
            if s not in inside:
                common[s] = common.get(s, 0) + 1
    pairs = sorted(common.items(), key=lambda kv: (-kv[1], kv[0]))
    out = set()
    for idx, (s, n) in enumerate(pairs):
        if n >= min_common and idx < min(max_neighbors, len(pairs)):
            out.add(s)
        else:
            break
    return out


def _neighborhood_sets(pr, central, radius, min_common, max_interior):
    shots_of_point, points_of_shot = {}, {}
    for s, p in zip(pr["obs_shot"], pr["obs_point"]):
        shots_of_point.setdefault(int(p), set()).add(int(s))
        points_of_shot.setdefault(int(s), set()).add(int(p))
    interior = {central}
    distance = 1
    while distance < radius and len(interior) < max_interior:
        remaining = max_interior - len(interior)
        interior |= _direct_neighbors_sets(shots_of_point, points_of_shot, interior, min_common, remaining)
        distance += 1
    boundary = _direct_neighbors_sets(shots_of_point, points_of_shot, interior, 1, 1000000)
    return interior, boundary


def test_shot_neighborhood_matches_set_transcription():
    pr = synthetic.make_ba_scene(50, 1200, 7, seed=5)
    for central, radius, mc, mx in ((25, 3, 20, 30), (0, 2, 5, 4), (49, 4, 50, 12), (10, 1, 20, 30)):
        i_m, b_m = bundle.shot_neighborhood(pr, central, radius, mc, mx)
        i_s, b_s = _neighborhood_sets(pr, central, radius, mc, mx)
        assert set(np.flatnonzero(i_m)) == i_s and set(np.flatnonzero(b_m)) == b_s
        assert not (i_m & b_m).any() and i_m[central]


def test_local_problem_structure():
    pr = synthetic.make_ba_scene(40, 900, 6, seed=6)
    sub, shot_ids, pt_ids, interior, boundary = bundle.local_problem(pr, 20, {"local_bundle_max_shots": 7})
    assert sub["cam_fixed"].all()  # constexpr bool fix_cameras{true}, ba_helpers.cc:137
    assert np.array_equal(sub["shot_fixed"].astype(bool), boundary[shot_ids])  # boundary instances are constant
    # every point seen from the interior is in, with all its observations from interior and boundary shots
    seen = np.unique(pr["obs_point"][interior[pr["obs_shot"]]])
    assert np.array_equal(pt_ids, seen)
    n_expected = int((np.isin(pr["obs_point"], seen) & (interior | boundary)[pr["obs_shot"]]).sum())
    assert len(sub["obs_shot"]) == n_expected
    assert (sub["shot_gps_sigma"][sub["shot_fixed"].astype(bool)] == 0).all()  # position priors on interior shots only
    assert np.array_equal(sub["shot_pose"], pr["shot_pose"][shot_ids])


def test_edges_follow_the_reference_union_order():
    pairs = np.array([[0, 2], [1, 2], [0, 1]], np.int32)
    counts = np.array([2, 0, 3], np.int32)
    matches = np.array([[5, 7], [1, 0], [2, 2], [3, 9], [0, 4]], np.int32)
    off = np.array([0, 10, 20, 30], np.int64)
    ea, eb = tracking.edges_from_match_graph(pairs, counts, matches, off)
    assert list(ea) == [5, 1, 2, 3, 0] and list(eb) == [27, 20, 12, 19, 14]


def test_shot_renumbering_restores_the_band_of_a_shuffled_sequence():
    """osfm_ba_shot_order is host-only code of the library (reverse Cuthill-McKee on the co-visibility graph): a
    shuffled street sequence gets its narrow band back; the result is a permutation."""
    import ctypes as C

    from opensfm_amd import _lib
    from opensfm_amd._ba_abi import BaProblem

    lib = _lib.load()
    pr = synthetic.make_ba_scene(300, 6000, 8, seed=9)
    rng = np.random.default_rng(2)
    inv = np.argsort(rng.permutation(300))
    obs_shot = np.ascontiguousarray(inv[pr["obs_shot"]], np.int32)
    obs_point = np.ascontiguousarray(pr["obs_point"], np.int32)
    P = BaProblem()
    P.n_cameras, P.n_shots, P.n_points, P.n_obs = 1, 300, 6000, len(obs_shot)
    P.obs_shot = obs_shot.ctypes.data_as(C.POINTER(C.c_int32))
    P.obs_point = obs_point.ctypes.data_as(C.POINTER(C.c_int32))
    order = np.zeros(300, np.int32)
    b0, b1 = C.c_int32(), C.c_int32()
    rc = lib.osfm_ba_shot_order(C.byref(P), order.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(b0), C.byref(b1))
    assert rc == 0
    assert b0.value > 150 and b1.value <= 10
    assert np.array_equal(np.sort(order), np.arange(300))
    new_shot = order[obs_shot]
    span = np.zeros(6000, np.int64)
    for p in range(6000):
        s = new_shot[obs_point == p]
        span[p] = s.max() - s.min()
    assert span.max() == b1.value


def test_shot_renumbering_sweeps_a_block_survey_along_its_long_side():
    """A rows x cols block survey numbered line after line has a co-visibility half-width of ~2 cols; breadth-first levels from a
    corner are diagonals and do not narrow it, the sweep along the long side (bins of one shot spacing along the principal axis of
    the shot positions, the second axis inside a bin) does: ~2 rows."""
    import ctypes as C

    from opensfm_amd import _lib
    from opensfm_amd._ba_abi import BaProblem

    lib = _lib.load()
    rows, cols = 12, 40
    pr = synthetic.make_ba_scene_grid(rows, cols, 4000, 9, seed=3)
    obs_shot = np.ascontiguousarray(pr["obs_shot"], np.int32)
    obs_point = np.ascontiguousarray(pr["obs_point"], np.int32)
    pose = np.ascontiguousarray(pr["shot_pose"], np.float64)
    P = BaProblem()
    P.n_cameras, P.n_shots, P.n_points, P.n_obs = 1, rows * cols, 4000, len(obs_shot)
    P.obs_shot = obs_shot.ctypes.data_as(C.POINTER(C.c_int32))
    P.obs_point = obs_point.ctypes.data_as(C.POINTER(C.c_int32))
    P.shot_pose = pose.ctypes.data_as(C.POINTER(C.c_double))
    order = np.zeros(rows * cols, np.int32)
    b0, b1 = C.c_int32(), C.c_int32()
    assert lib.osfm_ba_shot_order(C.byref(P), order.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(b0), C.byref(b1)) == 0
    assert b0.value == 2 * cols + 2
    assert b1.value <= 3 * rows, (b0.value, b1.value)  # 2 rows + 2 when every line is recognised, 3 rows - 1 for a plain sweep
    assert np.array_equal(np.sort(order), np.arange(rows * cols))


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under opensfm_amd/ (Python or HIP sources) may import, include or load it."""
    import os
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opensfm_amd")
    offenders = []
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+oracle\b|liboracle|oracle/[a-z_]+\.(c|so)\b.*#include|#include\s+\"[^\"]*oracle", text, re.M):
                    offenders.append(os.path.join(dirpath, f))
    assert offenders == []


def test_match_images_is_preselection_then_batch(monkeypatch):
    """matching.match_images (matching.py:27-60): exifs of every image, match_candidates_from_metadata, match_images_with_pairs"""
    import types

    import numpy as np

    from opensfm_amd import matching, preselection

    calls = {}

    def from_metadata(ref, cand, exifs, data, cfg, **kw):
        calls["pre"] = (tuple(ref), tuple(cand), sorted(exifs), kw)
        return [("a", "b")], {"num_pairs_order": 1}

    def with_pairs(data, cfg, exifs, pairs, poses=None):
        calls["match"] = (list(pairs), sorted(exifs))
        return {("a", "b"): np.zeros((3, 2), int)}

    monkeypatch.setattr(preselection, "match_candidates_from_metadata", from_metadata)
    monkeypatch.setattr(matching, "match_images_with_pairs", with_pairs)
    data = types.SimpleNamespace(load_exif=lambda im: {"camera": im})
    m, rep = matching.match_images(data, {"matching_order_neighbors": 2}, ["a"], ["a", "b"], bow_histograms={})
    assert calls["pre"][:3] == (("a",), ("a", "b"), ["a", "b"]) and calls["pre"][3] == {"bow_histograms": {}}
    assert calls["match"] == ([("a", "b")], ["a", "b"]) and rep == {"num_pairs_order": 1} and list(m) == [("a", "b")]


def test_bench_gpus_flag_launches_ranks_and_never_mislabels(monkeypatch):
    """bench.py --gpus N: without a launcher it re-executes itself under torch.distributed.run with N ranks; under a launcher whose
    WORLD_SIZE differs from --gpus it refuses to print a line (`--gpus 8` used to be a silent 1-GPU run)"""
    import importlib
    import os
    import subprocess
    import sys
    import types

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.self_launch(types.SimpleNamespace(gpus=4)) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a launcher that started 2 ranks for --gpus 4: assertion before anything is measured
    monkeypatch.setenv("WORLD_SIZE", "2")
    import pytest

    with pytest.raises(AssertionError, match="--gpus 4 but WORLD_SIZE=2"):
        bench.main()


def test_bench_line_at_n_above_one_is_the_headline_workload_only():
    """the secondary workloads and the CPU baselines are one-GPU / host measurements: reported at N = 1, skipped on rank 0 at N > 1"""
    import importlib
    import os
    import sys
    import types

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    flags = ("no_overlap", "no_calibrated", "no_float", "no_guided", "no_cpu_baseline", "no_tracks", "no_hahog", "no_ba")
    a = types.SimpleNamespace(all_sections=False, **{f: False for f in flags})
    assert not bench.headline_only_for_ranks(a, 1) and not any(getattr(a, f) for f in flags)
    assert bench.headline_only_for_ranks(a, 8) and all(getattr(a, f) for f in flags)
    b = types.SimpleNamespace(all_sections=True, **{f: False for f in flags})
    assert not bench.headline_only_for_ranks(b, 8) and not any(getattr(b, f) for f in flags)
