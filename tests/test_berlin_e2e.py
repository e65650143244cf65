"""data/berlin end to end on the CPU checkers: the reference's HAHOG compiled from /root/reference (oracle/_ref/libhahog_ref.so) + the
matching / RANSAC / tracks oracles, against data/berlin/tracks_example.csv.  This is the outside evidence for the cv2-shaped restatements
(oracle/match_oracle.c, oracle/ransac_oracle.c): the file was written by real HAHOG + cv2 matching (FLANN at the time, whose search is
approximate, and cv2's RANSAC) + the reference's track creation.  The same flow runs on the GPU in tests/test_gpu_berlin_e2e.py."""
import numpy as np
import pytest

import berlin_e2e
import oracle

# REGRESSION GUARDS, not parity statements: measured with this flow (tests/berlin_e2e.py) 0.981 / 0.887 / 0.867, set a margin below.
# What the 11.3 % of the file's track edges that are NOT reproduced (222 of 1 971) are made of is measured by
# berlin_e2e.classify_missing_edges and asserted below: 67 an endpoint feature absent (extraction), 51 + 74 the exact mutual nearest
# neighbour / the exact ratio test disagree with what the file's approximate (FLANN) search returned, 29 rejected by the F-RANSAC
# restatement, 1 lost in the track filter.  The robust stage -- the cv2.findFundamentalMat-shaped part -- accounts for 13 % of the gap
# = 1.5 % of the file's edges; the descriptor stage's share is the approximate search's, not the float accumulation order's (the
# root-HAHOG levels are integers: every order gives the same distances).
MIN_FEATURE_ROWS, MIN_REF_EDGES, MIN_OUR_EDGES = 0.975, 0.87, 0.85
MAX_RANSAC_SHARE_OF_GAP, MAX_RANSAC_SHARE_OF_EDGES = 0.20, 0.02


def oracle_extract(gray, cfg):
    pts, desc = oracle.hahog_ref(gray.astype(np.float32) / 255, cfg["hahog_peak_threshold"], cfg["hahog_edge_threshold"], cfg["feature_min_frames"])
    return pts, (362 * np.sqrt(desc)).clip(0, 255).round().astype(np.float32)  # features.py:526-534


def oracle_match_pairs(descs, xys, pairs, cfg):
    off = np.concatenate([[0], np.cumsum([len(d) for d in descs])]).astype(np.int64)
    return oracle.match_pairs(np.concatenate(descs), np.concatenate(xys), off, pairs, ratio=cfg["lowes_ratio"],
                              min_match=cfg["robust_matching_min_match"], thr=cfg["robust_matching_threshold"])


def check_report(rep):
    assert rep["ref_feature_rows_found"] >= MIN_FEATURE_ROWS, rep
    assert rep["ref_track_edges_reproduced"] >= MIN_REF_EDGES, rep
    assert rep["our_track_edges_in_ref"] >= MIN_OUR_EDGES, rep
    assert abs(rep["tracks"] - rep["ref_tracks"]) <= 0.05 * rep["ref_tracks"], rep
    gap = rep["missing_edges"]
    assert sum(gap["counts"].values()) == gap["missing_edges"] == round((1 - rep["ref_track_edges_reproduced"]) * rep["ref_track_edges"]), gap
    assert gap["share_of_missing"]["ransac_rejected"] <= MAX_RANSAC_SHARE_OF_GAP, gap
    assert gap["counts"]["ransac_rejected"] <= MAX_RANSAC_SHARE_OF_EDGES * rep["ref_track_edges"], gap
    assert gap["counts"]["track_stage"] <= 5, gap


@pytest.mark.skipif(oracle.build_hahog_ref() is None, reason="needs the reference HAHOG compiled from /root/reference")
def test_berlin_tracks_example_is_reproduced_by_the_checkers():
    feats, matches, tracks, rep = berlin_e2e.run(oracle_extract, oracle_match_pairs, oracle.tracks)
    print(rep)
    check_report(rep)


def _rodrigues(r):
    r = np.asarray(r, float)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


@pytest.mark.skipif(oracle.build_hahog_ref() is None, reason="needs the reference HAHOG compiled from /root/reference")
def test_ransac_inliers_agree_with_the_reference_reconstruction():
    """Second piece of outside evidence, for the robust stage: data/berlin/reconstruction_example.json (tests/golden/berlin_example.json) holds the
    camera and the three poses the reference's own pipeline converged to (cv2 matching, Ceres).  The matches the F-RANSAC restatement keeps
    must lie on the epipolar planes of THAT geometry -- angle between the second bearing and the plane spanned by the baseline and the first
    bearing --, and the matches it rejects mostly must not.  Measured: 99.9 / 92.2 / 90.8 % of the inliers within 0.006 rad (medians
    0.4 - 0.9 mrad), 42 / 0 / 22 % of the rejected ones."""
    import json
    import os

    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "berlin_example.json")))
    cam = list(rec["cameras"].values())[0]
    names = ["01.jpg", "02.jpg", "03.jpg"]
    pose = {n: (_rodrigues(rec["shots"][n]["rotation"]), np.asarray(rec["shots"][n]["translation"], float)) for n in names}
    gray, mask, _ = berlin_e2e.load()
    feats = []
    for k in range(3):
        pts, desc = oracle_extract(gray[k], berlin_e2e.CONFIG)
        feats.append(berlin_e2e.finish_features(pts, desc, gray[k].shape[1], gray[k].shape[0], mask[k]))
    descs, xys = [f[1] for f in feats], [f[0][:, :2] for f in feats]
    off = np.concatenate([[0], np.cumsum([len(d) for d in descs])]).astype(np.int64)
    cfg = berlin_e2e.CONFIG
    raw = oracle.match_pairs(np.concatenate(descs), np.concatenate(xys), off, berlin_e2e.PAIRS, ratio=cfg["lowes_ratio"],
                             min_match=cfg["robust_matching_min_match"], thr=cfg["robust_matching_threshold"], stage=0)
    kept = oracle_match_pairs(descs, xys, berlin_e2e.PAIRS, cfg)

    def bearings(xy):
        return oracle.pixel_bearings("perspective", np.array([cam["k1"], cam["k2"], cam["focal"]]), np.ascontiguousarray(xy, np.float64))

    for p, (a, b) in enumerate(berlin_e2e.PAIRS):
        (Ra, ta), (Rb, tb) = pose[names[a]], pose[names[b]]
        R = Rb @ Ra.T
        t = tb - R @ ta  # x_b = R x_a + t

        def angle(m):
            b1, b2 = bearings(xys[a][m[:, 0]]), bearings(xys[b][m[:, 1]])
            n = np.cross(t / np.linalg.norm(t), b1 @ R.T)
            n /= np.linalg.norm(n, axis=1, keepdims=True)
            return np.abs(np.arcsin(np.clip((b2 * n).sum(1), -1, 1)))

        mi, mr = np.asarray(kept[p]).reshape(-1, 2), np.asarray(raw[p]).reshape(-1, 2)
        si = set(map(tuple, mi.tolist()))
        rej = np.array([x for x in mr.tolist() if tuple(x) not in si]).reshape(-1, 2)
        ei = angle(mi)
        assert len(mi) >= 200 and np.median(ei) < 1.5e-3 and (ei < 0.006).mean() >= 0.88, (a, b, len(mi), np.median(ei), (ei < 0.006).mean())
        if len(rej) >= 10:
            assert (angle(rej) < 0.006).mean() <= 0.6, (a, b, len(rej))


@pytest.mark.skipif(oracle.build_hahog_ref() is None, reason="needs the reference HAHOG compiled from /root/reference")
def test_calibrated_relative_poses_agree_with_the_reference_reconstruction():
    """Third piece, for the calibrated branch (five-point LO-RANSAC + the three refinement stages, oracle/relpose_oracle.c): the relative pose
    it returns for every pair of data/berlin against the relative pose of the reference's converged reconstruction (Ceres over cv2's matches,
    with that reconstruction's intrinsics).  Measured: rotations within 0.25 - 1.25 degrees, translation directions within 0.14 - 3.0 degrees,
    94 - 99 % of the descriptor matches kept."""
    import json
    import os

    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "berlin_example.json")))
    cam = list(rec["cameras"].values())[0]
    c = np.array([cam["k1"], cam["k2"], cam["focal"]])
    names = ["01.jpg", "02.jpg", "03.jpg"]
    pose = {n: (_rodrigues(rec["shots"][n]["rotation"]), np.asarray(rec["shots"][n]["translation"], float)) for n in names}
    gray, mask, _ = berlin_e2e.load()
    feats = []
    for k in range(3):
        pts, desc = oracle_extract(gray[k], berlin_e2e.CONFIG)
        feats.append(berlin_e2e.finish_features(pts, desc, gray[k].shape[1], gray[k].shape[0], mask[k]))
    descs, xys = [f[1] for f in feats], [f[0][:, :2] for f in feats]
    off = np.concatenate([[0], np.cumsum([len(d) for d in descs])]).astype(np.int64)
    raw = oracle.match_pairs(np.concatenate(descs), np.concatenate(xys), off, berlin_e2e.PAIRS, ratio=0.8, min_match=20, thr=0.004, stage=0)
    for p, (a, b) in enumerate(berlin_e2e.PAIRS):
        m = np.asarray(raw[p]).reshape(-1, 2)
        b1 = oracle.pixel_bearings("perspective", c, np.ascontiguousarray(xys[a][m[:, 0]]))
        b2 = oracle.pixel_bearings("perspective", c, np.ascontiguousarray(xys[b][m[:, 1]]))
        r = oracle.robust_match_calibrated_bearings(b1, b2, 0.004)
        (Ra, ta), (Rb, tb) = pose[names[a]], pose[names[b]]
        Rrel = Ra @ Rb.T  # x_a = Rrel x_b + trel: the pose of the second camera in the first, as robust_match_calibrated returns it
        trel = ta - Rrel @ tb
        rot = np.degrees(np.arccos(np.clip((np.trace(r["R"] @ Rrel.T) - 1) / 2, -1, 1)))
        tdir = np.degrees(np.arccos(np.clip(abs(r["t"] @ trel) / np.linalg.norm(r["t"]) / np.linalg.norm(trel), -1, 1)))
        assert rot < 2.0 and tdir < 5.0 and r["mask"].mean() > 0.9, (a, b, rot, tdir, r["mask"].mean())
