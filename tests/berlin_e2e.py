"""End-to-end check on the reference's example data set (data/berlin): images -> HAHOG -> matching -> tracks, compared with
data/berlin/tracks_example.csv -- the one artefact in the reference that real HAHOG + cv2 matching + track creation produced.

The stages are injected so that the same flow runs (a) on the CPU with the compiled reference HAHOG + the oracle matcher
(`tests/test_berlin_e2e.py`, where /root/reference exists) and (b) on the GPU with the product (`tests/test_gpu_berlin_e2e.py`).
Fixture: tests/golden/berlin_e2e.npz (generator: tests/golden/gen_berlin_e2e.py).
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "berlin_e2e.npz")
# OpenSfM defaults (opensfm/config.py): the data set's own config.yaml leaves these alone except feature_process_size, and the
# scales in tracks_example.csv say the file was produced at 2048
CONFIG = {"feature_min_frames": 4000, "hahog_peak_threshold": 1e-5, "hahog_edge_threshold": 10.0, "feature_root": True,
          "hahog_normalize_to_uchar": True, "lowes_ratio": 0.8, "symmetric_matching": True, "matcher_type": "BRUTEFORCE",
          "robust_matching_threshold": 0.004, "robust_matching_min_match": 20, "min_track_length": 2}
PAIRS = np.array([[0, 1], [0, 2], [1, 2]], np.int32)


def _decode_gray(delta: np.ndarray) -> np.ndarray:
    """inverse of the generator's row predictor (pixel - left neighbour, mod 256): a running sum along the row"""
    return (np.cumsum(delta.astype(np.int64), axis=2) % 256).astype(np.uint8)


def load():
    z = np.load(FIXTURE)
    gray = _decode_gray(z["gray_delta"])
    mask = np.unpackbits(z["mask_bits"], axis=2)[:, :, : int(z["mask_shape"][1])].astype(bool)
    ref = {"image": z["ref_image"], "track": z["ref_track"], "feature": z["ref_feature"], "xys": z["ref_xys"]}
    return gray, mask, ref


def finish_features(points: np.ndarray, desc: np.ndarray, width: int, height: int, mask: np.ndarray):
    """what follows the extractor in the reference: normalised coordinates (features.py:324-354), the mask look-up at upright pixel
    coordinates (masking.py:54-98, orientation 1: u = (x + 0.5) * mask_width, truncated), ascending size (features_processing.py:314-318)"""
    p = np.asarray(points, np.float64).copy()
    size = max(width, height)
    p[:, 0] = (p[:, 0] + 0.5 - width / 2.0) / size
    p[:, 1] = (p[:, 1] + 0.5 - height / 2.0) / size
    p[:, 2] /= size
    mh, mw = mask.shape
    # upright.opensfm_to_upright (upright.py:36-61) with orientation 1 and new size = the mask's; exif width / height = the mask's here
    fw, fh = float(mw), float(mh)
    s = max(fw, fh)
    u = (mw * (p[:, 0] * (s / fw) + 0.5)).astype(int)
    v = (mh * (p[:, 1] * (s / fh) + 0.5)).astype(int)
    keep = mask[np.clip(v, 0, mh - 1), np.clip(u, 0, mw - 1)]
    p, d = p[keep], np.asarray(desc)[keep]
    order = np.argsort(p[:, 2], kind="stable")
    return p[order], d[order]


def run(extract, match_pairs, create_tracks):
    """extract(gray_u8, config) -> (points px, descriptors); match_pairs(descs, points_norm_xy, pairs, config) -> list of (K, 2);
    create_tracks(edge_a, edge_b, node_offsets, min_length) -> (n_tracks, obs_track, obs_image, obs_feature)"""
    gray, mask, ref = load()
    feats = []
    for k in range(len(gray)):
        pts, desc = extract(gray[k], CONFIG)
        feats.append(finish_features(pts, desc, gray[k].shape[1], gray[k].shape[0], mask[k]))
    matches = match_pairs([f[1] for f in feats], [f[0][:, :2] for f in feats], PAIRS, CONFIG)
    off = np.concatenate([[0], np.cumsum([len(f[0]) for f in feats])]).astype(np.int64)
    ea = np.concatenate([off[PAIRS[p, 0]] + np.asarray(matches[p], np.int64).reshape(-1, 2)[:, 0] for p in range(len(PAIRS))])
    eb = np.concatenate([off[PAIRS[p, 1]] + np.asarray(matches[p], np.int64).reshape(-1, 2)[:, 1] for p in range(len(PAIRS))])
    ntr, ot, oi, of = create_tracks(ea.astype(np.int32), eb.astype(np.int32), off, CONFIG["min_track_length"])
    rep = compare(feats, (ntr, ot, oi, of), ref)
    rep["missing_edges"] = classify_missing_edges(feats, matches, (ntr, ot, oi, of), ref)
    return feats, matches, (ntr, ot, oi, of), rep


def _reference_candidates(feats, ref):
    """per row of the reference file: our features at its (x, y, scale) -- see compare()"""
    from scipy.spatial import cKDTree

    cands = [[] for _ in range(len(ref["track"]))]
    for k in range(len(feats)):
        sel = np.flatnonzero(ref["image"] == k)
        hits = cKDTree(feats[k][0][:, :2]).query_ball_point(ref["xys"][sel, :2], 1e-6)
        for r, h in zip(sel, hits):
            cands[r] = [int(i) for i in h if abs(feats[k][0][i, 2] - ref["xys"][r, 2]) < 1e-6]
    return cands


def classify_missing_edges(feats, matches, tracks, ref, ratio=None):
    """Why is an edge of data/berlin/tracks_example.csv (two rows of one of its tracks) not an edge of ours?  Every such edge gets the FIRST
    reason that applies, walking the pipeline in order:

      feature_missing    one of its two rows has no feature of ours at its (x, y, scale)                       [extraction]
      nn_differs         the exact mutual nearest neighbours (L2 over the 128 root-HAHOG levels) do not pair the two features: the
                         file's matcher (FLANN's randomised kd-forest at the time) returned a neighbour the exhaustive search does not,
                         or the edge is transitive in the file (the two images were linked through the third)  [approximate search]
      ratio_test         they ARE mutual nearest neighbours, but d1 < ratio * d2 fails in one direction with the exact second
                         neighbour (an approximate search that misses the true second neighbour passes it)       [approximate search]
      ransac_rejected    the pair passes the symmetric ratio test and the F-RANSAC restatement (oracle/ransac_oracle.c, on the GPU
                         ransac.hip) drops it -- the only share that speaks about the cv2.findFundamentalMat restatement [robust stage]
      track_stage        the match survives RANSAC and the edge is still missing: our track was dropped or split by the track
                         filter (an image twice in a track, tracking.py:100-115)                                 [tracks]

    Returns counts and shares over the missing edges (with both rows found / in all)."""
    ratio = CONFIG["lowes_ratio"] if ratio is None else ratio
    ntr, ot, oi, of = tracks
    cands = _reference_candidates(feats, ref)
    track_of = {(int(im), int(f)): int(t) for t, im, f in zip(ot, oi, of)}
    kept = {}
    for p, (a, b) in enumerate(PAIRS):
        kept[(int(a), int(b))] = set(map(tuple, np.asarray(matches[p], np.int64).reshape(-1, 2).tolist()))
    # exact top-2 in both directions for every image pair (fp32 squared distances on the integer-valued levels: exact)
    top = {}
    for a, b in PAIRS:
        A, B = feats[a][1].astype(np.float64), feats[b][1].astype(np.float64)
        d = (A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2.0 * (A @ B.T)
        for (x, y, dm) in ((int(a), int(b), d), (int(b), int(a), d.T)):
            idx = np.argpartition(dm, 1, axis=1)[:, :2]
            dd = np.take_along_axis(dm, idx, 1)
            sw = dd[:, 0] > dd[:, 1]
            idx[sw] = idx[sw][:, ::-1]
            dd[sw] = dd[sw][:, ::-1]
            top[(x, y)] = (idx[:, 0], np.sqrt(np.maximum(dd, 0.0)))
    out = {k: 0 for k in ("feature_missing", "nn_differs", "ratio_test", "ransac_rejected", "track_stage")}
    missing = 0
    for t in np.unique(ref["track"]):
        rows = np.flatnonzero(ref["track"] == t)
        for ia in range(len(rows)):
            for ib in range(ia + 1, len(rows)):
                ra, rb = rows[ia], rows[ib]
                ka, kb = int(ref["image"][ra]), int(ref["image"][rb])
                if ka == kb:
                    continue
                if ka > kb:
                    ra, rb, ka, kb = rb, ra, kb, ka
                ta = {track_of[(ka, f)] for f in cands[ra] if (ka, f) in track_of}
                tb = {track_of[(kb, f)] for f in cands[rb] if (kb, f) in track_of}
                if ta & tb:
                    continue
                missing += 1
                if not cands[ra] or not cands[rb]:
                    out["feature_missing"] += 1
                    continue
                (nn_ab, d_ab), (nn_ba, d_ba) = top[(ka, kb)], top[(kb, ka)]
                pairs = [(fa, fb) for fa in cands[ra] for fb in cands[rb]]
                mutual = [(fa, fb) for fa, fb in pairs if nn_ab[fa] == fb and nn_ba[fb] == fa]
                if not mutual:
                    out["nn_differs"] += 1
                    continue
                passing = [(fa, fb) for fa, fb in mutual if d_ab[fa, 0] < ratio * d_ab[fa, 1] and d_ba[fb, 0] < ratio * d_ba[fb, 1]]
                if not passing:
                    out["ratio_test"] += 1
                    continue
                if not any(pq in kept[(ka, kb)] for pq in passing):
                    out["ransac_rejected"] += 1
                    continue
                out["track_stage"] += 1
    found = missing - out["feature_missing"]
    rep = {"missing_edges": missing, "counts": dict(out), "share_of_missing": {k: v / max(1, missing) for k, v in out.items()},
           "share_of_missing_with_features": {k: v / max(1, found) for k, v in out.items() if k != "feature_missing"}}
    return rep


def compare(feats, tracks, ref):
    """fractions of the reference file reproduced: its feature rows (x, y, scale) among the extracted features, and its track edges
    (two observations of one track) among ours"""
    from scipy.spatial import cKDTree

    ntr, ot, oi, of = tracks
    out = {"features_per_image": [int(len(f[0])) for f in feats], "tracks": int(ntr), "observations": int(len(ot)),
           "ref_tracks": int(len(np.unique(ref["track"]))), "ref_observations": int(len(ref["track"]))}
    # a keypoint with several dominant orientations is several features at one (x, y, scale) (hahog.cc:98-123) and the file has no angle
    # column: a reference row maps to the SET of our features at its position
    cands = [[] for _ in range(len(ref["track"]))]
    for k in range(len(feats)):
        sel = np.flatnonzero(ref["image"] == k)
        hits = cKDTree(feats[k][0][:, :2]).query_ball_point(ref["xys"][sel, :2], 1e-6)
        for r, h in zip(sel, hits):
            cands[r] = [(k, int(i)) for i in h if abs(feats[k][0][i, 2] - ref["xys"][r, 2]) < 1e-6]
    found = np.array([len(c) > 0 for c in cands])
    out["ref_feature_rows_found"] = float(found.mean())
    out["ref_feature_rows_found_per_image"] = [float(found[ref["image"] == k].mean()) for k in range(len(feats))]
    track_of = {}
    for t, im, f in zip(ot, oi, of):
        track_of[(int(im), int(f))] = int(t)
    n_edges = n_hit = n_edges_found = 0
    ref_edges = set()
    for t in np.unique(ref["track"]):
        rows = np.flatnonzero(ref["track"] == t)
        for a in range(len(rows)):
            for b in range(a + 1, len(rows)):
                ra, rb = rows[a], rows[b]
                n_edges += 1
                if not cands[ra] or not cands[rb]:
                    continue
                n_edges_found += 1
                for ka in cands[ra]:
                    for kb in cands[rb]:
                        ref_edges.add((min(ka, kb), max(ka, kb)))
                ta = {track_of[ka] for ka in cands[ra] if ka in track_of}
                tb = {track_of[kb] for kb in cands[rb] if kb in track_of}
                n_hit += int(len(ta & tb) > 0)
    out["ref_track_edges"] = n_edges
    out["ref_track_edges_reproduced"] = n_hit / max(1, n_edges)
    out["ref_track_edges_reproduced_given_features"] = n_hit / max(1, n_edges_found)
    # the converse: our edges among the reference's
    by_track = {}
    for t, im, f in zip(ot, oi, of):
        by_track.setdefault(int(t), []).append((int(im), int(f)))
    mine = hit = 0
    for obs in by_track.values():
        for a in range(len(obs)):
            for b in range(a + 1, len(obs)):
                mine += 1
                hit += int((min(obs[a], obs[b]), max(obs[a], obs[b])) in ref_edges)
    out["our_track_edges"] = mine
    out["our_track_edges_in_ref"] = hit / max(1, mine)
    return out
