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
    return feats, matches, (ntr, ot, oi, of), compare(feats, (ntr, ot, oi, of), ref)


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
