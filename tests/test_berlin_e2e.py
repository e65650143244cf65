"""data/berlin end to end on the CPU checkers: the reference's HAHOG compiled from /root/reference (oracle/_ref/libhahog_ref.so) + the
matching / RANSAC / tracks oracles, against data/berlin/tracks_example.csv.  This is the outside evidence for the cv2-shaped restatements
(oracle/match_oracle.c, oracle/ransac_oracle.c): the file was written by real HAHOG + cv2 matching (FLANN at the time, whose search is
approximate, and cv2's RANSAC) + the reference's track creation.  The same flow runs on the GPU in tests/test_gpu_berlin_e2e.py."""
import numpy as np
import pytest

import berlin_e2e
import oracle

# measured with this flow (tests/berlin_e2e.py): 0.981 / 0.887 / 0.867; the exact matcher cannot reproduce FLANN's misses
MIN_FEATURE_ROWS, MIN_REF_EDGES, MIN_OUR_EDGES = 0.975, 0.87, 0.85


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


@pytest.mark.skipif(oracle.build_hahog_ref() is None, reason="needs the reference HAHOG compiled from /root/reference")
def test_berlin_tracks_example_is_reproduced_by_the_checkers():
    feats, matches, tracks, rep = berlin_e2e.run(oracle_extract, oracle_match_pairs, oracle.tracks)
    print(rep)
    check_report(rep)
