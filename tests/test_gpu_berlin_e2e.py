"""data/berlin end to end on the MI355X: osfm_hahog_extract -> osfm_match_pairs -> osfm_tracks_create on the reference's three example
images, (a) against data/berlin/tracks_example.csv (real HAHOG + cv2 + track creation: the fractions of its feature rows and track edges
that come out) and (b) stage by stage against the CPU checkers run on the same inputs (compiled reference HAHOG, oracle matcher, oracle
tracks): identical features, matches and tracks."""
import numpy as np
import pytest

import berlin_e2e
import oracle
from test_berlin_e2e import check_report, oracle_extract, oracle_match_pairs

pytestmark = pytest.mark.gpu


def gpu_extract(gray, cfg):
    from opensfm_amd import features

    return features.extract_features_hahog(gray, cfg, cfg["feature_min_frames"])


def gpu_match_pairs(descs, xys, pairs, cfg):
    from opensfm_amd import matching

    store = matching.DescriptorStore(descs, xys)
    try:
        counts, matches = matching.match_pairs(store, pairs, cfg)
    finally:
        store.close()
    return matching.split_matches(counts, matches)


def gpu_tracks(ea, eb, off, min_length):
    from opensfm_amd import tracking

    return tracking.create_tracks_arrays(ea, eb, off, min_length)


def test_berlin_end_to_end_on_the_gpu():
    feats, matches, tracks, rep = berlin_e2e.run(gpu_extract, gpu_match_pairs, gpu_tracks)
    print(rep)
    check_report(rep)
    if oracle.build_hahog_ref() is None:
        pytest.skip("reference HAHOG not built: the comparison with tracks_example.csv passed, the stage-by-stage one needs oracle/_ref")
    feats_o, matches_o, tracks_o, rep_o = berlin_e2e.run(oracle_extract, oracle_match_pairs, oracle.tracks)
    for k, (f, fo) in enumerate(zip(feats, feats_o)):
        assert f[0].shape == fo[0].shape, (k, f[0].shape, fo[0].shape)
        assert np.array_equal(f[0][:, :3], fo[0][:, :3]), f"image {k}: x, y, size differ"
        assert np.abs(f[0][:, 3] - fo[0][:, 3]).max() <= 1e-4  # atan2f of the device library against glibc's, degrees
        assert np.array_equal(f[1], fo[1]), f"image {k}: descriptors differ"
    for p, (m, mo) in enumerate(zip(matches, matches_o)):
        assert np.array_equal(np.asarray(m).reshape(-1, 2), np.asarray(mo).reshape(-1, 2)), f"pair {p}: matches differ"
    assert tracks[0] == tracks_o[0]
    for a, b in zip(tracks[1:], tracks_o[1:]):
        assert np.array_equal(a, b)
    assert rep == rep_o
