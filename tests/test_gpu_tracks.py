"""GPU parity: tracks creation (connected components + _good_track) == CPU oracle, observation by
observation, including the reference's track numbering and member order."""
import numpy as np
import pytest

from test_oracle_tracks import random_match_graph, reference_tracks, to_edges

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_images,n_feat,n_pairs,per_pair,min_length,seed",
                         [(6, 40, 10, 15, 2, 0), (12, 100, 40, 30, 2, 1), (20, 60, 120, 25, 3, 2), (40, 500, 400, 200, 2, 3),
                          (100, 2000, 1500, 600, 2, 4)])
def test_tracks_equal_oracle(oracle_lib, gpu_ctx, n_images, n_feat, n_pairs, per_pair, min_length, seed):
    from opensfm_amd import tracking

    rng = np.random.default_rng(seed)
    matches = random_match_graph(rng, n_images, n_feat, n_pairs, per_pair)
    ea, eb, off = to_edges(matches, n_images, n_feat)
    nt_o, ot_o, oi_o, of_o = oracle_lib.tracks(ea, eb, off, min_length)
    nt_g, ot_g, oi_g, of_g = tracking.create_tracks_arrays(ea, eb, off, min_length)
    assert nt_g == nt_o
    assert np.array_equal(ot_g, ot_o) and np.array_equal(oi_g, oi_o) and np.array_equal(of_g, of_o)


def test_create_tracks_manager_signature(oracle_lib, gpu_ctx):
    """Same call as tracking.create_tracks_manager (tracking.py:68-78) on a dict of matches."""
    from opensfm_amd import tracking

    rng = np.random.default_rng(9)
    m = random_match_graph(rng, 8, 50, 20, 20)
    names = [f"im{i:02d}.jpg" for i in range(8)]
    matches = {(names[a], names[b]): v for (a, b), v in m.items()}
    features = {n: np.zeros((50, 3)) for n in names}
    tm = tracking.create_tracks_manager(features, {}, {}, {}, matches, 2, {})
    ref = reference_tracks(matches, 2)
    assert tm.num_tracks() == len(ref)
    for k, t in enumerate(ref):
        assert tm.get_track_observations(str(k)) == {im: f for im, f in t}


def test_tracks_edge_cases(gpu_ctx):
    from opensfm_amd import tracking

    off = np.array([0, 5, 10, 15], np.int64)
    nt, ot, oi, of = tracking.create_tracks_arrays(np.zeros(0, np.int32), np.zeros(0, np.int32), off, 2)
    assert nt == 0 and len(ot) == 0
    nt, ot, oi, of = tracking.create_tracks_arrays(np.array([1, 7]), np.array([7, 13]), off, 2)
    assert nt == 1 and list(oi) == [0, 1, 2] and list(of) == [1, 2, 3]
    nt, *_ = tracking.create_tracks_arrays(np.array([1, 7, 4]), np.array([7, 13, 13]), off, 2)
    assert nt == 0


@pytest.mark.gpu
def test_gpu_reproduces_the_reference_tracks():
    """The committed output of the reference's own create_tracks_manager (tests/golden/tracks_golden.json)."""
    import test_oracle_tracks as t

    from opensfm_amd import tracking

    t.check_against_golden(tracking.create_tracks_manager)
