"""The batch seam on the device: ``match_images_with_pairs`` (matching.py:63-98) with and without poses (guided matching) and with the
ad-hoc filters between the descriptor and the robust stage (matching.py:323-334, 399-411), and ``match_images`` (matching.py:27-60).

What the device returns is compared with the same Python flow running on the host emulations of every C-ABI call
(tests/test_reference_flow.py::emulate_product_leaves) -- which tests/test_reference_flow.py shows, in the build container, to equal
the reference's own matching.py executed from /root/reference for the same collection.  (/root/reference is not on the GPU box.)"""
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rows(a):
    a = np.asarray(a).reshape(-1, 2)
    return a[np.lexsort((a[:, 1], a[:, 0]))]


def _data(oracle_lib, guided, filters, seed):
    import test_reference_flow as rf

    rng = np.random.default_rng(seed)
    images, cams, cam_of, feats, masks, poses, config = rf._collection(oracle_lib, rng, guided)
    config["matching_use_filters"] = filters
    exif_of = {"a": {"make": "BlackVue", "model": "DR900"}, "b": {"make": "Canon", "model": "X"}, "c": {"make": "VTrans_Camera", "model": "VTrans_Camera"},
               "d": {"make": "blackvue", "model": "x"}}
    for im in images:
        exif_of[im]["camera"] = cam_of[im]
    data = types.SimpleNamespace(config=config, load_camera_models=lambda: cams, load_features=lambda im: feats[im],
                                 load_features_mask=lambda im, pts: masks[im], load_exif=lambda im: exif_of[im], load_reference=lambda: None)
    exifs = {im: {"camera": cam_of[im]} for im in images}
    pairs = [(a, b) for i, a in enumerate(images) for b in images[i + 1:]]
    return data, exifs, pairs, poses, images


@pytest.mark.parametrize("guided,filters", [(False, False), (True, False), (False, True), (True, True)])
def test_match_images_with_pairs_on_the_device(oracle_lib, gpu_ctx, monkeypatch, guided, filters):
    import test_reference_flow as rf
    from opensfm_amd import matching as product

    data, exifs, pairs, poses, _ = _data(oracle_lib, guided, filters, 17 if guided else 16)
    got = product.match_images_with_pairs(data, {}, exifs, pairs, poses if guided else None)
    with monkeypatch.context() as mp:
        rf.emulate_product_leaves(mp, oracle_lib)
        want = product.match_images_with_pairs(data, {}, exifs, pairs, poses if guided else None)
    survivors = 0
    for pair in pairs:
        assert np.array_equal(_rows(got[pair]), _rows(want[pair])), pair
        survivors += len(want[pair]) > 0
    assert survivors >= 5


def test_match_images_on_the_device(oracle_lib, gpu_ctx, monkeypatch):
    """preselection (order neighbours: no device search needed beyond the batch itself) + the batch, as match_images composes them"""
    import test_reference_flow as rf
    from opensfm_amd import matching as product

    data, exifs, pairs, poses, images = _data(oracle_lib, False, False, 21)
    override = {"matching_order_neighbors": 2, "matching_gps_distance": 0, "matching_gps_neighbors": 0, "matching_time_neighbors": 0,
                "matching_bow_neighbors": 0, "matching_vlad_neighbors": 0, "matching_graph_rounds": 0}
    got, report = product.match_images(data, override, images, images, bow_histograms={})
    with monkeypatch.context() as mp:
        rf.emulate_product_leaves(mp, oracle_lib)
        want, _ = product.match_images(data, override, images, images, bow_histograms={})
    assert set(got) == set(want) == {("a", "b"), ("b", "c"), ("c", "d")}  # order neighbours 2: one position either side
    for pair in want:
        assert np.array_equal(_rows(got[pair]), _rows(want[pair])), pair
    assert report["num_pairs_order"] == 3


@pytest.mark.parametrize("guided", [False, True])
def test_lmeds_branch_in_reach_goes_through_the_leaf(oracle_lib, gpu_ctx, monkeypatch, guided):
    """robust_matching_min_match = 8 and an image with a dozen features: the pinhole pair reaches cv2.findFundamentalMat's LMedS branch
    (8 <= n < 15), which the batched launch does not implement -- the batch path sends those pairs' robust stage through the leaf
    (osfm_ransac_fundamental; rounds 1-5 raised NotImplementedError).  Device flow = host-emulated flow, pair for pair."""
    import test_reference_flow as rf
    from opensfm_amd import matching as product

    data, exifs, pairs, poses, _ = _data(oracle_lib, guided, False, 17 if guided else 16)
    data.config["robust_matching_min_match"] = 8
    m = data.load_features_mask("b", None)
    keep = np.flatnonzero(m)[:12]
    m[:] = False
    m[keep] = True
    got = product.match_images_with_pairs(data, {}, exifs, pairs, poses if guided else None)
    with monkeypatch.context() as mp:
        rf.emulate_product_leaves(mp, oracle_lib)
        want = product.match_images_with_pairs(data, {}, exifs, pairs, poses if guided else None)
    for pair in pairs:
        assert np.array_equal(_rows(got[pair]), _rows(want[pair])), pair
    if not guided:
        assert 8 <= len(want["a", "b"]) < 15
