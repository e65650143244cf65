"""HAHOG extraction on the GPU (csrc/hahog.hip) against the REFERENCE's own features::hahog -- opensfm/src/features/src/hahog.cc over the
vendored vlfeat, compiled from /root/reference into oracle/_ref/libhahog_ref.so (oracle.hahog_ref) -- and against the golden vectors
that library produced in the build container (tests/golden/hahog_berlin01.npz, made by tests/golden/make_hahog_golden.py from the
reference's data/berlin/images/01.jpg).

Stated tolerance.  The set of features, their order, x, y, size and all 128 descriptor values must be the reference's BIT FOR BIT (the
kernels follow vlfeat operation for operation): the assertions are exact -- 0 differing keypoint rows, 0 differing descriptor values,
0 differing uint8 levels.  The one exception is the orientation angle in degrees (column 3 of the points), which goes through atan2f,
whose last bit differs between glibc and the device library: |angle difference| <= 1e-4 degrees."""
import os

import numpy as np
import pytest

from opensfm_amd import features

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hahog_berlin01.npz")
CFG = {"feature_root": True, "hahog_normalize_to_uchar": True, "hahog_peak_threshold": 1e-5, "hahog_edge_threshold": 10}


def _compare(pts, desc, rp, rd):
    assert pts.shape == rp.shape and desc.shape == rd.shape, (pts.shape, rp.shape)
    if len(pts) == 0:
        return
    bad_rows = int((pts[:, :3] != rp[:, :3]).any(axis=1).sum())
    bad_vals = int((desc != rd).sum())
    print("hahog parity: %d features, %d differing keypoint rows, %d differing descriptor values of %d" % (len(pts), bad_rows, bad_vals, desc.size))
    assert bad_rows == 0 and bad_vals == 0, (bad_rows, bad_vals)
    dang = np.abs(pts[:, 3] - rp[:, 3])
    dang = np.minimum(dang, 360.0 - dang)
    assert dang.max() <= 1e-4, dang.max()


def test_golden_vectors_of_the_reference(gpu_ctx):
    g = np.load(GOLDEN)
    grey = g["grey"]
    pts, desc = features.hahog(grey.astype(np.float32) / 255, 1e-5, 10.0, 1500)
    assert pts.shape == g["points"].shape
    assert np.array_equal(pts[:, :3], g["points"][:, :3])
    assert np.abs(pts[:, 3] - g["points"][:, 3]).max() <= 1e-4
    assert np.array_equal(desc[:64], g["desc_f32_head"]), int((desc[:64] != g["desc_f32_head"]).sum())
    # as features.extract_features_hahog returns them (square root, x 362, clip, round): integer levels
    p8, d8 = features.extract_features_hahog(grey, CFG, 1500)
    assert np.array_equal(p8[:, :3], g["points"][:, :3])
    assert d8.dtype == np.float32 and np.array_equal(d8, np.round(d8)) and d8.min() >= 0 and d8.max() <= 255
    diff = np.abs(d8 - g["desc_u8"].astype(np.float32))
    print("hahog golden: %d features, %d differing uint8 levels of %d" % (len(p8), np.count_nonzero(diff), d8.size))
    assert np.count_nonzero(diff) == 0, (diff.max(), np.count_nonzero(diff))


def _texture(rows, cols, seed):
    """blobs, edges and noise at several scales, grey levels in [0, 1]"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float32)
    im = np.zeros((rows, cols), np.float32)
    for _ in range(300):
        cx, cy, s = rng.uniform(0, cols), rng.uniform(0, rows), rng.uniform(1.5, 18)
        im += rng.uniform(-1, 1) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    im += 0.3 * np.sin(xx / 7.0) * np.cos(yy / 11.0) + 0.05 * rng.standard_normal((rows, cols)).astype(np.float32)
    im -= im.min()
    return np.ascontiguousarray(im / im.max(), np.float32)


@pytest.mark.parametrize("rows,cols,target,seed", [(333, 517, 400, 1), (480, 640, 100000, 2), (257, 129, 50, 3), (900, 1200, 3000, 4), (64, 64, 10, 5)])
def test_matches_the_compiled_reference(oracle_lib, gpu_ctx, rows, cols, target, seed):
    """odd sizes (the octaves halve by truncation), a target above the number of detections (no sort: vlfeat's detection order), features
    at the border (padded patches), a tiny image"""
    im = _texture(rows, cols, seed)
    ref = oracle_lib.hahog_ref(im, 1e-5, 10.0, target)
    if ref is None:
        pytest.skip("oracle/_ref/libhahog_ref.so is not available (it is built where /root/reference is mounted)")
    pts, desc = features.hahog(im, 1e-5, 10.0, target)
    _compare(pts, desc, *ref)


@pytest.mark.parametrize("rows,cols,seed", [(75, 64, 11), (76, 65, 12), (90, 250, 13), (91, 251, 14), (83, 311, 15), (26, 498, 16), (19, 17, 17)])
def test_extremum_search_tile_borders(oracle_lib, gpu_ctx, rows, cols, seed):
    """round 6: the extremum search tiles an octave by 62 columns per wavefront (lanes 0 and 63 carry the columns beside them), 248 per
    workgroup and 8 rows: interiors of exactly 62, 63, 248, 249 columns, a second workgroup of one column, heights of 8 k + 1 ... rows, the
    smallest image that has an octave -- every detection of the reference, nothing else, and the keypoints / descriptors that follow from them"""
    rng = np.random.default_rng(seed)
    im = _texture(rows, cols, seed)
    im = np.ascontiguousarray(np.clip(im + 0.2 * rng.standard_normal((rows, cols)).astype(np.float32), 0, 1), np.float32)  # extrema everywhere
    ref = oracle_lib.hahog_ref(im, 1e-5, 10.0, 100000)
    if ref is None:
        pytest.skip("oracle/_ref/libhahog_ref.so is not available (it is built where /root/reference is mounted)")
    pts, desc = features.hahog(im, 1e-5, 10.0, 100000)
    _compare(pts, desc, *ref)


def test_thresholds_and_empty_results(oracle_lib, gpu_ctx):
    im = _texture(200, 300, 9)
    for peak, edge, target in ((1e-3, 10.0, 500), (1e-5, 2.0, 500), (10.0, 10.0, 500), (1e-5, 10.0, 0)):
        pts, desc = features.hahog(im, peak, edge, target)
        ref = oracle_lib.hahog_ref(im, peak, edge, target)
        if ref is None:
            pytest.skip("oracle/_ref/libhahog_ref.so is not available")
        _compare(pts, desc, *ref)
    assert len(features.hahog(im, 10.0, 10.0, 500)[0]) == 0
    assert len(features.hahog(im, 1e-5, 10.0, 0)[0]) == 0  # hahog.cc:24-27 keeps `target` = 0 of the sorted list
    assert features.hahog(np.zeros((0, 0), np.float32), 1e-5, 10.0, 10) is None  # hahog.cc:127-129
    # below one 16-pixel octave vlfeat has no scale space: no features, no exception (thumbnails, masked crops)
    for shape in ((16, 300), (40, 9), (1, 1)):
        p0, d0 = features.hahog(np.full(shape, 0.5, np.float32), 1e-5, 10.0, 10)
        assert p0.shape == (0, 4) and d0.shape == (0, 128)


def test_descriptors_feed_the_matcher(gpu_ctx):
    """two crops of the example image: the integer-valued descriptors go straight into the resident store of the matcher"""
    from opensfm_amd import matching

    g = np.load(GOLDEN)
    grey = g["grey"]
    a, b = grey[:, :560], grey[:, 80:]
    pa, da = features.extract_features_hahog(a, CFG, 1200)
    pb, db = features.extract_features_hahog(b, CFG, 1200)
    cfg = {"lowes_ratio": 0.8}
    m = matching.match_brute_force_symmetric(da, db, cfg)
    m = np.asarray(m)
    assert len(m) > 200
    dx = pb[m[:, 1], 0] - pa[m[:, 0], 0]
    assert np.median(np.abs(dx + 80)) < 0.5  # the crops are 80 pixels apart


def test_batch_equals_image_by_image(gpu_ctx):
    """osfm_hahog_extract_batch: images of different sizes in flight on separate streams and host threads; every image's keypoints (all
    four columns) and descriptors are bit for bit what the single-image call returns, whatever the concurrency."""
    from opensfm_amd import features

    rng = np.random.default_rng(5)
    sizes = [(240, 320), (333, 257), (480, 640), (64, 64), (200, 500), (301, 299), (16, 40), (128, 130), (480, 640)]
    ims = []
    for r, c in sizes:
        im = rng.random((r, c)).astype(np.float32)
        k = np.ones(5, np.float32) / 5  # some structure: box blur of noise
        im = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, im))
        ims.append(np.ascontiguousarray(im, np.float32))
    single = [features._extract(im, 1e-5, 10.0, 300, features.HAHOG_ROOT | features.HAHOG_UCHAR, gpu_ctx) for im in ims]
    for conc in (1, 3, 8):
        got = features.hahog_batch(ims, 1e-5, 10.0, 300, features.HAHOG_ROOT | features.HAHOG_UCHAR, concurrency=conc, ctx=gpu_ctx)
        assert len(got) == len(single)
        for (p, d), (ps, ds) in zip(got, single):
            assert np.array_equal(p, ps) and np.array_equal(d, ds)
    assert len(single[6][0]) == 0  # below one octave: no features, no error
    assert sum(len(p) for p, _ in single) > 500


def test_uint8_images_equal_their_float_form(gpu_ctx):
    """OSFM_HAHOG_IMAGE_U8: grey levels handed over as bytes, level / 255 formed in float32 on the device -- the host-side
    image.astype(np.float32) / 255 of features.extract_features_hahog (opensfm/features.py:524) bit for bit: identical keypoints and
    descriptors, single image and batch"""
    from opensfm_amd import features

    rng = np.random.default_rng(11)
    ims = []
    for r, c in ((240, 320), (333, 257), (480, 640)):
        im = rng.random((r, c))
        k = np.ones(5) / 5
        im = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, im))
        ims.append(np.ascontiguousarray(np.round(255 * im / im.max()), np.uint8))
    fl = features.HAHOG_ROOT | features.HAHOG_UCHAR
    for im in ims:
        pf, df = features._extract(im.astype(np.float32) / 255, 1e-5, 10.0, 400, fl, gpu_ctx)
        pu, du = features._extract(im, 1e-5, 10.0, 400, fl | features.HAHOG_IMAGE_U8, gpu_ctx)
        assert len(pf) > 50 and np.array_equal(pf, pu) and np.array_equal(df, du)
        p2, d2 = features.extract_features_hahog(im, CFG, 400, ctx=gpu_ctx)  # the facade takes the byte path for uint8 input
        p3, d3 = features.extract_features_hahog(im.astype(np.float64), CFG, 400, ctx=gpu_ctx)  # ... and the float path otherwise
        assert np.array_equal(p2, p3) and np.array_equal(d2, d3)
    got = features.hahog_batch(ims, 1e-5, 10.0, 400, fl, concurrency=3, ctx=gpu_ctx)
    ref = features.hahog_batch([im.astype(np.float32) / 255 for im in ims], 1e-5, 10.0, 400, fl, concurrency=3, ctx=gpu_ctx)
    for (p, d), (pr, dr) in zip(got, ref):
        assert np.array_equal(p, pr) and np.array_equal(d, dr)


def test_fused_smoothing_equals_the_two_passes(gpu_ctx, monkeypatch):
    """The scale space's separable Gaussian as one LDS-tiled launch (smooth_fused_kernel) against the column kernel followed by the row
    kernel (OSFM_HAHOG_TWO_PASS): the same float operations in the same order, so every keypoint and descriptor is identical -- on sizes
    that are not multiples of the tile and smaller than one."""
    from opensfm_amd import features

    rng = np.random.default_rng(11)
    for r, c in [(480, 640), (333, 257), (65, 33), (40, 700)]:
        im = rng.random((r, c)).astype(np.float32)
        im = np.ascontiguousarray((im + np.roll(im, 1, 0) + np.roll(im, 1, 1) + np.roll(im, 2, 0) + np.roll(im, 2, 1)) / 5, np.float32)
        a = features.hahog(im, 1e-5, 10.0, 500, ctx=gpu_ctx)
        monkeypatch.setenv("OSFM_HAHOG_TWO_PASS", "1")
        b = features.hahog(im, 1e-5, 10.0, 500, ctx=gpu_ctx)
        monkeypatch.delenv("OSFM_HAHOG_TWO_PASS")
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (r, c)
    assert len(a[0]) > 0


def test_batch_reports_the_image_that_failed(gpu_ctx):
    """an image whose features do not fit the rows the caller offers stops the batch with OSFM_E_INVALID and names the image; the other
    images' results are not handed out (the call failed), and the context keeps working afterwards"""
    import ctypes as C

    from opensfm_amd import _lib, features
    from opensfm_amd._lib import OsfmError

    rng = np.random.default_rng(2)
    ims = [np.ascontiguousarray(rng.random((120, 160)), np.float32) for _ in range(3)]
    lib = _lib.load()
    n = len(ims)
    caps = [64, 2, 64]  # image 1 has more than two features
    pts = [np.empty((c, 4), np.float32) for c in caps]
    desc = [np.empty((c, 128), np.float32) for c in caps]
    ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in ims])
    rows, cols = (C.c_int * n)(*[120] * n), (C.c_int * n)(*[160] * n)
    pp, dp = (C.c_void_p * n)(*[a.ctypes.data for a in pts]), (C.c_void_p * n)(*[a.ctypes.data for a in desc])
    nf = (C.c_int * n)()
    rc = lib.osfm_hahog_extract_batch(gpu_ctx.handle, n, ptrs, rows, cols, 1e-5, 10.0, 16, 0, pp, dp, (C.c_int * n)(*caps), nf, 2)
    assert rc != 0 and b"image 1" in lib.osfm_last_error()
    with pytest.raises(OsfmError):
        _lib.check(rc, "osfm_hahog_extract_batch")
    ok = features.hahog_batch(ims, 1e-5, 10.0, 16, ctx=gpu_ctx)
    assert len(ok) == 3 and all(len(p) > 2 for p, _ in ok)
