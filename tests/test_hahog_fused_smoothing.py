"""smooth_fused_kernel (hahog.hip, round 4) restated in numpy, tile by tile, against the plain definitions it replaces: the column pass
followed by the row pass with padding by continuity (conv_v_kernel / conv_h_kernel = vl_imconvcol_vf twice) and the Hessian response with its
"nearest interior pixel" border (hessian_kernel = _vl_det_hessian_response).  The restatement follows the kernel's indexing -- tile origin
two output pixels outside the 60 x 28 interior, source tile with a halo of W read through clamped indices, the column pass kept for the
clamped columns, the response read from the output tile at clamped centres -- so that an indexing mistake in the tiling shows up without
a GPU; float32 throughout, sums in tap order, product and sum rounded separately.  The kernel itself is compared bit for bit with the two
kernels it replaces in tests/test_gpu_hahog.py."""
import numpy as np
import pytest

SX, SY, HALO = 64, 32, 2


def taps_for(W, rng):
    t = rng.random(2 * W + 1).astype(np.float32)
    return (t / t.sum()).astype(np.float32)


def two_pass(img, taps, W):
    h, w = img.shape
    tmp = np.zeros_like(img)
    for j in range(2 * W + 1):  # acc = acc + src[clamp(y - W + j)] * taps[2W - j], j ascending
        rows = np.clip(np.arange(h) - W + j, 0, h - 1)
        tmp = (tmp + (img[rows, :] * taps[2 * W - j]).astype(np.float32)).astype(np.float32)
    out = np.zeros_like(img)
    for j in range(2 * W + 1):
        cols = np.clip(np.arange(w) - W + j, 0, w - 1)
        out = (out + (tmp[:, cols] * taps[2 * W - j]).astype(np.float32)).astype(np.float32)
    return out


def hessian(level, factor):
    h, w = level.shape
    c = np.clip(np.arange(w), 1, w - 2)
    r = np.clip(np.arange(h), 1, h - 2)
    R, C = np.meshgrid(r, c, indexing="ij")
    p = lambda dy, dx: level[R + dy, C + dx]
    f = np.float32
    Lxx = (-p(0, -1) + f(2) * p(0, 0) - p(0, 1)).astype(f)
    Lyy = (-p(-1, 0) + f(2) * p(0, 0) - p(1, 0)).astype(f)
    Lxy = ((p(-1, -1) - p(1, -1) - p(-1, 1) + p(1, 1)).astype(f) / f(4)).astype(f)
    return ((Lxx * Lyy - Lxy * Lxy).astype(f) * f(factor)).astype(f)


def fused_tiles(img, taps, W, factor):
    h, w = img.shape
    out, css = np.zeros_like(img), np.zeros_like(img)
    sx, sy = SX - 2 * HALO, SY - 2 * HALO
    for by in range((h + sy - 1) // sy):
        for bx in range((w + sx - 1) // sx):
            x0, y0 = bx * sx - HALO, by * sy - HALO
            gy = np.clip(y0 - W + np.arange(SY + 2 * W), 0, h - 1)
            gx = np.clip(x0 - W + np.arange(SX + 2 * W), 0, w - 1)
            S = img[np.ix_(gy, gx)]                                   # source tile, clamped
            T = np.zeros((SY, SX + 2 * W), np.float32)                # column pass for every column of the tile
            for j in range(2 * W + 1):
                T = (T + (S[j:j + SY, :] * taps[2 * W - j]).astype(np.float32)).astype(np.float32)
            O = np.zeros((SY, SX), np.float32)                        # row pass: the output tile with its halo
            for j in range(2 * W + 1):
                O = (O + (T[:, j:j + SX] * taps[2 * W - j]).astype(np.float32)).astype(np.float32)
            for r in range(HALO, SY - HALO):
                for c in range(HALO, SX - HALO):
                    x, y = x0 + c, y0 + r
                    if x >= w or y >= h:
                        continue
                    out[y, x] = O[r, c]
                    cc, rr = min(max(x, 1), w - 2) - x0, min(max(y, 1), h - 2) - y0
                    assert 1 <= cc <= SX - 2 and 1 <= rr <= SY - 2
                    q = lambda dy, dx: O[rr + dy, cc + dx]
                    f = np.float32
                    Lxx = f(f(-q(0, -1) + f(2) * q(0, 0)) - q(0, 1))
                    Lyy = f(f(-q(-1, 0) + f(2) * q(0, 0)) - q(1, 0))
                    Lxy = f(f(f(f(q(-1, -1) - q(1, -1)) - q(-1, 1)) + q(1, 1)) / f(4))
                    css[y, x] = f(f(f(Lxx * Lyy) - f(Lxy * Lxy)) * f(factor))
    return out, css


@pytest.mark.parametrize("h,w,W", [(45, 70, 4), (33, 61, 8), (28, 60, 1), (64, 129, 5), (16, 12, 6), (3, 200, 2)])
def test_tiled_smoothing_and_response_equal_the_plain_passes(h, w, W):
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.random((h, w)).astype(np.float32)
    taps = taps_for(W, rng)
    want = two_pass(img, taps, W)
    got, css = fused_tiles(img, taps, W, 1.7)
    assert np.array_equal(got, want)
    assert np.array_equal(css, hessian(want, 1.7))
