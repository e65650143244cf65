"""The two reorganisations of HAHOG's sequential histogram loops (opensfm_amd/csrc/hahog.hip) restated in numpy and held
against the sequential loops they replace, bit for bit: (1) orientation_kernel's stable counting sort of the 2 x 1 681 (bin, product)
records with ranks from per-chunk / per-wave ballots (covdet.c:2769-2781 is the loop), (2) descriptor_kernel's walk over only the rows
and columns whose measured bin ranges admit a bin (sift.c:1806-1850).  The kernels themselves are checked against the compiled
reference on the GPU (tests/test_gpu_hahog.py); this pins the index arithmetic and the '+0.0 leaves the sum unchanged' argument."""
import numpy as np

N_OR, BINS = 41 * 41, 36


def _orientation_sequential(hbin, cx, cy):
    hist = np.zeros(BINS)
    for k in range(N_OR):
        hist[hbin[k]] += cx[k]
        hist[(hbin[k] + 1) % BINS] += cy[k]
    return hist


def _orientation_sorted(hbin, cx, cy):
    tot = np.bincount(hbin, minlength=BINS)
    start = np.zeros(BINS + 1, np.int64)
    for v in range(BINS):
        start[v + 1] = start[v] + tot[v] + tot[(v - 1) % BINS]
    order = np.full(2 * N_OR, -1, np.int64)
    run = np.zeros(BINS, np.int64)
    for c0 in range(0, N_OR, 256):  # a chunk of 256 pixels = 4 waves of 64 lanes
        cw = np.zeros((4, BINS), np.int64)
        pre = {}
        for w in range(4):
            lanes = [c0 + 64 * w + l for l in range(64)]
            b = [hbin[t] if t < N_OR else -1 for t in lanes]
            for v in range(BINS):
                mask = [x == v for x in b]
                cw[w, v] = sum(mask)
                for l, t in enumerate(lanes):
                    if t < N_OR:
                        pre[(t, v)] = sum(mask[:l])  # popcount of the ballot below the lane
        for w in range(4):
            for l in range(64):
                t = c0 + 64 * w + l
                if t >= N_OR:
                    continue
                b0 = hbin[t]
                bm1, bp1 = (b0 - 1) % BINS, (b0 + 1) % BINS
                e = {v: run[v] + cw[:w, v].sum() + pre[(t, v)] for v in (bm1, b0, bp1)}
                order[start[b0] + e[b0] + e[bm1]] = 2 * t
                order[start[bp1] + e[bp1] + e[b0]] = 2 * t + 1
        run += cw.sum(0)
    assert (order >= 0).all() and len(np.unique(order)) == 2 * N_OR
    rec = np.stack([cx, cy], 1).ravel()
    hist = np.zeros(BINS)
    for b in range(BINS):
        h = 0.0
        for k in range(start[b], start[b + 1]):
            h += rec[order[k]]
        hist[b] = h
    return hist


def test_sorted_records_add_up_like_the_sequential_orientation_loop():
    rng = np.random.default_rng(5)
    for trial in range(3):
        hbin = rng.integers(0, BINS, N_OR) if trial else np.full(N_OR, 7)  # trial 0: every pixel in one bin
        m = rng.random(N_OR) * 10.0 ** rng.integers(-6, 3, N_OR)
        w2 = rng.random(N_OR)
        cx, cy = (1.0 - w2) * m, w2 * m
        a, b = _orientation_sequential(hbin, cx, cy), _orientation_sorted(hbin, cx, cy)
        assert np.array_equal(a.view(np.int64), b.view(np.int64))


SIDE, NBP, NBO = 31, 4, 8


def _descriptor_pixels(rng, angle):
    ys, xs = np.mgrid[0:SIDE, 0:SIDE]
    dx, dy = (xs - 15.0).astype(np.float32), (ys - 15.0).astype(np.float32)
    st, ct = np.sin(angle), np.cos(angle)
    sbp = 6.0 + 2.220446049250313e-16
    nx = ((ct * dx + st * dy) / sbp).astype(np.float32)
    ny = ((-st * dx + ct * dy) / sbp).astype(np.float32)
    nt = (rng.random((SIDE, SIDE)) * 8).astype(np.float32)
    binx, biny, bint = np.floor(nx - np.float32(0.5)).astype(int), np.floor(ny - np.float32(0.5)).astype(int), np.floor(nt).astype(int)
    rx = (nx - (binx + 0.5)).astype(np.float32)
    ry = (ny - (biny + 0.5)).astype(np.float32)
    rt = nt - bint.astype(np.float32)
    wm = (rng.random((SIDE, SIDE)) * 10.0 ** rng.integers(-5, 2, (SIDE, SIDE))).astype(np.float32)
    return binx, biny, bint, rx, ry, rt, wm


def _bin_sum(bx, by, bt, px, pixels):
    binx, biny, bint, rx, ry, rt, wm = px
    acc = np.float32(0)
    for (y, x) in pixels:
        dbx, dby = bx - binx[y, x], by - biny[y, x]
        inside = 0 <= dbx <= 1 and 0 <= dby <= 1
        base = wm[y, x] * np.abs(np.float32(1 - dbx) - rx[y, x]) * np.abs(np.float32(1 - dby) - ry[y, x])
        v0, v1 = base * np.abs(np.float32(1) - rt[y, x]), base * np.abs(np.float32(0) - rt[y, x])
        b0, b1 = bint[y, x] % NBO, (bint[y, x] + 1) % NBO
        acc = acc + (v0 if (inside and b0 == bt) else (v1 if (inside and b1 == bt) else np.float32(0)))
    return acc


def test_rows_and_columns_that_can_feed_a_bin_give_the_full_walk():
    rng = np.random.default_rng(8)
    for angle in (np.pi / 2, 0.3):  # the product's rotation and an arbitrary one: the masks are measured, not derived
        px = _descriptor_pixels(rng, angle)
        binx, biny = px[0], px[1]
        everything = [(y, x) for y in range(SIDE) for x in range(SIDE)]
        for bx in range(-NBP // 2, NBP // 2):
            for by in range(-NBP // 2, NBP // 2):
                rows = [y for y in range(SIDE) if binx[y].min() <= bx and binx[y].max() >= bx - 1 and biny[y].min() <= by and biny[y].max() >= by - 1]
                cols = [x for x in range(SIDE) if binx[:, x].min() <= bx and binx[:, x].max() >= bx - 1 and biny[:, x].min() <= by and biny[:, x].max() >= by - 1]
                some = [(y, x) for y in rows for x in cols]
                for bt in (0, 5):
                    full, part = _bin_sum(bx, by, bt, px, everything), _bin_sum(bx, by, bt, px, some)
                    assert np.float32(full).view(np.int32) == np.float32(part).view(np.int32)
        if angle == np.pi / 2:
            assert len(some) < len(everything) // 3  # and it is a real saving for the product's axis-aligned bins


def test_the_list_of_a_spatial_bin_gives_the_full_walk():
    """round 6: a lane walks the LIST of the pixels that feed its spatial bin (raster order; which pixels and through which of their four spatial
    products (window x modulus) |1 - dx - rx| |1 - dy - ry| does not depend on the feature), the products formed once per pixel: the same float32
    sum as the walk over all 961 pixels, for every bin"""
    rng = np.random.default_rng(9)
    for angle in (np.pi / 2, 0.3):
        px = _descriptor_pixels(rng, angle)
        binx, biny, bint, rx, ry, rt, wm = px
        everything = [(y, x) for y in range(SIDE) for x in range(SIDE)]
        # once per pixel: base[2 dx + dy], the first three factors in the reference's order
        base = np.empty((SIDE, SIDE, 4), np.float32)
        for dx in (0, 1):
            for dy in (0, 1):
                base[:, :, 2 * dx + dy] = wm * np.abs(np.float32(1 - dx) - rx) * np.abs(np.float32(1 - dy) - ry)
        nt = bint.astype(np.float32) + rt
        for bx in range(-NBP // 2, NBP // 2):
            for by in range(-NBP // 2, NBP // 2):
                lst = [(y, x, 2 * (bx - binx[y, x]) + (by - biny[y, x])) for (y, x) in everything if 0 <= bx - binx[y, x] <= 1 and 0 <= by - biny[y, x] <= 1]
                for bt in (0, 3, 7):
                    acc = np.float32(0)
                    for (y, x, idx) in lst:
                        fl = np.floor(nt[y, x])
                        r = nt[y, x] - fl
                        sb = int(fl)
                        v0, v1 = base[y, x, idx] * np.abs(np.float32(1) - r), base[y, x, idx] * np.abs(np.float32(0) - r)
                        acc = acc + (v0 if (sb & 7) == bt else (v1 if ((sb + 1) & 7) == bt else np.float32(0)))
                    full = _bin_sum(bx, by, bt, px, everything)
                    assert np.float32(full).view(np.int32) == np.float32(acc).view(np.int32), (angle, bx, by, bt)
        if angle == np.pi / 2:
            assert len(lst) < len(everything) // 3
