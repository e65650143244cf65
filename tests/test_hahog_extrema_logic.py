"""The extremum search of hahog.hip (round 6) restated in numpy: the 26 strict comparisons of vl_find_local_extrema_3 (covdet.c:1044-1117) as ONE
comparison against the largest / smallest neighbour built from row maxima (max3 along x, then along y, then across the levels), and the tiling
of an octave by wavefronts of 62 columns (lanes 0 and 63 carry the neighbouring columns), workgroups of 4 x 62 columns x 8 rows.  Ties matter: a
sample equal to a neighbour is NOT an extremum, so the volumes here are quantised to a few levels."""
import numpy as np

K_ROWS, K_WAVE_COLS, K_WAVES = 8, 62, 4  # kExRows, kExWaveCols, waves per workgroup (hahog.hip)


def brute_force(css, thr):
    lev, h, w = css.shape
    out = set()
    for z in range(1, lev - 1):
        for y in range(1, h - 1):
            for x in range(1, w - 1):
                v = css[z, y, x]
                nb = css[z - 1:z + 2, y - 1:y + 2, x - 1:x + 2].copy().ravel()
                nb = np.delete(nb, 13)
                if (v >= thr and (v > nb).all()) or (v <= -thr and (v < nb).all()):
                    out.add((z, y, x))
    return out


def row_maxima(css, thr):
    """what phase A of extrema_kernel computes, array-wide"""
    lev, h, w = css.shape
    pad = np.pad(css, ((0, 0), (1, 1), (1, 1)), mode="edge")  # clamped loads; border samples are never candidates
    left, right, c = pad[:, :, :-2], pad[:, :, 2:], pad[:, :, 1:-1]
    hm, hn = np.maximum(np.maximum(left, c), right), np.minimum(np.minimum(left, c), right)  # along x, rows -1 .. h
    sm, sn = np.maximum(left, right)[:, 1:-1], np.minimum(left, right)[:, 1:-1]                 # the sample's own row without itself
    vm = np.maximum(np.maximum(hm[:, :-2], hm[:, 1:-1]), hm[:, 2:])                              # 3 x 3 window of every level
    vn = np.minimum(np.minimum(hn[:, :-2], hn[:, 1:-1]), hn[:, 2:])
    out = set()
    for z in range(1, lev - 1):
        nbmax = np.maximum(np.maximum(vm[z - 1], vm[z + 1]), np.maximum(np.maximum(hm[z, :-2], hm[z, 2:]), sm[z]))
        nbmin = np.minimum(np.minimum(vn[z - 1], vn[z + 1]), np.minimum(np.minimum(hn[z, :-2], hn[z, 2:]), sn[z]))
        v = css[z]
        ext = ((v >= thr) & (v > nbmax)) | ((v <= -thr) & (v < nbmin))
        ext[0, :] = ext[-1, :] = False
        ext[:, 0] = ext[:, -1] = False
        out |= {(z, int(y), int(x)) for y, x in zip(*np.nonzero(ext))}
    return out


def test_one_comparison_against_the_row_maxima_is_the_26_strict_comparisons():
    rng = np.random.default_rng(3)
    for shape, levels in (((5, 19, 23), 4), ((5, 12, 70), 7), ((5, 9, 9), 50), ((5, 3, 3), 3)):
        css = (rng.integers(-levels, levels + 1, size=shape) / levels).astype(np.float32)  # many equal neighbours
        for thr in (0.0, 0.5):
            assert row_maxima(css, thr) == brute_force(css, thr), (shape, levels, thr)


def test_every_interior_sample_belongs_to_one_lane_of_one_workgroup():
    """xs = tile_x * 248 + wave * 62 + lane, lanes 1 .. 62 decide; rows yb .. yb + 7 of tile_y; grid = ceil((w - 2) / 248) x ceil((h - 2) / 8)"""
    cols = K_WAVE_COLS * K_WAVES
    for w, h in ((3, 3), (64, 11), (65, 10), (250, 9), (251, 17), (499, 26), (2048, 19)):
        gx, gy = (w - 2 + cols - 1) // cols, (h - 2 + K_ROWS - 1) // K_ROWS
        seen = np.zeros((h, w), np.int32)
        for ty in range(gy):
            for tx in range(gx):
                for wave in range(K_WAVES):
                    for lane in range(64):
                        xs = tx * cols + wave * K_WAVE_COLS + lane
                        if not (1 <= lane <= K_WAVE_COLS and xs <= w - 2):
                            continue
                        assert xs - 1 >= 0 and (lane > 0) and (lane < 63)  # the columns beside it are lanes lane - 1 and lane + 1 of the same wavefront
                        for j in range(1, K_ROWS + 1):
                            y = ty * K_ROWS + j
                            if y <= h - 2:
                                seen[y, xs] += 1
        assert (seen[1:-1, 1:-1] == 1).all() and seen[0].sum() == 0 and seen[-1].sum() == 0 and seen[:, 0].sum() == 0 and seen[:, -1].sum() == 0, (w, h)
