"""Soundness of the float-descriptor bounds of the fused matcher's FQ mode, checked on the CPU against the oracle's
float32 arithmetic: with x^ = round((v - lo) * 255 / (hi - lo)) - 128 and E = max row residual norm,
    | scale * d_float32(q, t) - ||x^_q - x^_t|| |  <=  E_Q + E_T + kappa * D
must hold for every pair of rows, and the two decisions the kernel takes from quantised distances alone -- "surely fails" and "surely
passes with this winner" -- must never contradict the oracle's outcome.  (The kernel itself is tested on the GPU,
tests/test_gpu_float_descriptors.py; this file tests the inequalities it relies on, with the quantisation restated from api.hip.)"""
import numpy as np
import pytest

KAPPA = 8e-6


def quantise(images):
    """store_upload's quantisation: one value range for the store, per image the largest residual norm (rounded up)"""
    allv = np.concatenate(images).astype(np.float32).astype(np.float64)
    lo, hi = allv.min(), allv.max()
    scale = 255.0 / (hi - lo)
    out = []
    for d in images:
        x = (d.astype(np.float32).astype(np.float64) - lo) * scale - 128.0
        xq = np.clip(np.rint(x), -128, 127)
        e = np.sqrt(((x - xq) ** 2).sum(1)) * (1 + 1e-9) + 1e-9
        out.append((xq, float(np.nextafter(np.float32(e.max()), np.float32(np.inf)))))
    return out, scale


def decisions(xq_q, xq_t, eps, ratio):
    """per query: (surely_fails, surely_passes, winner) from exact quantised distances"""
    d2 = (xq_q ** 2).sum(1)[:, None] + (xq_t ** 2).sum(1)[None, :] - 2.0 * xq_q @ xq_t.T
    order = np.argsort(d2, axis=1, kind="stable")[:, :2]
    D0 = np.sqrt(np.take_along_axis(d2, order[:, :1], 1)[:, 0])
    D1 = np.sqrt(np.take_along_axis(d2, order[:, 1:2], 1)[:, 0])
    D0lo, D0hi = np.maximum(D0 - eps, 0) * (1 - KAPPA), (D0 + eps) * (1 + KAPPA)
    D1lo, D1hi = np.maximum(D1 - eps, 0) * (1 - KAPPA), (D1 + eps) * (1 + KAPPA)
    return D0lo >= ratio * D1hi, (D0hi < ratio * D1lo) & (D0hi < D1lo), order[:, 0]


def transforms(rng):
    return {
        "root": lambda d: np.sqrt(d / np.maximum(d.sum(1, keepdims=True), 1e-7)),
        "signed": lambda d: (d - 40.0) * 0.37,
        "wide": lambda d: d * d * 1e-3 + rng.uniform(0, 1e-3, d.shape),
        "tiny": lambda d: d * 1e-5,
    }


@pytest.mark.parametrize("kind", ["root", "signed", "wide", "tiny"])
def test_bounds_hold_and_decisions_agree_with_the_oracle(oracle_lib, kind):
    rng = np.random.default_rng(11)
    tf = transforms(rng)[kind]
    n = 260
    base = rng.integers(0, 90, (n, 128)).astype(np.float64)
    imgs = []
    for _ in range(3):
        amp = rng.uniform(0.0, 38.0, (n, 1))
        d = np.clip(base + rng.normal(0, 1, (n, 128)) * amp, 0, 255)
        d[5] = d[9]  # an exact duplicate inside the image
        d[17] = np.clip(d[18] + rng.normal(0, 0.02, 128), 0, 255)  # a near-duplicate below the quantisation step
        imgs.append(tf(d[rng.permutation(n)]).astype(np.float32))
    q, scale = quantise(imgs)
    seen = {"fail": 0, "pass": 0, "open": 0}
    for a in range(3):
        for b in range(3):
            if a == b:
                continue
            (xa, ea), (xb, eb) = q[a], q[b]
            eps = ea + eb
            # (1) the distance bound (real-valued distances; the float32 rounding of the reference's sum is what kappa covers)
            dq = np.sqrt((xa ** 2).sum(1)[:, None] + (xb ** 2).sum(1)[None, :] - 2.0 * xa @ xb.T)
            fa, fb = imgs[a].astype(np.float64), imgs[b].astype(np.float64)
            dreal = np.sqrt(np.maximum(((fa[:, None, :] - fb[None, :, :]) ** 2).sum(2), 0)) * scale  # real-valued distances, scaled
            assert np.all(np.abs(dreal - dq) <= eps + 1e-9), kind
            # (2) the decisions never contradict the oracle's matcher (float32 arithmetic in cv2's order, ratio test in doubles)
            fails, passes, winner = decisions(xa, xb, eps, 0.8)
            got = {int(i): int(j) for i, j in oracle_lib.match_brute_force(imgs[a], imgs[b], 0.8)}
            for i in range(n):
                if fails[i]:
                    assert i not in got, (kind, a, b, i)
                    seen["fail"] += 1
                elif passes[i]:
                    assert got.get(i) == int(winner[i]), (kind, a, b, i)
                    seen["pass"] += 1
                else:
                    seen["open"] += 1
    assert seen["fail"] > 100 and seen["pass"] > 100 and seen["open"] > 0, seen  # all three outcomes occur
