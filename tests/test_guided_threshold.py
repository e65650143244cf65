"""Guided matching tests (a + b) / 2 < c* instead of pi/2 - acos((a + b) / 2) < threshold (guided.hip), with c* found by
bisection over the doubles against libm's acos.  The two tests are the same function of c wherever acos is monotone: this file restates
the bisection (osfm_guided_cos_threshold) and checks the equivalence on a dense set of doubles around c*, for the thresholds in use."""
import math

import numpy as np
import pytest


def cos_threshold(threshold: float) -> float:
    below = lambda c: (math.pi / 2.0 - math.acos(c) < threshold) if c <= 1.0 else False  # noqa: E731  (NaN compares false)
    if not below(0.0):
        return 0.0
    if below(1.0):
        return float(np.nextafter(1.0, 2.0))
    lo, hi = 0.0, 1.0
    while True:
        mid = lo + (hi - lo) / 2.0
        if not (lo < mid < hi):
            break
        if below(mid):
            lo = mid
        else:
            hi = mid
    return hi


@pytest.mark.parametrize("thr", [1e-4, 0.002, 0.004, 0.006, 0.01, 0.05, 0.3, 1.0, 1.5])
def test_cosine_threshold_is_the_angle_threshold(thr):
    cstar = cos_threshold(thr)
    assert 0.0 < cstar <= 1.0 + 1e-15
    rng = np.random.default_rng(int(thr * 1e6))
    # every double within 4000 ulps of c*, and a random sample of the whole range
    near = [cstar]
    x = cstar
    for _ in range(4000):
        x = float(np.nextafter(x, 0.0))
        near.append(x)
    x = cstar
    for _ in range(4000):
        x = float(np.nextafter(x, 2.0))
        if x <= 1.0:
            near.append(x)
    for c in near + list(rng.uniform(0.0, 1.0, 20000)):
        assert (math.pi / 2.0 - math.acos(c) < thr) == (c < cstar), (thr, c, cstar)


def test_degenerate_thresholds():
    assert cos_threshold(0.0) == 0.0 and cos_threshold(-1.0) == 0.0  # nothing is allowed
    assert cos_threshold(2.0) > 1.0  # every defined angle is allowed (c <= 1); c > 1 gives NaN in the reference: not allowed
