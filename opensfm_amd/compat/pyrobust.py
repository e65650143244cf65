"""``pyrobust`` (opensfm/src/robust/python/pybind.cc:26-56): ``ransac_relative_pose`` -- the estimator of the calibrated robust-matching
branch (robust/src/instanciations.cc:33-48) -- with ``RobustEstimatorParams`` and ``RansacType``.  The other estimators of the
reference's module (line, essential, relative rotation, absolute pose, similarity) are off the matching path and not provided."""
import enum

import numpy as np

from .. import matching as _matching


class RansacType(enum.IntEnum):
    RANSAC = 0
    MSAC = 1
    LMedS = 2


RANSAC, MSAC, LMedS = RansacType.RANSAC, RansacType.MSAC, RansacType.LMedS  # export_values()


class RobustEstimatorParams:
    """robust/robust_estimator.h: iterations 100, probability 0.99, local optimisation and iteration reduction on"""

    def __init__(self):
        self.iterations = 100
        self.probability = 0.99
        self.use_local_optimization = True
        self.use_iteration_reduction = True


class ScoreInfoMatrix34d:
    def __init__(self):
        self.score = 0.0
        self.model = np.zeros((3, 4))
        self.lo_model = np.zeros((3, 4))
        self.inliers_indices = []


def ransac_relative_pose(b1, b2, threshold: float, parameters: RobustEstimatorParams, ransac_type: RansacType = RansacType.RANSAC):
    """robust::RANSACRelativePose: LO-RANSAC of the relative pose on unit bearings (n x 3 each); ScoreInfo with the 3 x 4 model [R | t]"""
    if int(ransac_type) != int(RansacType.RANSAC):
        raise NotImplementedError("only RansacType.RANSAC is on the GPU path (what multiview.relative_pose_ransac asks for)")
    if not parameters.use_iteration_reduction:
        raise NotImplementedError("use_iteration_reduction = False is not on the GPU path")
    b1, b2 = np.asarray(b1, np.float64).reshape(-1, 3), np.asarray(b2, np.float64).reshape(-1, 3)
    if len(b1) != len(b2):
        raise RuntimeError("Features matrices have different sizes.")  # instanciations.cc:20-22
    res, mask, _ = _matching.relpose_pairs(b1, b2, [0, len(b1)], threshold, mode="ransac", iterations=parameters.iterations,
                                           probability=parameters.probability, use_lo=parameters.use_local_optimization)
    out = ScoreInfoMatrix34d()
    out.score, out.model, out.lo_model = res[0]["score"], res[0]["model"], res[0]["lo_model"]
    out.inliers_indices = [int(i) for i in np.flatnonzero(mask)]
    return out
