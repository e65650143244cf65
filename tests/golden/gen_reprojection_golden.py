"""Generates tests/golden/reprojection_golden.json: residual + Jacobian of the reference's
reprojection error at the reference's OWN test inputs, evaluated independently of any C code.

Inputs: opensfm/src/bundle/test/reprojection_errors_test.cc:17-18,105-111,123-128 (point (1,2,3),
rt (0.1..0.6), observed (0.5,0.5), std_deviation 0.1, perspective camera array {0.3, 0.1, -0.03}
in native parameter order [k1, k2, focal]) and opensfm/src/geometry/test/camera_functions_test.cc
(point (0.1,0.2,0.3) style inputs, focal 0.4, k1 -0.1, k2 0.01); the FISHEYE cases use the fisheye test's
inputs (reprojection_errors_test.cc:131-137) and FisheyeProjection::Forward (camera_projections_functions.h:11-22).  The reference's expected values
are "autodiff of the same formulas"; here the formulas of transformations_functions.h:112-144,
camera_projections_functions.h:88-93, camera_distortions_functions.h:106-113,
transformations_functions.h:55-58 and projection_errors.h:203-205 are written in mpmath (50
digits) and differentiated numerically in 50-digit arithmetic (central differences, h = 1e-20),
i.e. accurate to far better than the 1e-14 the reference asserts.

Run:  python tests/golden/gen_reprojection_golden.py
"""
import json
import os

import mpmath as mp

mp.mp.dps = 50


def residual(X, pose, cam, obs, sd, model="perspective"):
    r, t = pose[:3], pose[3:]
    p = [X[i] - t[i] for i in range(3)]
    a = [-r[i] for i in range(3)]
    th2 = sum(v * v for v in a)
    cp = [a[1] * p[2] - a[2] * p[1], a[2] * p[0] - a[0] * p[2], a[0] * p[1] - a[1] * p[0]]
    if th2 > mp.mpf(2) ** -52:
        th = mp.sqrt(th2)
        c, s = mp.cos(th), mp.sin(th) / th
        dot = sum(a[i] * p[i] for i in range(3)) * (1 - c) / th2
        Xc = [p[i] * c + s * cp[i] + a[i] * dot for i in range(3)]
    else:
        Xc = [p[i] + cp[i] for i in range(3)]
    def persp(P):
        return P[0] / P[2], P[1] / P[2]

    def fish(P):  # camera_projections_functions.h:11-22 (FisheyeProjection::Forward)
        r = mp.sqrt(P[0] ** 2 + P[1] ** 2)
        theta = mp.atan2(r, P[2])
        return theta / r * P[0], theta / r * P[1]

    if model in GENERIC:  # ProjectGeneric<PROJ, DISTO, AFF>: parameters [PROJ][DISTO][AFF] (camera_instances.h:127-160)
        proj, disto, nd, na = GENERIC[model]
        k = list(cam)
        if proj == "dual":  # camera_projections_functions.h:122-134
            t = k.pop(0)
            a, b = persp(Xc), fish(Xc)
            u, v = t * a[0] + (1 - t) * b[0], t * a[1] + (1 - t) * b[1]
        else:
            u, v = (fish if proj == "fisheye" else persp)(Xc)
        kd, ka = k[:nd], k[nd:nd + na]
        r2 = u * u + v * v
        tx = ty = mp.mpf(0)
        if disto == "disto2":
            rad = 1 + r2 * kd[0]
        elif disto == "disto24":
            rad = 1 + r2 * (kd[0] + kd[1] * r2)
        elif disto == "disto2468":
            rad = 1 + r2 * (kd[0] + r2 * (kd[1] + r2 * (kd[2] + r2 * kd[3])))
        elif disto == "brown":
            rad = 1 + r2 * (kd[0] + r2 * (kd[1] + r2 * kd[2]))
            p1, p2 = kd[3], kd[4]
            tx, ty = 2 * p1 * u * v + p2 * (r2 + 2 * u * u), 2 * p2 * u * v + p1 * (r2 + 2 * v * v)
        else:  # disto62 / disto624 (camera_distortions_functions.h:481-492,700-703)
            rad = 1 + r2 * (kd[0] + r2 * (kd[1] + r2 * (kd[2] + r2 * (kd[3] + r2 * (kd[4] + r2 * kd[5])))))
            p1, p2 = kd[6], kd[7]
            tx, ty = 2 * p1 * u * v + p2 * (r2 + 2 * u * u), 2 * p2 * u * v + p1 * (r2 + 2 * v * v)
            if disto == "disto624":
                tx += kd[8] * r2 + kd[9] * r2 * r2
                ty += kd[10] * r2 + kd[11] * r2 * r2
        du, dv = u * rad + tx, v * rad + ty
        if na == 4:  # Affine: focal, aspect ratio, cx, cy (transformations_functions.h:11-20)
            pu, pv = ka[0] * du + ka[2], ka[0] * ka[1] * dv + ka[3]
        else:
            pu, pv = ka[0] * du, ka[0] * dv
        return [(pu - obs[0]) / sd, (pv - obs[1]) / sd]
    if model == "fisheye":
        u, v = fish(Xc)
    else:
        u, v = persp(Xc)
    r2 = u * u + v * v
    k1, k2, f = cam
    d = 1 + r2 * (k1 + k2 * r2)
    return [(f * d * u - obs[0]) / sd, (f * d * v - obs[1]) / sd]


# model -> (projection, distortion, #distortion parameters, #affine parameters), camera_instances.h:183-192
GENERIC = {
    "brown": ("perspective", "brown", 5, 4), "fisheye_opencv": ("fisheye", "disto2468", 4, 4),
    "fisheye62": ("fisheye", "disto62", 8, 4), "fisheye624": ("fisheye", "disto624", 12, 4),
    "dual": ("dual", "disto24", 2, 1), "radial": ("perspective", "disto24", 2, 4),
    "simple_radial": ("perspective", "disto2", 1, 4),
}


def jac(fun, x):
    h = mp.mpf(10) ** -20
    cols = []
    for i in range(len(x)):
        xp, xm = list(x), list(x)
        xp[i] += h
        xm[i] -= h
        fp, fm = fun(xp), fun(xm)
        cols.append([(fp[k] - fm[k]) / (2 * h) for k in range(2)])
    return [[cols[i][k] for i in range(len(x))] for k in range(2)]


CASES = [
    # reprojection_errors_test.cc PerspectiveAnalyticErrorEvaluatesOK inputs (single pose)
    dict(X=[1.0, 2.0, 3.0], pose=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6], cam=[0.3, 0.1, -0.03], obs=[0.5, 0.5], sd=0.1),
    # camera_functions_test.cc style: focal 0.4, k1 -0.1, k2 0.01
    dict(X=[0.1, 0.2, 0.3], pose=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6], cam=[-0.1, 0.01, 0.4], obs=[0.0, 0.0], sd=1.0),
    # synthetic scene camera (synthetic_examples.py:56,81), generic pose, sigma 0.004
    dict(X=[3.7, -1.2, 9.5], pose=[0.02, -0.03, 0.01, 3.1, 0.05, -0.02], cam=[-0.1, 0.01, 0.7], obs=[0.04, -0.09], sd=0.004),
    # tiny rotation (below the reference's epsilon branch) and a small-but-regular one
    dict(X=[0.5, 0.4, 6.0], pose=[1e-9, -2e-9, 1e-9, 0.1, 0.0, 0.0], cam=[-0.1, 0.01, 0.7], obs=[0.1, 0.1], sd=0.004),
    dict(X=[0.5, 0.4, 6.0], pose=[1e-4, -2e-4, 1e-4, 0.1, 0.0, 0.0], cam=[-0.1, 0.01, 0.7], obs=[0.1, 0.1], sd=0.004),
    dict(X=[-2.0, 1.0, 5.0], pose=[1.2, -0.7, 2.1, -1.0, 0.3, 0.8], cam=[0.05, -0.002, 0.9], obs=[-0.2, 0.3], sd=0.01),
    # reprojection_errors_test.cc:131-137 FisheyeAnalyticErrorEvaluatesOK inputs (same point / pose / camera array)
    dict(model="fisheye", X=[1.0, 2.0, 3.0], pose=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6], cam=[0.3, 0.1, -0.03], obs=[0.5, 0.5], sd=0.1),
    # fisheye, wide angle (70 degrees off axis) and a point behind the image plane (theta > 90 degrees)
    dict(model="fisheye", X=[5.5, -1.0, 2.0], pose=[0.02, -0.03, 0.01, 0.1, 0.05, -0.02], cam=[-0.05, 0.004, 0.45], obs=[0.3, -0.1], sd=0.004),
    dict(model="fisheye", X=[2.0, 1.5, -0.5], pose=[0.3, 0.1, -0.2, 0.0, 0.1, 0.2], cam=[-0.02, 0.001, 0.4], obs=[0.4, 0.35], sd=0.004),
    # the other 2-D models at the reference's own test inputs (reprojection_errors_test.cc:114-174: the camera arrays are
    # handed to the functors as they are, i.e. read in the native order [projection][distortion][affine])
    dict(model="brown", X=[1.0, 2.0, 3.0], pose=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6], cam=[0.3, 1.0, 0.001, -0.02, 0.1, -0.03, 0.001, -0.005, 0.001],
         obs=[0.5, 0.5], sd=0.1),
    dict(model="fisheye_opencv", X=[1.0, 2.0, 3.0], pose=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6], cam=[0.3, 1.0, 0.001, -0.02, 0.1, -0.03, 0.001, -0.005],
         obs=[0.5, 0.5], sd=0.1),
    dict(model="fisheye62", X=[1.0, 2.0, 3.0], pose=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6],
         cam=[0.3, 1.0, 0.001, -0.02, 0.1, -0.03, 0.001, -0.005, 0.01, 0.006, 0.02, 0.003], obs=[0.5, 0.5], sd=0.1),
    dict(model="fisheye624", X=[1.0, 2.0, 3.0], pose=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6],
         cam=[0.3, 1.0, 0.001, -0.02, 0.1, -0.03, 0.001, -0.005, 0.01, 0.006, 0.02, 0.003, 0.001, -0.009, -0.01, 0.03], obs=[0.5, 0.5], sd=0.1),
    dict(model="dual", X=[1.0, 2.0, 3.0], pose=[0.1, 0.2, 0.3, 0.4, 0.5, 0.6], cam=[0.5, 0.3, 0.1, -0.03], obs=[0.5, 0.5], sd=0.1),
    # realistic parameter values in the native order
    dict(model="brown", X=[3.7, -1.2, 9.5], pose=[0.02, -0.03, 0.01, 3.1, 0.05, -0.02], cam=[-0.12, 0.03, -0.004, 0.001, -0.0007, 0.72, 1.003, 0.004, -0.006],
         obs=[0.04, -0.09], sd=0.004),
    dict(model="radial", X=[3.7, -1.2, 9.5], pose=[0.02, -0.03, 0.01, 3.1, 0.05, -0.02], cam=[-0.1, 0.01, 0.7, 0.998, -0.003, 0.002], obs=[0.04, -0.09], sd=0.004),
    dict(model="simple_radial", X=[0.5, 0.4, 6.0], pose=[0.2, -0.1, 0.05, 0.1, 0.0, 0.0], cam=[-0.08, 0.65, 1.01, 0.001, 0.002], obs=[0.1, 0.1], sd=0.004),
    dict(model="fisheye624", X=[5.5, -1.0, 2.0], pose=[0.02, -0.03, 0.01, 0.1, 0.05, -0.02],
         cam=[-0.03, 0.004, -0.0006, 0.0001, 0.00002, -0.000004, 0.0004, -0.0003, 0.0002, -0.0001, 0.0003, 0.00005, 0.42, 0.999, 0.002, -0.001],
         obs=[0.3, -0.1], sd=0.004),
]


def main():
    out = []
    for c in CASES:
        X = [mp.mpf(repr(v)) for v in c["X"]]
        pose = [mp.mpf(repr(v)) for v in c["pose"]]
        cam = [mp.mpf(repr(v)) for v in c["cam"]]
        obs = [mp.mpf(repr(v)) for v in c["obs"]]
        sd = mp.mpf(repr(c["sd"]))
        model = c.get("model", "perspective")
        r = residual(X, pose, cam, obs, sd, model)
        Jp = jac(lambda x: residual(x, pose, cam, obs, sd, model), X)
        Jc = jac(lambda x: residual(X, x, cam, obs, sd, model), pose)
        f = lambda m: [[float(v) for v in row] for row in m]
        rec = dict(c, residual=[float(v) for v in r], Jp=f(Jp), Jc=f(Jc))
        # intrinsics Jacobian: "Jk" for the models whose intrinsics the solver optimises ([k1, k2, focal]); "Jk_full" (2 x all native
        # parameters) for the others, groundwork for optimising them too
        rec["Jk" if model not in GENERIC else "Jk_full"] = f(jac(lambda x: residual(X, pose, x, obs, sd, model), cam))
        out.append(rec)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reprojection_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
