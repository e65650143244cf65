// camera_ref.cc -- TEST INFRASTRUCTURE: the REFERENCE's own camera projection / distortion functions compiled on this box.
//
// Compiled (oracle/Makefile, target _ref/libcamera_ref.so) against the reference's headers where they lie:
//   /root/reference/opensfm/src/geometry/camera_projections_functions.h   Perspective / Fisheye / Dual / Spherical ::Forward, ::Backward
//   /root/reference/opensfm/src/geometry/camera_distortions_functions.h   Disto2 / Disto24 / Disto2468 ::Forward, ::Backward
//   /root/reference/opensfm/src/foundation/newton_raphson.h (+ src/newton_raphson.cc: the scalar SolveDecr)
// Eigen is not installed here; the functions above are plain scalar code and compile against a declarations-only stand-in
// (stubs/Eigen/Eigen).  DistoBrown / Disto62 / Disto624 and the Affine / UniformScale / Pose functors need real Eigen types, so
// they are NOT covered: the adapter applies the affine stage itself (two divisions / multiplications, transformations_functions.h:
// 42-47,55-58,74-78 -- stated here, not compiled from there).
// What this pins: the oracle's bearings (ProjectGeneric::Backward) and forward projections of the perspective, fisheye, dual,
// radial, simple_radial, fisheye_opencv and spherical cameras against the reference's actual code, bit for bit where no libm call
// is involved.
#include <cmath>

#include <foundation/newton_raphson.h>
#include <geometry/camera_distortions_functions.h>
#include <geometry/camera_projections_functions.h>

using namespace geometry;

namespace {
template <class PROJ, class DISTO>
void backward(const double* proj_par, const double* disto_par, const double* aff, int na, const double* px, double* bearing) {
  double d[2], u[2];
  if (na == 1) {  // UniformScale::Backward
    d[0] = px[0] / aff[0];
    d[1] = px[1] / aff[0];
  } else {  // Affine::Backward
    d[0] = (px[0] - aff[2]) / aff[0];
    d[1] = (px[1] - aff[3]) / (aff[1] * aff[0]);
  }
  DISTO::Backward(d, disto_par, u);
  PROJ::Backward(u, proj_par, bearing);
}
template <class PROJ, class DISTO>
void forward(const double* proj_par, const double* disto_par, const double* aff, int na, const double* X, double* px) {
  double u[2], d[2];
  PROJ::Forward(X, proj_par, u);
  DISTO::Forward(u, disto_par, d);
  if (na == 1) {  // UniformScale::Forward
    px[0] = aff[0] * d[0];
    px[1] = aff[0] * d[1];
  } else {  // Affine::Forward
    px[0] = aff[0] * d[0] + aff[2];
    px[1] = aff[0] * aff[1] * d[1] + aff[3];
  }
}
}  // namespace

// Jacobians through the reference's own analytic derivative functions (ForwardDerivatives of PROJ and DISTO; the composition
// and the focal scaling are written out here, as ComposeForwardDerivatives needs Eigen): Jx = d pixel / d camera-frame point (2 x 3),
// Jk = d pixel / d [k1, k2, focal] (2 x 3) for the Perspective / Fisheye + Disto24 + UniformScale cameras (model 0 / 1).
extern "C" int ref_camera_jacobian(int model, const double* par, const double* X, int n, double* Jx, double* Jk) {
  if (model != 0 && model != 1) return 0;
  for (int i = 0; i < n; i++) {
    double u[2], jp[6], d[2], jd[8];
    if (model == 0)
      PerspectiveProjection::ForwardDerivatives<double, false>(X + 3 * i, par, u, jp);
    else
      FisheyeProjection::ForwardDerivatives<double, false>(X + 3 * i, par, u, jp);
    Disto24::ForwardDerivatives<double, true>(u, par, d, jd);  // rows of stride 4: [d/du, d/dv, d/dk1, d/dk2]
    const double f = par[2];
    for (int r = 0; r < 2; r++) {
      for (int c = 0; c < 3; c++) Jx[6 * i + 3 * r + c] = f * (jd[4 * r] * jp[c] + jd[4 * r + 1] * jp[3 + c]);
      Jk[6 * i + 3 * r + 0] = f * jd[4 * r + 2];
      Jk[6 * i + 3 * r + 1] = f * jd[4 * r + 3];
      Jk[6 * i + 3 * r + 2] = d[r];
    }
  }
  return 1;
}

// model ids as OSFM_CAMERA_*; par in the native order [projection][distortion][affine].  Returns 0 for models this file cannot cover.
extern "C" int ref_camera(int model, int backward_not_forward, const double* par, const double* in, int n, double* out) {
  for (int i = 0; i < n; i++) {
    const double* x = in + (backward_not_forward ? 2 : 3) * i;
    double* y = out + (backward_not_forward ? 3 : 2) * i;
#define OSFM_RUN(PROJ, DISTO, pp, dp, ap, na)                      \
  if (backward_not_forward)                                        \
    backward<PROJ, DISTO>(pp, dp, ap, na, x, y);                   \
  else                                                             \
    forward<PROJ, DISTO>(pp, dp, ap, na, x, y);
    switch (model) {
      case 0: OSFM_RUN(PerspectiveProjection, Disto24, par, par, par + 2, 1) break;
      case 1: OSFM_RUN(FisheyeProjection, Disto24, par, par, par + 2, 1) break;
      case 3: OSFM_RUN(FisheyeProjection, Disto2468, par, par, par + 4, 4) break;
      case 6: OSFM_RUN(DualProjection, Disto24, par, par + 1, par + 3, 1) break;
      case 7: OSFM_RUN(PerspectiveProjection, Disto24, par, par, par + 2, 4) break;
      case 8: OSFM_RUN(PerspectiveProjection, Disto2, par, par, par + 1, 4) break;
      case 9:
        if (backward_not_forward)
          SphericalProjection::Backward(x, par, y);
        else
          SphericalProjection::Forward(x, par, y);
        break;
      default: return 0;
    }
#undef OSFM_RUN
  }
  return 1;
}
