// camera_ref_eigen.cc -- TEST INFRASTRUCTURE: the REFERENCE's 2-D undistortions compiled on this box.
//
// DistoBrown / Disto62 / Disto624 ::Backward (geometry/camera_distortions_functions.h) are Newton iterations on a 2-vector written with
// Eigen types; they are compiled here, unmodified, from where they lie under /root/reference against a stand-in that implements the dozen
// Eigen operations they use (stubs_small_eigen/Eigen/Eigen, which says what is and is not the reference's arithmetic).  The projection
// (PerspectiveProjection / FisheyeProjection ::Backward, plain scalar code) is the reference's too; the affine stage is written out
// as in camera_ref.cc (transformations_functions.h:42-47,55-58,74-78 need more of Eigen).
// Pins, to rounding: the oracle's bearings for the brown, fisheye62 and fisheye624 cameras -- the three camera_ref.cc cannot cover.
#include <cmath>

#include <foundation/newton_raphson.h>
#include <geometry/camera_distortions_functions.h>
#include <geometry/camera_projections_functions.h>

using namespace geometry;

namespace {
template <class PROJ, class DISTO>
void backward(const double* proj_par, const double* disto_par, const double* aff, const double* px, double* bearing) {
  double d[2], u[2];
  d[0] = (px[0] - aff[2]) / aff[0];  // Affine::Backward
  d[1] = (px[1] - aff[3]) / (aff[1] * aff[0]);
  DISTO::Backward(d, disto_par, u);
  PROJ::Backward(u, proj_par, bearing);
}
}  // namespace

// model ids as OSFM_CAMERA_* (2 brown, 4 fisheye62, 5 fisheye624); par in the native order [distortion][affine: focal, aspect_ratio, cx, cy]
extern "C" int ref_camera_eigen_backward(int model, const double* par, const double* px, int n, double* out) {
  for (int i = 0; i < n; i++) {
    const double *x = px + 2 * i;
    double* y = out + 3 * i;
    switch (model) {
      case 2: backward<PerspectiveProjection, DistoBrown>(par, par, par + 5, x, y); break;
      case 4: backward<FisheyeProjection, Disto62>(par, par, par + 8, x, y); break;
      case 5: backward<FisheyeProjection, Disto624>(par, par, par + 12, x, y); break;
      default: return 0;
    }
  }
  return 1;
}
