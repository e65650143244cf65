// bundle_general_oracle.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never on the product path).
//
// CPU restatement of what bundle::BundleAdjuster::Run (opensfm/src/bundle/src/bundle_adjuster.cc:595-1121) minimises for the
// parameter blocks and residual families sfm::BAHelpers::Bundle wires (opensfm/src/sfm/src/ba_helpers.cc:581-763):
//
//   ReprojectionError2D / 3D          bundle/error/projection_errors.h:31-57,208-246 (the reference's AUTODIFF twins, restated with
//                                     forward-mode jets exactly as ceres::AutoDiffCostFunction evaluates them) through
//   WorldToCameraCoordinatesRig       bundle/error/error_utils.h:52-85
//   ProjectFunction of every camera   geometry/camera_instances.h:127-160 = PROJ (camera_projections_functions.h) o DISTO
//                                     (camera_distortions_functions.h) o AFFINE (transformations_functions.h:9-79), Forward only
//   DataPriorError (+ log scale)      bundle/error/prior_error.h:19-114, AddCameraPriorError bundle_adjuster.cc:568-593
//   ParameterBarrier                  bundle/error/parameters_errors.h:21-38 (dual camera transition in [0, 1])
//   SimilarityPriorTransform          bundle/data/bias.h:33-53 (rig instance position prior through the camera's GPS bias)
//   UpVectorError                     bundle/error/absolute_motion_errors.h:12-39 with ShotRotationFunctor (position_functors.h:45-63),
//                                     MultRotations (error_utils.h:14-24), CauchyLoss(1) (bundle_adjuster.cc:955-970)
//   Ceres 2.1 trust-region LM         defaults the reference leaves untouched (third party, restated from its documentation): jacobi
//                                     scaling, LM diagonal clamp [1e-6, 1e32], radius update, step quality, tolerances, loss
//                                     corrector without the second-order term (rho'' <= 0 for SoftLOne / Huber / Cauchy)
//
// Deliberately DIFFERENT from the product (opensfm_amd/csrc/ba_general.hip) wherever the mathematics allows a second route: every
// Jacobian here comes from jets (the product differentiates by hand), and the linear system is the FULL dense normal matrix over
// points and everything else, factorised directly (the product eliminates the points first).  Agreement of the two is therefore a
// check of the derivatives and of the elimination, not a comparison of a program with itself.
// PARITY STATUS: residual values pinned by the golden vectors of the C oracle's projections (tests/test_oracle_bundle_general.py);
// the LM trajectory is "parity unpinned" against Ceres itself (not available in this image).
#include <omp.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr double kEps = 2.220446049250313e-16;

// ---- forward-mode jets ----
template <int N>
struct Jet {
  double v;
  double d[N];
  Jet() : v(0.0) {
    for (int i = 0; i < N; i++) d[i] = 0.0;
  }
  Jet(double c) : v(c) {  // NOLINT: implicit, like ceres::Jet
    for (int i = 0; i < N; i++) d[i] = 0.0;
  }
  Jet(double c, int k) : v(c) {
    for (int i = 0; i < N; i++) d[i] = 0.0;
    d[k] = 1.0;
  }
};
#define JOP template <int N> inline Jet<N>
JOP operator+(const Jet<N>& a, const Jet<N>& b) { Jet<N> r; r.v = a.v + b.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
JOP operator-(const Jet<N>& a, const Jet<N>& b) { Jet<N> r; r.v = a.v - b.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
JOP operator-(const Jet<N>& a) { Jet<N> r; r.v = -a.v; for (int i = 0; i < N; i++) r.d[i] = -a.d[i]; return r; }
JOP operator*(const Jet<N>& a, const Jet<N>& b) { Jet<N> r; r.v = a.v * b.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
JOP operator/(const Jet<N>& a, const Jet<N>& b) { Jet<N> r; const double ib = 1.0 / b.v; r.v = a.v * ib; for (int i = 0; i < N; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
JOP operator+(const Jet<N>& a, double c) { Jet<N> r = a; r.v += c; return r; }
JOP operator+(double c, const Jet<N>& a) { return a + c; }
JOP operator-(const Jet<N>& a, double c) { Jet<N> r = a; r.v -= c; return r; }
JOP operator-(double c, const Jet<N>& a) { return Jet<N>(c) - a; }
JOP operator*(const Jet<N>& a, double c) { Jet<N> r; r.v = a.v * c; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * c; return r; }
JOP operator*(double c, const Jet<N>& a) { return a * c; }
JOP operator/(const Jet<N>& a, double c) { return a * (1.0 / c); }
JOP operator/(double c, const Jet<N>& a) { return Jet<N>(c) / a; }
JOP sqrt(const Jet<N>& a) { Jet<N> r; r.v = std::sqrt(a.v); const double h = 0.5 / r.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * h; return r; }
JOP sin(const Jet<N>& a) { Jet<N> r; r.v = std::sin(a.v); const double c = std::cos(a.v); for (int i = 0; i < N; i++) r.d[i] = a.d[i] * c; return r; }
JOP cos(const Jet<N>& a) { Jet<N> r; r.v = std::cos(a.v); const double s = -std::sin(a.v); for (int i = 0; i < N; i++) r.d[i] = a.d[i] * s; return r; }
JOP asin(const Jet<N>& a) { Jet<N> r; r.v = std::asin(a.v); const double k = 1.0 / std::sqrt(1.0 - a.v * a.v); for (int i = 0; i < N; i++) r.d[i] = a.d[i] * k; return r; }
JOP log(const Jet<N>& a) { Jet<N> r; r.v = std::log(a.v); const double ia = 1.0 / a.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * ia; return r; }
JOP atan2(const Jet<N>& y, const Jet<N>& x) { Jet<N> r; r.v = std::atan2(y.v, x.v); const double den = 1.0 / (x.v * x.v + y.v * y.v); for (int i = 0; i < N; i++) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * den; return r; }
template <int N> inline bool operator<(const Jet<N>& a, double c) { return a.v < c; }
template <int N> inline bool operator>(const Jet<N>& a, double c) { return a.v > c; }
#undef JOP
inline double val(double x) { return x; }
template <int N> inline double val(const Jet<N>& x) { return x.v; }

// ---- ceres/rotation.h ----
template <class T>
void AngleAxisRotatePoint(const T aa[3], const T pt[3], T out[3]) {
  const T th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (val(th2) > kEps) {
    const T th = sqrt(th2), ct = cos(th), st = sin(th), ti = T(1.0) / th;
    const T w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
    const T wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - ct);
    for (int i = 0; i < 3; i++) out[i] = pt[i] * ct + wxp[i] * st + w[i] * tmp;
  } else {
    const T wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
    for (int i = 0; i < 3; i++) out[i] = pt[i] + wxp[i];
  }
}
template <class T>
void AngleAxisToQuaternion(const T aa[3], T q[4]) {
  const T th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (val(th2) > 0.0) {
    const T th = sqrt(th2), half = th * 0.5, k = sin(half) / th;
    q[0] = cos(half);
    for (int i = 0; i < 3; i++) q[1 + i] = aa[i] * k;
  } else {
    q[0] = T(1.0);
    for (int i = 0; i < 3; i++) q[1 + i] = aa[i] * 0.5;
  }
}
template <class T>
void QuaternionToAngleAxis(const T q[4], T aa[3]) {
  const T s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (val(s2) > 0.0) {
    const T s = sqrt(s2);
    const T two_theta = T(2.0) * (val(q[0]) < 0.0 ? atan2(-s, -q[0]) : atan2(s, q[0]));
    const T k = two_theta / s;
    for (int i = 0; i < 3; i++) aa[i] = q[1 + i] * k;
  } else {
    for (int i = 0; i < 3; i++) aa[i] = q[1 + i] * 2.0;
  }
}
template <class T>
void MultRotations(const T r1[3], const T r2[3], T out[3]) {  // error_utils.h:14-24
  T a[4], b[4], c[4];
  AngleAxisToQuaternion(r1, a);
  AngleAxisToQuaternion(r2, b);
  c[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  c[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  c[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  c[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  QuaternionToAngleAxis(c, out);
}

// error_utils.h:52-85
template <class T>
void WorldToLocal(const T* shot, const T world[3], T local[3]) {
  const T pt[3] = {world[0] - shot[3], world[1] - shot[4], world[2] - shot[5]};
  const T Rt[3] = {-shot[0], -shot[1], -shot[2]};
  AngleAxisRotatePoint(Rt, pt, local);
}
template <class T>
void WorldToCameraCoordinatesRig(const T* rig_instance, const T* rig_camera, const T world[3], T cam[3]) {
  if (rig_camera) {
    T ip[3];
    WorldToLocal(rig_instance, world, ip);
    WorldToLocal(rig_camera, ip, cam);
  } else {
    WorldToLocal(rig_instance, world, cam);
  }
}

// ---- cameras: Forward of ProjectGeneric<PROJ, DISTO, AFF> (camera_instances.h:127-160) ----
enum { PERSPECTIVE = 0, FISHEYE, BROWN, FISHEYE_OPENCV, FISHEYE62, FISHEYE624, DUAL, RADIAL, SIMPLE_RADIAL, SPHERICAL };
int num_params(int model) {
  static const int n[10] = {3, 3, 9, 8, 12, 16, 4, 6, 5, 0};
  return n[model];
}
template <class T>
void perspective_forward(const T p[3], T out[2]) {  // camera_projections_functions.h:88-95
  out[0] = p[0] / p[2];
  out[1] = p[1] / p[2];
}
template <class T>
void fisheye_forward(const T p[3], T out[2]) {  // camera_projections_functions.h:9-24
  const T r2 = p[0] * p[0] + p[1] * p[1];
  const T r = sqrt(r2);
  if (val(r) < 1e-8) {  // the reference guards the division with a perspective fall-back near the axis
    out[0] = p[0] / p[2];
    out[1] = p[1] / p[2];
    return;
  }
  const T theta = atan2(r, p[2]);
  out[0] = theta / r * p[0];
  out[1] = theta / r * p[1];
}
template <class T>
void project(int model, const T* par, const T Xc[3], T out[2]) {
  T uv[2];
  const T* kd = par;
  int kind, nd, na;
  switch (model) {
    case PERSPECTIVE: perspective_forward(Xc, uv); kind = 1; nd = 2; na = 1; break;
    case FISHEYE: fisheye_forward(Xc, uv); kind = 1; nd = 2; na = 1; break;
    case BROWN: perspective_forward(Xc, uv); kind = 3; nd = 5; na = 4; break;
    case FISHEYE_OPENCV: fisheye_forward(Xc, uv); kind = 2; nd = 4; na = 4; break;
    case FISHEYE62: fisheye_forward(Xc, uv); kind = 4; nd = 8; na = 4; break;
    case FISHEYE624: fisheye_forward(Xc, uv); kind = 5; nd = 12; na = 4; break;
    case DUAL: {  // camera_projections_functions.h:122-134
      T a[2], b[2];
      perspective_forward(Xc, a);
      fisheye_forward(Xc, b);
      for (int i = 0; i < 2; i++) uv[i] = par[0] * a[i] + (T(1.0) - par[0]) * b[i];
      kd = par + 1;
      kind = 1; nd = 2; na = 1;
    } break;
    case RADIAL: perspective_forward(Xc, uv); kind = 1; nd = 2; na = 4; break;
    default: perspective_forward(Xc, uv); kind = 0; nd = 1; na = 4; break;  // SIMPLE_RADIAL
  }
  const T x = uv[0], y = uv[1];
  const T r2 = x * x + y * y;
  T rad;
  switch (kind) {  // camera_distortions_functions.h: Disto2 :58-69, Disto24 :103-116, Disto2468 :205-222, Brown :640-668, 62 / 624 :450-520
    case 0: rad = T(1.0) + r2 * kd[0]; break;
    case 1: rad = T(1.0) + r2 * (kd[0] + kd[1] * r2); break;
    case 2: rad = T(1.0) + r2 * (kd[0] + r2 * (kd[1] + r2 * (kd[2] + r2 * kd[3]))); break;
    case 3: rad = T(1.0) + r2 * (kd[0] + r2 * (kd[1] + r2 * kd[2])); break;
    default: rad = T(1.0) + r2 * (kd[0] + r2 * (kd[1] + r2 * (kd[2] + r2 * (kd[3] + r2 * (kd[4] + r2 * kd[5]))))); break;
  }
  T dx = x * rad, dy = y * rad;
  if (kind >= 3) {
    const T p1 = kind == 3 ? kd[3] : kd[6], p2 = kind == 3 ? kd[4] : kd[7];
    dx = dx + (T(2.0) * p1 * x * y + p2 * (r2 + T(2.0) * x * x));
    dy = dy + (T(2.0) * p2 * x * y + p1 * (r2 + T(2.0) * y * y));
  }
  if (kind == 5) {
    dx = dx + (kd[8] * r2 + kd[9] * r2 * r2);
    dy = dy + (kd[10] * r2 + kd[11] * r2 * r2);
  }
  const T* ka = kd + nd;
  if (na == 1) {  // UniformScale
    out[0] = ka[0] * dx;
    out[1] = ka[0] * dy;
  } else {  // Affine: focal, aspect ratio, cx, cy
    out[0] = ka[0] * dx + ka[2];
    out[1] = ka[0] * ka[1] * dy + ka[3];
  }
}

struct Problem {  // mirrors osfm_bundle_problem (include/osfm_mi355.h)
  int32_t n_cameras;
  const int32_t* cam_model;
  double* cam_params;
  const double* cam_prior;
  const double* cam_sigma;
  const uint8_t* cam_fixed;
  double* bias;
  const uint8_t* bias_fixed;
  int32_t n_rig_cameras;
  double* rig_camera_pose;
  const double* rig_camera_prior;
  const double* rig_camera_sigma;
  const uint8_t* rig_camera_fixed;
  int32_t n_rig_instances;
  double* rig_instance_pose;
  const uint8_t* rig_instance_fixed;
  const double* rig_instance_gps;
  const double* rig_instance_gps_sigma;
  const int32_t* rig_instance_bias_camera;
  int32_t n_shots;
  const int32_t *shot_rig_instance, *shot_rig_camera, *shot_camera;
  const double* shot_up;
  const double* shot_up_sigma;
  int32_t n_points;
  double* points;
  const uint8_t* point_fixed;
  const double* point_prior;
  const double* point_prior_sigma;
  const uint8_t* point_prior_has_altitude;
  int64_t n_obs;
  const int32_t *obs_shot, *obs_point;
  const double* obs_xy;
  const double* obs_sigma;
  double* reproj_err;
  const double *shot_pan, *shot_pan_sigma, *shot_tilt, *shot_tilt_sigma, *shot_roll, *shot_roll_sigma;
  const double *obs_depth, *obs_depth_sigma;  // n_obs each or null; sigma <= 0: no depth prior for the observation
  const uint8_t* obs_depth_radial;            // n_obs or null = all radial (map::Depth::is_radial)
};
struct Options {  // mirrors osfm_ba_options
  int32_t loss;
  double loss_threshold;
  int32_t max_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance, initial_radius;
  int32_t verbose;
  double pcg_tolerance;
  int32_t pcg_max_iterations, preconditioner;
};
struct Report {
  int32_t iterations, successful_steps, termination;
  double initial_cost, final_cost;
  double cost_history[256];
};

void loss_eval(int loss, double a, double s, double* rho, double* rho1) {
  const double b = a * a;
  switch (loss) {
    case 1: { const double sum = 1.0 + s / b, tmp = std::sqrt(sum); *rho = 2.0 * b * (tmp - 1.0); *rho1 = 1.0 / tmp; } break;
    case 2: if (s > b) { const double r = std::sqrt(s); *rho = 2.0 * a * r - b; *rho1 = a / r; } else { *rho = s; *rho1 = 1.0; } break;
    case 3: { const double sum = 1.0 + s / b; *rho = b * std::log(sum); *rho1 = 1.0 / sum; } break;
    default: *rho = s; *rho1 = 1.0;
  }
}

// The whole problem as a list of residual blocks over a flat parameter vector
struct Layout {
  std::vector<int> cam, bias, rc, inst, pt;  // start index in the free-parameter vector, or -1 (constant block)
  int n = 0;
};
struct State {
  std::vector<double> cam, bias, rc, inst, pts;
};

struct Accum {  // normal equations: dense, or (nr >= 0) in ARROW form for problems with many points
  int n;
  std::vector<double> H, g;
  double cost = 0.0;
  // Arrow form.  The free-parameter vector lists the points last (Layout below): unknowns [0, nr) are everything else ("reduced"), then
  // three per free point.  A residual block touches at most one point, so H = [R W; W^T blockdiag(Hpp)]: R dense nr x nr, per point the
  // 3 x 3 block Hpp and the columns of W it has (reduced index -> 3 values).  Same sums as the dense form, entry by entry; the solve
  // eliminates the points first (Schur complement) instead of factorising the whole matrix.
  int nr = -1;
  std::vector<double> R, Hpp;            // nr x nr, 9 per point
  std::vector<std::vector<int>> wcol;    // per point: reduced indices
  std::vector<std::vector<double>> wval; // per point: 3 values per entry of wcol
  explicit Accum(int n_) : n(n_), H((size_t)n_ * n_, 0.0), g(n_, 0.0) {}
  Accum(int n_, int nr_) : n(n_), g(n_, 0.0), nr(nr_), R((size_t)nr_ * nr_, 0.0), Hpp((size_t)(n_ - nr_) * 3, 0.0), wcol((size_t)(n_ - nr_) / 3), wval((size_t)(n_ - nr_) / 3) {}
  double *w_entry(int pt, int col) {
    std::vector<int> &c = wcol[(size_t)pt];
    for (size_t k = 0; k < c.size(); k++)
      if (c[k] == col) return &wval[(size_t)pt][3 * k];
    c.push_back(col);
    wval[(size_t)pt].insert(wval[(size_t)pt].end(), 3, 0.0);
    return &wval[(size_t)pt][3 * (c.size() - 1)];
  }
  double diag(int i) const { return nr < 0 ? H[(size_t)i * n + i] : (i < nr ? R[(size_t)i * nr + i] : Hpp[(size_t)((i - nr) / 3) * 9 + 4 * ((i - nr) % 3)]); }
};

// one residual block: nres jets over N local parameters with global indices idx[N] (-1: constant); wt = sqrt(rho')
template <int N>
void add_block(Accum* A, const Jet<N>* r, int nres, double wt, const int* idx) {
  if (!A) return;
  for (int x = 0; x < N; x++) {
    if (idx[x] < 0) continue;
    double g = 0.0;
    for (int e = 0; e < nres; e++) g += wt * r[e].d[x] * wt * r[e].v;
    A->g[idx[x]] += g;
    for (int y = 0; y < N; y++) {
      if (idx[y] < 0) continue;
      double h = 0.0;
      for (int e = 0; e < nres; e++) h += wt * r[e].d[x] * wt * r[e].d[y];
      if (A->nr < 0) {
        A->H[(size_t)idx[x] * A->n + idx[y]] += h;
      } else {
        const int ix = idx[x], iy = idx[y], nr = A->nr;
        if (ix < nr && iy < nr) A->R[(size_t)ix * nr + iy] += h;
        else if (ix >= nr && iy >= nr) A->Hpp[(size_t)((ix - nr) / 3) * 9 + 3 * ((ix - nr) % 3) + (iy - nr) % 3] += h;
        else if (ix < nr) A->w_entry((iy - nr) / 3, ix)[(iy - nr) % 3] += h;  // (the mirror entry is not stored)
      }
    }
  }
}

void bearing_of(const double xy[2], double b[3]) {  // ReprojectionError3D's constructor, projection_errors.h:214-219
  const double lon = xy[0] * 2 * M_PI, lat = -xy[1] * 2 * M_PI;
  b[0] = std::cos(lat) * std::sin(lon);
  b[1] = -std::sin(lat);
  b[2] = std::cos(lat) * std::cos(lon);
}

bool rig_camera_useful(const Problem& P, int q) {  // IsRigCameraUseful, bundle_adjuster.cc:17-20
  bool zero = true;
  for (int k = 0; k < 6; k++) zero = zero && P.rig_camera_pose[6 * q + k] == 0.0;
  return !(P.rig_camera_fixed[q] && zero);
}

// cost (and, with A, the normal equations) at a state
double evaluate(const Problem& P, const Options& O, const Layout& L, const State& X, const std::vector<uint8_t>& useful, Accum* A,
                double* reproj /* n_obs x 3, sigma 1, or null */) {
  double cost = 0.0;
  // ---- reprojection errors: parameters [camera 16 | rig instance 6 | rig camera 6 | point 3] ----
  constexpr int NR_ = 31;
  // The jets of a chunk of observations are evaluated on all cores; the sums (cost, normal equations) are then taken by ONE thread in
  // observation order, so the result does not depend on the thread count, bit for bit.
  struct ObsOut {
    Jet<NR_> r[3], rd;
    int idx[NR_], nres;
    double wt, half_rho, wt_d, half_rho_d;
    bool has_depth;
  };
  const int64_t CH = 4096;
  const int jet_threads = std::max(1, std::min(omp_get_max_threads(), 16));  // 256 observations per thread and chunk: more threads only add barriers
  std::vector<ObsOut> buf((size_t)std::min<int64_t>(CH, std::max<int64_t>(P.n_obs, 1)));
  for (int64_t o0 = 0; o0 < P.n_obs; o0 += CH) {
    const int64_t o1 = std::min(P.n_obs, o0 + CH);
#pragma omp parallel for schedule(static) num_threads(jet_threads) if (o1 - o0 > 256)
    for (int64_t o = o0; o < o1; o++) {
      ObsOut &B = buf[(size_t)(o - o0)];
      const int s = P.obs_shot[o], p = P.obs_point[o];
      const int c = P.shot_camera[s], i = P.shot_rig_instance[s], q = P.shot_rig_camera[s];
      const int model = P.cam_model[c], nk = num_params(model);
      typedef Jet<NR_> T;
      T cam[16], inst[6], rcam[6], pt[3];
      int *idx = B.idx;
      for (int k = 0; k < 16; k++) {
        cam[k] = T(X.cam[16 * c + k], k);
        idx[k] = (L.cam[c] >= 0 && k < nk) ? L.cam[c] + k : -1;
      }
      for (int k = 0; k < 6; k++) {
        inst[k] = T(X.inst[6 * i + k], 16 + k);
        idx[16 + k] = L.inst[i] >= 0 ? L.inst[i] + k : -1;
        rcam[k] = T(X.rc[6 * q + k], 22 + k);
        idx[22 + k] = (L.rc[q] >= 0 && useful[q]) ? L.rc[q] + k : -1;
      }
      for (int k = 0; k < 3; k++) {
        pt[k] = T(X.pts[3 * p + k], 28 + k);
        idx[28 + k] = L.pt[p] >= 0 ? L.pt[p] + k : -1;
      }
      T Xc[3];
      T *r = B.r;
      WorldToCameraCoordinatesRig<T>(inst, useful[q] ? rcam : nullptr, pt, Xc);
      const double is = 1.0 / P.obs_sigma[o];
      int nres;
      if (model == SPHERICAL) {
        nres = 3;
        const T n = sqrt(Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2]);
        double b[3];
        bearing_of(P.obs_xy + 2 * o, b);
        for (int a = 0; a < 3; a++) r[a] = (Xc[a] / n - b[a]) * is;
      } else {
        nres = 2;
        T pr[2];
        project<T>(model, cam, Xc, pr);
        r[0] = (pr[0] - P.obs_xy[2 * o]) * is;
        r[1] = (pr[1] - P.obs_xy[2 * o + 1]) * is;
      }
      B.nres = nres;
      double sq = 0.0;
      for (int e = 0; e < nres; e++) sq += r[e].v * r[e].v;
      double rho, rho1;
      loss_eval(O.loss, O.loss_threshold, sq, &rho, &rho1);
      B.half_rho = 0.5 * rho;
      B.wt = std::sqrt(rho1);
      if (reproj)
        for (int e = 0; e < 3; e++) reproj[3 * o + e] = e < nres ? r[e].v * P.obs_sigma[o] : 0.0;
      // RelativeDepthError (bundle/error/relative_depth_error.h:11-46, added right after the reprojection block of the same observation
      // with the SAME loss function, bundle_adjuster.cc:497-528,812): one residual on [rig instance | rig camera | point]
      B.has_depth = P.obs_depth && P.obs_depth_sigma && P.obs_depth_sigma[o] > 0;
      if (B.has_depth) {
        T depth = Xc[2];
        if (!P.obs_depth_radial || P.obs_depth_radial[o]) depth = sqrt(Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2]);
        B.rd = (depth - P.obs_depth[o]) * (1.0 / P.obs_depth_sigma[o]);
        double rho_d, rho1_d;
        loss_eval(O.loss, O.loss_threshold, B.rd.v * B.rd.v, &rho_d, &rho1_d);
        B.half_rho_d = 0.5 * rho_d;
        B.wt_d = std::sqrt(rho1_d);
      }
    }
    for (int64_t o = o0; o < o1; o++) {
      const ObsOut &B = buf[(size_t)(o - o0)];
      cost += B.half_rho;
      add_block<NR_>(A, B.r, B.nres, B.wt, B.idx);
      if (B.has_depth) {
        cost += B.half_rho_d;
        add_block<NR_>(A, &B.rd, 1, B.wt_d, B.idx);
      }
    }
  }
  // ---- camera priors (+ the dual barrier) ----
  for (int c = 0; c < P.n_cameras; c++) {
    const int model = P.cam_model[c], nk = num_params(model);
    typedef Jet<16> T;
    T r[17];
    int idx[16];
    int ifocal, na;
    switch (model) {
      case PERSPECTIVE: case FISHEYE: ifocal = 2; na = 1; break;
      case BROWN: ifocal = 5; na = 4; break;
      case FISHEYE_OPENCV: ifocal = 4; na = 4; break;
      case FISHEYE62: ifocal = 8; na = 4; break;
      case FISHEYE624: ifocal = 12; na = 4; break;
      case DUAL: ifocal = 3; na = 1; break;
      case RADIAL: ifocal = 2; na = 4; break;
      case SIMPLE_RADIAL: ifocal = 1; na = 4; break;
      default: ifocal = -1; na = 0;
    }
    for (int k = 0; k < 16; k++) idx[k] = (L.cam[c] >= 0 && k < nk) ? L.cam[c] + k : -1;
    int nres = 0;
    for (int k = 0; k < nk; k++) {
      const T v(X.cam[16 * c + k], k);
      const double scale = 1.0 / std::max(P.cam_sigma[16 * c + k], kEps);
      const bool logarithmic = k == ifocal || (na == 4 && k == ifocal + 1);  // Focal and AspectRatio, bundle_adjuster.cc:576-583
      r[nres++] = (logarithmic ? log(v / P.cam_prior[16 * c + k]) : v - P.cam_prior[16 * c + k]) * scale;
    }
    if (model == DUAL) {
      const T v(X.cam[16 * c], 0);
      const double zero = 2.0 * std::log((1.0 - 0.0) * 0.5);
      r[nres++] = log(v - 0.0 + 1e-10) + log(1.0 - v + 1e-10) + zero;
    }
    for (int e = 0; e < nres; e++) cost += 0.5 * r[e].v * r[e].v;
    add_block<16>(A, r, nres, 1.0, idx);
  }
  // ---- rig camera pose priors ----
  if (P.rig_camera_prior && P.rig_camera_sigma)
    for (int q = 0; q < P.n_rig_cameras; q++) {
      typedef Jet<6> T;
      T r[6];
      int idx[6];
      for (int k = 0; k < 6; k++) {
        idx[k] = L.rc[q] >= 0 ? L.rc[q] + k : -1;
        r[k] = (T(X.rc[6 * q + k], k) - P.rig_camera_prior[6 * q + k]) * (1.0 / std::max(P.rig_camera_sigma[6 * q + k], kEps));
        cost += 0.5 * r[k].v * r[k].v;
      }
      add_block<6>(A, r, 6, 1.0, idx);
    }
  // ---- rig instance position priors through the bias: DataPriorError<Pose, SimilarityPriorTransform> on TX, TY, TZ ----
  if (P.rig_instance_gps && P.rig_instance_gps_sigma)
    for (int i = 0; i < P.n_rig_instances; i++) {
      if (!(P.rig_instance_gps_sigma[3 * i] > 0)) continue;
      const int bc = P.rig_instance_bias_camera[i];
      typedef Jet<13> T;
      T b[7], point[3], rp[3], r[3];
      int idx[13];
      for (int k = 0; k < 6; k++) idx[k] = L.inst[i] >= 0 ? L.inst[i] + k : -1;
      for (int k = 0; k < 7; k++) {
        b[k] = T(X.bias[7 * bc + k], 6 + k);
        idx[6 + k] = L.bias[bc] >= 0 ? L.bias[bc] + k : -1;
      }
      for (int k = 0; k < 3; k++) point[k] = T(P.rig_instance_gps[3 * i + k]);
      AngleAxisRotatePoint<T>(b, point, rp);  // R = parameters[RX..RZ] of the similarity, t = parameters[TX..TZ]
      for (int k = 0; k < 3; k++) {
        const T prior = b[6] * rp[k] + b[3 + k];
        r[k] = (T(X.inst[6 * i + 3 + k], 3 + k) - prior) * (1.0 / std::max(P.rig_instance_gps_sigma[3 * i + k], kEps));
        cost += 0.5 * r[k].v * r[k].v;
      }
      add_block<13>(A, r, 3, 1.0, idx);
    }
  // ---- point priors ----
  if (P.point_prior && P.point_prior_sigma)
    for (int p = 0; p < P.n_points; p++) {
      if (!(P.point_prior_sigma[3 * p] > 0)) continue;
      const int n = (!P.point_prior_has_altitude || P.point_prior_has_altitude[p]) ? 3 : 2;
      typedef Jet<3> T;
      T r[3];
      int idx[3];
      for (int k = 0; k < 3; k++) idx[k] = L.pt[p] >= 0 ? L.pt[p] + k : -1;
      for (int k = 0; k < n; k++) {
        r[k] = (T(X.pts[3 * p + k], k) - P.point_prior[3 * p + k]) * (1.0 / std::max(P.point_prior_sigma[3 * p + k], kEps));
        cost += 0.5 * r[k].v * r[k].v;
      }
      add_block<3>(A, r, n, 1.0, idx);
    }
  // ---- absolute up vectors, CauchyLoss(1) ----
  if (P.shot_up && P.shot_up_sigma)
    for (int s = 0; s < P.n_shots; s++) {
      if (!(P.shot_up_sigma[s] > 0)) continue;
      const int i = P.shot_rig_instance[s], q = P.shot_rig_camera[s];
      typedef Jet<12> T;
      T ri[3], rr[3], R[3], acc[3], z[3], r[3];
      int idx[12];
      const double* u = P.shot_up + 3 * s;
      const double nrm = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
      for (int k = 0; k < 3; k++) {
        ri[k] = T(X.inst[6 * i + k], k);
        rr[k] = T(X.rc[6 * q + k], 6 + k);
        acc[k] = T(u[k] / nrm);
      }
      for (int k = 0; k < 6; k++) {
        idx[k] = L.inst[i] >= 0 ? L.inst[i] + k : -1;
        idx[6 + k] = L.rc[q] >= 0 ? L.rc[q] + k : -1;
      }
      MultRotations<T>(ri, rr, R);  // ShotRotationFunctor(0, 1): the rig camera block is always handed over
      AngleAxisRotatePoint<T>(R, acc, z);
      double sq = 0.0;
      for (int k = 0; k < 3; k++) {
        r[k] = (z[k] - (k == 2 ? 1.0 : 0.0)) * (1.0 / P.shot_up_sigma[s]);
        sq += r[k].v * r[k].v;
      }
      double rho, rho1;
      loss_eval(3, 1.0, sq, &rho, &rho1);
      cost += 0.5 * rho;
      add_block<12>(A, r, 3, std::sqrt(rho1), idx);
    }
  // ---- absolute pan / tilt / roll (absolute_motion_errors.h:40-137), CauchyLoss(1) each (bundle_adjuster.cc:972-1022) ----
  for (int which = 0; which < 3; which++) {
    const double* ang = which == 0 ? P.shot_pan : (which == 1 ? P.shot_tilt : P.shot_roll);
    const double* sg = which == 0 ? P.shot_pan_sigma : (which == 1 ? P.shot_tilt_sigma : P.shot_roll_sigma);
    if (!ang || !sg) continue;
    for (int s = 0; s < P.n_shots; s++) {
      if (!(sg[s] > 0)) continue;
      const int i = P.shot_rig_instance[s], q = P.shot_rig_camera[s];
      typedef Jet<12> T;
      T ri[3], rr[3], R[3], zw[3], xw[3], r[1];
      int idx[12];
      for (int k = 0; k < 3; k++) {
        ri[k] = T(X.inst[6 * i + k], k);
        rr[k] = T(X.rc[6 * q + k], 6 + k);
      }
      for (int k = 0; k < 6; k++) {
        idx[k] = L.inst[i] >= 0 ? L.inst[i] + k : -1;
        idx[6 + k] = L.rc[q] >= 0 ? L.rc[q] + k : -1;
      }
      MultRotations<T>(ri, rr, R);
      const T ex[3] = {T(1.0), T(0.0), T(0.0)}, ez[3] = {T(0.0), T(0.0), T(1.0)};
      AngleAxisRotatePoint<T>(R, ez, zw);
      auto diff = [](const T& a, double b) {  // DiffBetweenAngles, error_utils.h:88-97
        T dd = a - b;
        if (dd.v > M_PI) return dd - 2 * M_PI;
        if (dd.v < -M_PI) return dd + 2 * M_PI;
        return dd;
      };
      bool zero = false;
      if (which == 0) {
        if (std::fabs(zw[0].v) < 1e-8 && std::fabs(zw[1].v) < 1e-8) zero = true;
        else r[0] = diff(atan2(zw[0], zw[1]), ang[s]) * (1.0 / sg[s]);
      } else if (which == 1) {
        const T l = sqrt(zw[0] * zw[0] + zw[1] * zw[1]);
        r[0] = diff(-atan2(zw[2], l), ang[s]) * (1.0 / sg[s]);
      } else {
        AngleAxisRotatePoint<T>(R, ex, xw);
        T a[3] = {zw[1], -zw[0], T(0.0)};
        const T la = sqrt(a[0] * a[0] + a[1] * a[1]);
        if (la.v < 1e-5) {
          zero = true;
        } else {
          a[0] = a[0] / la;
          a[1] = a[1] / la;
          const T b[3] = {xw[1] * a[2] - xw[2] * a[1], xw[2] * a[0] - xw[0] * a[2], xw[0] * a[1] - xw[1] * a[0]};  // ceres::CrossProduct
          const T sin_roll = zw[0] * b[0] + zw[1] * b[1] + zw[2] * b[2];
          if (sin_roll.v <= -(1.0 - 1e-5)) zero = true;
          else r[0] = diff(asin(sin_roll), ang[s]) * (1.0 / sg[s]);
        }
      }
      if (zero) continue;
      double rho, rho1;
      loss_eval(3, 1.0, r[0].v * r[0].v, &rho, &rho1);
      cost += 0.5 * rho;
      add_block<12>(A, r, 1, std::sqrt(rho1), idx);
    }
  }
  if (A) A->cost = cost;
  return cost;
}

// In place, lower.  Round 6: the factorisation keeps to the ENVELOPE of the matrix (first[i] = the first non-zero column of row i; the factor's
// fill stays inside it): the reduced system of a sequence is a band plus a few dense border rows, and the plain triple loop -- 9e12 multiply-adds
// at the 30 016 unknowns of configs[4] -- was what kept the oracle from following the product to that size.  The terms left out are products
// with exact zeros, so every entry is the same sum as before: the same bits (tests/test_oracle_bundle_general.py compares the two forms).
bool cholesky_solve(std::vector<double>& Amat, std::vector<double>& b, int n) {
  std::vector<int> first((size_t)n);
#pragma omp parallel for schedule(static) num_threads(8) if (n > 512)
  for (int i = 0; i < n; i++) {
    int k = 0;
    while (k < i && Amat[(size_t)i * n + k] == 0.0) k++;
    first[(size_t)i] = k;
  }
  for (int j = 0; j < n; j++) {
    double dgn = Amat[(size_t)j * n + j];
    const int fj = first[(size_t)j];
    for (int k = fj; k < j; k++) dgn -= Amat[(size_t)j * n + k] * Amat[(size_t)j * n + k];
    if (!(dgn > 0) || !std::isfinite(dgn)) return false;
    const double l = std::sqrt(dgn);
    Amat[(size_t)j * n + j] = l;
#pragma omp parallel for schedule(static) num_threads(8) if (n - j > 512)
    for (int i = j + 1; i < n; i++) {
      if (first[(size_t)i] > j) continue;  // outside the envelope: the entry is zero and stays zero
      double v = Amat[(size_t)i * n + j];
      for (int k = std::max(first[(size_t)i], fj); k < j; k++) v -= Amat[(size_t)i * n + k] * Amat[(size_t)j * n + k];
      Amat[(size_t)i * n + j] = v / l;
    }
  }
  for (int i = 0; i < n; i++) {
    double v = b[i];
    for (int k = first[(size_t)i]; k < i; k++) v -= Amat[(size_t)i * n + k] * b[k];
    b[i] = v / Amat[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {  // (column i of the factor: rows k whose envelope reaches column i)
    double v = b[i];
    for (int k = i + 1; k < n; k++)
      if (first[(size_t)k] <= i) v -= Amat[(size_t)k * n + i] * b[k];
    b[i] = v / Amat[(size_t)i * n + i];
  }
  return true;
}

}  // namespace

extern "C" int oracle_bundle_solve(Problem* P, const Options* O, Report* Rp) {
  memset(Rp, 0, sizeof(*Rp));
  Layout L;
  const int NC = P->n_cameras, NR = P->n_rig_cameras, NI = P->n_rig_instances, NP = P->n_points;
  L.cam.resize(NC); L.bias.resize(NC); L.rc.resize(NR); L.inst.resize(NI); L.pt.resize(NP);
  std::vector<uint8_t> useful(NR);
  int n = 0;
  for (int i = 0; i < NI; i++) {
    const bool fixed = P->rig_instance_fixed && P->rig_instance_fixed[i];
    L.inst[i] = fixed ? -1 : n;
    if (!fixed) n += 6;
  }
  for (int q = 0; q < NR; q++) {
    useful[q] = rig_camera_useful(*P, q);
    L.rc[q] = P->rig_camera_fixed[q] ? -1 : n;
    if (!P->rig_camera_fixed[q]) n += 6;
  }
  for (int c = 0; c < NC; c++) {
    const int nk = num_params(P->cam_model[c]);
    const bool fixed = P->cam_fixed[c] || nk == 0;
    L.cam[c] = fixed ? -1 : n;
    if (!fixed) n += nk;
  }
  for (int c = 0; c < NC; c++) {
    const bool fixed = !P->bias || !P->bias_fixed || P->bias_fixed[c];
    L.bias[c] = fixed ? -1 : n;
    if (!fixed) n += 7;
  }
  for (int p = 0; p < NP; p++) {
    const bool fixed = P->point_fixed && P->point_fixed[p];
    L.pt[p] = fixed ? -1 : n;
    if (!fixed) n += 3;
  }
  L.n = n;
  State X;
  X.cam.assign(P->cam_params, P->cam_params + 16 * NC);
  X.bias.assign((size_t)7 * NC, 0.0);
  for (int c = 0; c < NC; c++) {
    if (P->bias)
      for (int k = 0; k < 7; k++) X.bias[7 * c + k] = P->bias[7 * c + k];
    else
      X.bias[7 * c + 6] = 1.0;
  }
  X.rc.assign(P->rig_camera_pose, P->rig_camera_pose + 6 * NR);
  X.inst.assign(P->rig_instance_pose, P->rig_instance_pose + 6 * NI);
  X.pts.assign(P->points, P->points + 3 * NP);

  auto flat = [&](const State& s, std::vector<double>& v) {
    v.assign(n, 0.0);
    for (int c = 0; c < NC; c++) {
      if (L.cam[c] >= 0)
        for (int k = 0; k < num_params(P->cam_model[c]); k++) v[L.cam[c] + k] = s.cam[16 * c + k];
      if (L.bias[c] >= 0)
        for (int k = 0; k < 7; k++) v[L.bias[c] + k] = s.bias[7 * c + k];
    }
    for (int q = 0; q < NR; q++)
      if (L.rc[q] >= 0)
        for (int k = 0; k < 6; k++) v[L.rc[q] + k] = s.rc[6 * q + k];
    for (int i = 0; i < NI; i++)
      if (L.inst[i] >= 0)
        for (int k = 0; k < 6; k++) v[L.inst[i] + k] = s.inst[6 * i + k];
    for (int p = 0; p < NP; p++)
      if (L.pt[p] >= 0)
        for (int k = 0; k < 3; k++) v[L.pt[p] + k] = s.pts[3 * p + k];
  };
  auto unflat = [&](const std::vector<double>& v, State& s) {
    for (int c = 0; c < NC; c++) {
      if (L.cam[c] >= 0)
        for (int k = 0; k < num_params(P->cam_model[c]); k++) s.cam[16 * c + k] = v[L.cam[c] + k];
      if (L.bias[c] >= 0)
        for (int k = 0; k < 7; k++) s.bias[7 * c + k] = v[L.bias[c] + k];
    }
    for (int q = 0; q < NR; q++)
      if (L.rc[q] >= 0)
        for (int k = 0; k < 6; k++) s.rc[6 * q + k] = v[L.rc[q] + k];
    for (int i = 0; i < NI; i++)
      if (L.inst[i] >= 0)
        for (int k = 0; k < 6; k++) s.inst[6 * i + k] = v[L.inst[i] + k];
    for (int p = 0; p < NP; p++)
      if (L.pt[p] >= 0)
        for (int k = 0; k < 3; k++) s.pts[3 * p + k] = v[L.pt[p] + k];
  };

  // many points: the arrow form (same normal equations, points eliminated first); ORACLE_BUNDLE_ARROW=1 / 0 forces either (tests)
  const int nr_first_point = [&] {
    int first = n;
    for (int p = 0; p < NP; p++)
      if (L.pt[p] >= 0) { first = L.pt[p]; break; }
    return first;
  }();
  const char* env_arrow = getenv("ORACLE_BUNDLE_ARROW");
  const bool arrow = env_arrow ? atoi(env_arrow) != 0 : n > 4000;
  const int nr = arrow ? nr_first_point : -1, npt = arrow ? (n - nr_first_point) / 3 : 0;
  auto fresh = [&]() { return arrow ? Accum(n, nr) : Accum(n); };
  Accum A = fresh();
  double cost = evaluate(*P, *O, L, X, useful, &A, nullptr);
  Rp->initial_cost = cost;
  Rp->cost_history[0] = cost;
  std::vector<double> scale(n, 1.0), x, xn, step(n), Hs(arrow ? (size_t)nr * nr : (size_t)n * n);
  // arrow form: solve (D H D + LM) s = -D g by eliminating the points; returns false when a block is not positive definite
  auto arrow_solve = [&](double radius_) -> bool {
    std::vector<double> Ainv((size_t)npt * 9), bp((size_t)npt * 3);
    for (int i = 0; i < nr; i++)
      for (int j = 0; j < nr; j++) Hs[(size_t)i * nr + j] = A.R[(size_t)i * nr + j] * scale[i] * scale[j];
    for (int i = 0; i < nr; i++) {
      const double dd = std::min(std::max(Hs[(size_t)i * nr + i], 1e-6), 1e32);
      Hs[(size_t)i * nr + i] += dd / radius_;
      step[i] = -A.g[i] * scale[i];
    }
    for (int p = 0; p < npt; p++) {
      double B[9];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) B[3 * a + b] = A.Hpp[(size_t)p * 9 + 3 * a + b] * scale[nr + 3 * p + a] * scale[nr + 3 * p + b];
      for (int a = 0; a < 3; a++) B[4 * a] += std::min(std::max(B[4 * a], 1e-6), 1e32) / radius_;
      const double c00 = B[4] * B[8] - B[5] * B[7], c01 = B[5] * B[6] - B[3] * B[8], c02 = B[3] * B[7] - B[4] * B[6];
      const double det = B[0] * c00 + B[1] * c01 + B[2] * c02;
      if (!(det > 0) || !std::isfinite(det)) return false;
      const double id = 1.0 / det;
      double* I = &Ainv[(size_t)p * 9];
      I[0] = c00 * id; I[1] = (B[2] * B[7] - B[1] * B[8]) * id; I[2] = (B[1] * B[5] - B[2] * B[4]) * id;
      I[3] = c01 * id; I[4] = (B[0] * B[8] - B[2] * B[6]) * id; I[5] = (B[2] * B[3] - B[0] * B[5]) * id;
      I[6] = c02 * id; I[7] = (B[1] * B[6] - B[0] * B[7]) * id; I[8] = (B[0] * B[4] - B[1] * B[3]) * id;
      for (int a = 0; a < 3; a++) bp[(size_t)3 * p + a] = -A.g[nr + 3 * p + a] * scale[nr + 3 * p + a];
      // Schur update: Hs -= Ws A^-1 Ws^T, step_r -= Ws A^-1 b_p   (Ws = D_r W D_p)
      const std::vector<int>& wc = A.wcol[(size_t)p];
      const size_t m = wc.size();
      std::vector<double> T(3 * m);  // Ws A^-1
      for (size_t k = 0; k < m; k++) {
        double ws[3];
        for (int a = 0; a < 3; a++) ws[a] = A.wval[(size_t)p][3 * k + a] * scale[wc[k]] * scale[nr + 3 * p + a];
        for (int b = 0; b < 3; b++) T[3 * k + b] = ws[0] * I[b] + ws[1] * I[3 + b] + ws[2] * I[6 + b];
        step[wc[k]] -= T[3 * k] * bp[(size_t)3 * p] + T[3 * k + 1] * bp[(size_t)3 * p + 1] + T[3 * k + 2] * bp[(size_t)3 * p + 2];
      }
      for (size_t k = 0; k < m; k++)
        for (size_t k2 = 0; k2 < m; k2++) {
          double v = 0.0;
          for (int a = 0; a < 3; a++) v += T[3 * k + a] * A.wval[(size_t)p][3 * k2 + a] * scale[wc[k2]] * scale[nr + 3 * p + a];
          Hs[(size_t)wc[k] * nr + wc[k2]] -= v;
        }
    }
    std::vector<double> sr(step.begin(), step.begin() + nr);
    if (nr > 0 && !cholesky_solve(Hs, sr, nr)) return false;
    for (int i = 0; i < nr; i++) step[i] = sr[i];
    for (int p = 0; p < npt; p++) {
      double rhs[3] = {bp[(size_t)3 * p], bp[(size_t)3 * p + 1], bp[(size_t)3 * p + 2]};
      const std::vector<int>& wc = A.wcol[(size_t)p];
      for (size_t k = 0; k < wc.size(); k++)
        for (int a = 0; a < 3; a++) rhs[a] -= A.wval[(size_t)p][3 * k + a] * scale[wc[k]] * scale[nr + 3 * p + a] * step[wc[k]];
      const double* I = &Ainv[(size_t)p * 9];
      for (int a = 0; a < 3; a++) step[nr + 3 * p + a] = I[3 * a] * rhs[0] + I[3 * a + 1] * rhs[1] + I[3 * a + 2] * rhs[2];
    }
    return true;
  };
  auto arrow_Hx = [&](const std::vector<double>& v, std::vector<double>& out) {  // out = H v
    out.assign(n, 0.0);
    for (int i = 0; i < nr; i++) {
      double a = 0.0;
      for (int j = 0; j < nr; j++) a += A.R[(size_t)i * nr + j] * v[j];
      out[i] = a;
    }
    for (int p = 0; p < npt; p++) {
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) out[nr + 3 * p + a] += A.Hpp[(size_t)p * 9 + 3 * a + b] * v[nr + 3 * p + b];
      const std::vector<int>& wc = A.wcol[(size_t)p];
      for (size_t k = 0; k < wc.size(); k++)
        for (int a = 0; a < 3; a++) {
          out[wc[k]] += A.wval[(size_t)p][3 * k + a] * v[nr + 3 * p + a];
          out[nr + 3 * p + a] += A.wval[(size_t)p][3 * k + a] * v[wc[k]];
        }
    }
  };
  bool have_scale = false;
  double radius = O->initial_radius, decrease = 2.0;
  int iter = 0, n_invalid = 0;
  Rp->termination = 0;
  for (;;) {
    if (!have_scale) {
      for (int i = 0; i < n; i++) scale[i] = 1.0 / (1.0 + std::sqrt(A.diag(i)));
      have_scale = true;
    }
    double gmax = 0.0;
    for (int i = 0; i < n; i++) gmax = std::max(gmax, std::fabs(A.g[i]));
    if (iter >= O->max_iterations) { Rp->termination = 0; break; }
    if (gmax <= O->gradient_tolerance) { Rp->termination = 2; break; }
    if (radius < 1e-32) { Rp->termination = 4; break; }
    iter++;
    if (iter < 256) Rp->cost_history[iter] = cost;
    bool ok;
    if (arrow) {
      ok = arrow_solve(radius);
    } else {
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Hs[(size_t)i * n + j] = A.H[(size_t)i * n + j] * scale[i] * scale[j];
      for (int i = 0; i < n; i++) {
        const double dd = std::min(std::max(Hs[(size_t)i * n + i], 1e-6), 1e32);
        Hs[(size_t)i * n + i] += dd / radius;
        step[i] = -A.g[i] * scale[i];
      }
      ok = cholesky_solve(Hs, step, n);
    }
    double model_change = 0.0, step_sq = 0.0, x_sq = 0.0;
    if (ok) {
      for (int i = 0; i < n; i++) step[i] *= scale[i];
      if (arrow) {
        std::vector<double> hd;
        arrow_Hx(step, hd);
        for (int i = 0; i < n; i++) model_change -= step[i] * (A.g[i] + 0.5 * hd[i]);
      } else
        for (int i = 0; i < n; i++) {
          double hd = 0.0;
          for (int j = 0; j < n; j++) hd += A.H[(size_t)i * n + j] * step[j];
          model_change -= step[i] * (A.g[i] + 0.5 * hd);
        }
      flat(X, x);
      for (int i = 0; i < n; i++) {
        step_sq += step[i] * step[i];
        x_sq += x[i] * x[i];
      }
    }
    if (!ok || !(model_change > 0) || !std::isfinite(model_change)) {
      radius *= 0.5;
      if (++n_invalid >= 5) { Rp->termination = -1; break; }
      continue;
    }
    n_invalid = 0;
    xn = x;
    for (int i = 0; i < n; i++) xn[i] += step[i];
    State Xn = X;
    unflat(xn, Xn);
    const double cost_n = evaluate(*P, *O, L, Xn, useful, nullptr, nullptr);
    if (std::sqrt(step_sq) <= O->parameter_tolerance * (std::sqrt(x_sq) + O->parameter_tolerance)) { Rp->termination = 3; break; }
    const double cost_change = cost - cost_n;
    if (std::fabs(cost_change) <= O->function_tolerance * cost) { Rp->termination = 1; break; }
    const double rho = cost_change / model_change;
    if (rho > 1e-3) {
      X = Xn;
      const double t = 2.0 * rho - 1.0;
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
      decrease = 2.0;
      Rp->successful_steps++;
      A = fresh();
      cost = evaluate(*P, *O, L, X, useful, &A, nullptr);
    } else {
      radius /= decrease;
      decrease *= 2.0;
    }
    if (iter < 256) Rp->cost_history[iter] = cost;
  }
  Rp->iterations = iter;
  Rp->final_cost = cost;
  memcpy(P->cam_params, X.cam.data(), sizeof(double) * 16 * NC);
  if (P->bias) memcpy(P->bias, X.bias.data(), sizeof(double) * 7 * NC);
  memcpy(P->rig_camera_pose, X.rc.data(), sizeof(double) * 6 * NR);
  memcpy(P->rig_instance_pose, X.inst.data(), sizeof(double) * 6 * NI);
  if (NP) memcpy(P->points, X.pts.data(), sizeof(double) * 3 * NP);
  if (P->reproj_err) evaluate(*P, *O, L, X, useful, nullptr, P->reproj_err);
  return 0;
}

// one reprojection residual with its full Jacobian, for the derivative tests: J is nres x 31 row-major over
// [camera 16 | rig instance 6 | rig camera 6 | point 3]
extern "C" int oracle_bundle_reprojection(int model, const double* cam, const double* inst, const double* rcam, int use_rig_camera, const double* pt,
                                          const double* obs, double sigma, double* res, double* J) {
  typedef Jet<31> T;
  T c[16], i6[6], r6[6], p3[3], Xc[3], r[3];
  for (int k = 0; k < 16; k++) c[k] = T(cam[k], k);
  for (int k = 0; k < 6; k++) {
    i6[k] = T(inst[k], 16 + k);
    r6[k] = T(rcam[k], 22 + k);
  }
  for (int k = 0; k < 3; k++) p3[k] = T(pt[k], 28 + k);
  WorldToCameraCoordinatesRig<T>(i6, use_rig_camera ? r6 : nullptr, p3, Xc);
  int nres = 2;
  if (model == SPHERICAL) {
    nres = 3;
    const T nn = sqrt(Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2]);
    double b[3];
    bearing_of(obs, b);
    for (int a = 0; a < 3; a++) r[a] = (Xc[a] / nn - b[a]) * (1.0 / sigma);
  } else {
    T pr[2];
    project<T>(model, c, Xc, pr);
    r[0] = (pr[0] - obs[0]) * (1.0 / sigma);
    r[1] = (pr[1] - obs[1]) * (1.0 / sigma);
  }
  for (int e = 0; e < nres; e++) {
    res[e] = r[e].v;
    for (int k = 0; k < 31; k++) J[31 * e + k] = r[e].d[k];
  }
  return nres;
}
