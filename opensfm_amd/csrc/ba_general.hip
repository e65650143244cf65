// ba_general.hip -- the general bundle adjustment: everything sfm::BAHelpers::Bundle (opensfm/src/sfm/src/ba_helpers.cc:581-763) puts
// into a bundle::BundleAdjuster and bundle::BundleAdjuster::Run (opensfm/src/bundle/src/bundle_adjuster.cc:595-1121) then solves:
//
//   parameter blocks  cameras (every projection type, all native parameters: camera_instances.h), GPS biases (bias.h), rig cameras,
//                     rig instances, points -- each constant or free as a whole (bundle_adjuster.cc:598-670)
//   residuals         reprojection errors 2-D / 3-D for spherical cameras through the rig (projection_errors.h:31-57,208-246,
//                     error_utils.h:68-85) under the shared robust loss; camera priors with logarithmic focal / aspect ratio
//                     (bundle_adjuster.cc:568-593, prior_error.h); the dual camera's transition barrier (parameters_errors.h:21-38);
//                     rig camera pose priors; rig instance position priors through the camera's bias (bias.h:33-53,
//                     bundle_adjuster.cc:745-778); point priors = ground control points (bundle_adjuster.cc:688-707,
//                     ba_helpers.cc:349-406); absolute up vectors with Cauchy(1) (absolute_motion_errors.h:12-39)
//   minimiser         Ceres' trust-region Levenberg-Marquardt with SPARSE_SCHUR's exact linear solve: points are eliminated, the
//                     reduced system over everything else is formed DENSE and factorised (Cholesky).
//
// ba.hip solves the configuration the benchmark is quoted on (perspective / fisheye cameras with [k1 k2 focal], identity rigs, no
// biases, no control points) without ever forming the reduced system, at 5 000 shots / 5 M observations.  This solver trades that
// scalability for generality: the reduced system is n_r x n_r doubles in HBM (n_r = 6 x free instances + free camera / rig camera /
// bias parameters; 30 000 unknowns = 7.2 GB of the 288), assembled with fp64 atomics, factorised by rocSOLVER.  Jacobians are
// analytic: rotations by their derivative matrices, cameras by project_full (ba_math.h); the small prior families by dual numbers.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

#include <rocsolver/rocsolver.h>

#include "ba_math.h"
#include "osfm_internal.h"

using namespace osfm_ba;

namespace {

constexpr int kMaxK = 16;                     // native camera parameters, at most
constexpr int kRowJ = 3 + 9 + 18 + 18 + 48;   // per observation: residual 3 | d point 3x3 | d instance 3x6 | d rig camera 3x6 | d camera 3x16

struct GDev {
  int NC, NR, NI, S, P;
  long M;
  int nred;
  // parameters: current and candidate
  double *cam, *bias, *rc, *inst, *pts;       // NC x 16, NC x 7, NR x 6, NI x 6, P x 3
  double *cam_n, *bias_n, *rc_n, *inst_n, *pts_n;
  const int *cam_model;
  const double *cam_prior, *cam_sigma;
  const double *rc_prior, *rc_sigma;          // or null
  const double *gps, *gps_sigma;              // NI x 3 each, or null
  const int *inst_bias_cam;
  const double *up, *up_sigma;                // S x 3, S, or null
  const double *pan, *pan_sigma, *tilt, *tilt_sigma, *roll, *roll_sigma;  // S each, or null
  const double *pt_prior, *pt_prior_sigma;    // P x 3 each, or null
  const unsigned char *pt_prior_alt;
  // reduced index of the first parameter of every block, -1 when the block is constant
  const int *cam_idx, *bias_idx, *rc_idx, *inst_idx;
  const unsigned char *rc_useful;             // rig camera enters the reprojection (not constant-identity)
  const unsigned char *pt_fixed;
  const int *shot_inst, *shot_rc, *shot_cam;
  // observations, point-major
  const int *o_shot, *o_point;
  const double *o_x, *o_y, *o_sigma;
  const unsigned char *o_kind;                // null, or per row: 0 = reprojection, 1 = depth prior on z, 2 = radial depth prior (o_x = depth, o_sigma = its sd)
  const long *pt_off;                         // P + 1
  double *J;                                  // M x kRowJ: corrected residual and Jacobian rows
  // points
  double *Hpp, *gpt, *Hhat, *sc_pt, *D_pt;    // 6, 3, 6, 3, 3 per point
  double *dpt;                                // 3 per point: step
  // reduced system
  double *H;                                  // nred x nred (column-major = row-major, symmetric; both triangles filled)
  double *Sm;                                 // nred x nred: scaled + damped Schur complement handed to the factorisation
  double *g, *diag, *sc, *D, *rhs, *delta;    // nred each
  double *scal;                               // device scalars / partial sums
};

__device__ __forceinline__ void atomic_add(double *p, double v) { atomicAdd(p, v); }

__device__ __forceinline__ void loss_eval(int loss, double a, double s, double &rho, double &rho1) {
  const double b = a * a;
  switch (loss) {
    case OSFM_LOSS_SOFTLONE: {
      const double sum = 1.0 + s / b, tmp = sqrt(sum);
      rho = 2.0 * b * (tmp - 1.0);
      rho1 = 1.0 / tmp;
    } break;
    case OSFM_LOSS_HUBER:
      if (s > b) {
        const double r = sqrt(s);
        rho = 2.0 * a * r - b;
        rho1 = a / r;
      } else {
        rho = s;
        rho1 = 1.0;
      }
      break;
    case OSFM_LOSS_CAUCHY: {
      const double sum = 1.0 + s / b;
      rho = b * log(sum);
      rho1 = 1.0 / sum;
    } break;
    default:
      rho = s;
      rho1 = 1.0;
  }
}

// x_local = R(-r) (x - t) for a CAM_TO_WORLD pose [r | t] (error_utils.h:52-66 WorldToLocal): value, and with JAC the 3 x 3 rotation
// and the three derivative matrices needed by the chain rule
struct PoseMap {
  double R[9], dR[27];
};

// residual (res, nres = 2 or 3) and Jacobian rows of one reprojection: Jp 3x3, Ji 3x6, Jr 3x6, Jk 3x16 (rows beyond nres zero)
template <bool JAC>
__device__ void reproj_eval(const GDev &d, const double *cam, const double *rcp, const double *instp, const double *pts, long o, double *res,
                            int &nres, double *Jp, double *Ji, double *Jr, double *Jk) {
  const int s = d.o_shot[o], p = d.o_point[o];
  const int i = d.shot_inst[s], rc = d.shot_rc[s], c = d.shot_cam[s];
  const int model = d.cam_model[c];
  const double *X = pts + 3 * (long)p, *pi = instp + 6 * (long)i, *pr = rcp + 6 * (long)rc;
  const bool use_rc = d.rc_useful[rc] != 0;
  PoseMap mi, mr;
  rot_and_derivs(pi, mi.R, mi.dR);
  const double q[3] = {X[0] - pi[3], X[1] - pi[4], X[2] - pi[5]};
  double Xi[3], Xc[3], w[3];
  for (int a = 0; a < 3; a++) Xi[a] = mi.R[3 * a] * q[0] + mi.R[3 * a + 1] * q[1] + mi.R[3 * a + 2] * q[2];
  if (use_rc) {
    rot_and_derivs(pr, mr.R, mr.dR);
    for (int a = 0; a < 3; a++) w[a] = Xi[a] - pr[3 + a];
    for (int a = 0; a < 3; a++) Xc[a] = mr.R[3 * a] * w[0] + mr.R[3 * a + 1] * w[1] + mr.R[3 * a + 2] * w[2];
  } else {
    for (int a = 0; a < 3; a++) Xc[a] = Xi[a];
  }
  // d residual / d Xc (nres x 3) and / d camera parameters
  double A[9], K[48];
  const double is = 1.0 / d.o_sigma[o];
  const int kind = d.o_kind ? d.o_kind[o] : 0;
  if (kind != 0) {  // RelativeDepthError (relative_depth_error.h:21-39): (depth in the camera - depth) / sd, on [instance | rig camera | point]
    nres = 1;
    const double n = sqrt(Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2]);
    res[0] = is * ((kind == 2 ? n : Xc[2]) - d.o_x[o]);
    res[1] = res[2] = 0.0;
    if (JAC) {
      for (int e = 0; e < 9; e++) A[e] = 0.0;
      if (kind == 2) {
        for (int e = 0; e < 3; e++) A[e] = is * Xc[e] / n;
      } else {
        A[2] = is;
      }
      for (int e = 0; e < 48; e++) K[e] = 0.0;
    }
  } else if (model == OSFM_CAMERA_SPHERICAL) {  // ReprojectionError3D: unit bearing minus the observed bearing
    nres = 3;
    const double n2 = Xc[0] * Xc[0] + Xc[1] * Xc[1] + Xc[2] * Xc[2], n = sqrt(n2);
    const double lon = d.o_x[o] * 2 * M_PI, lat = -d.o_y[o] * 2 * M_PI;
    const double b[3] = {cos(lat) * sin(lon), -sin(lat), cos(lat) * cos(lon)};
    for (int a = 0; a < 3; a++) res[a] = is * (Xc[a] / n - b[a]);
    if (JAC) {
      for (int a = 0; a < 3; a++)
        for (int e = 0; e < 3; e++) A[3 * a + e] = is * ((a == e ? 1.0 : 0.0) - Xc[a] * Xc[e] / n2) / n;
      for (int e = 0; e < 48; e++) K[e] = 0.0;
    }
  } else {
    nres = 2;
    double out[2], JX[6], JK[32];
    project_full<JAC>(model, cam + 16 * (long)c, Xc, out, JX, JK);
    res[0] = is * (out[0] - d.o_x[o]);
    res[1] = is * (out[1] - d.o_y[o]);
    res[2] = 0.0;
    if (JAC) {
      for (int e = 0; e < 6; e++) A[e] = is * JX[e];
      for (int e = 6; e < 9; e++) A[e] = 0.0;
      for (int e = 0; e < 32; e++) K[e] = is * JK[e];
      for (int e = 32; e < 48; e++) K[e] = 0.0;
    }
  }
  if (!JAC) return;
  for (int e = 0; e < 48; e++) Jk[e] = K[e];
  // B = A * Rrc (or A): d residual / d Xi
  double B[9];
  if (use_rc) {
    for (int a = 0; a < 3; a++)
      for (int e = 0; e < 3; e++) B[3 * a + e] = A[3 * a] * mr.R[e] + A[3 * a + 1] * mr.R[3 + e] + A[3 * a + 2] * mr.R[6 + e];
    for (int a = 0; a < 3; a++) {
      for (int k = 0; k < 3; k++) {  // rotation of the rig camera: dR_k (Xi - t)
        double v[3];
        for (int e = 0; e < 3; e++) v[e] = mr.dR[9 * k + 3 * e] * w[0] + mr.dR[9 * k + 3 * e + 1] * w[1] + mr.dR[9 * k + 3 * e + 2] * w[2];
        // the pose stores r, the map rotates by a = -r: d / d r_k = -d / d a_k
        Jr[6 * a + k] = -(A[3 * a] * v[0] + A[3 * a + 1] * v[1] + A[3 * a + 2] * v[2]);
      }
      for (int e = 0; e < 3; e++) Jr[6 * a + 3 + e] = -B[3 * a + e];
    }
  } else {
    for (int e = 0; e < 9; e++) B[e] = A[e];
    for (int e = 0; e < 18; e++) Jr[e] = 0.0;
  }
  for (int a = 0; a < 3; a++) {
    for (int e = 0; e < 3; e++) {
      const double v = B[3 * a] * mi.R[e] + B[3 * a + 1] * mi.R[3 + e] + B[3 * a + 2] * mi.R[6 + e];
      Jp[3 * a + e] = v;
      Ji[6 * a + 3 + e] = -v;
    }
    for (int k = 0; k < 3; k++) {
      double v[3];
      for (int e = 0; e < 3; e++) v[e] = mi.dR[9 * k + 3 * e] * q[0] + mi.dR[9 * k + 3 * e + 1] * q[1] + mi.dR[9 * k + 3 * e + 2] * q[2];
      Ji[6 * a + k] = -(B[3 * a] * v[0] + B[3 * a + 1] * v[1] + B[3 * a + 2] * v[2]);
    }
  }
}

// block-wide sum of one value per thread into out (atomic per block)
__device__ __forceinline__ void block_accumulate(double v, double *out) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if ((threadIdx.x & 63) == 0 && v != 0.0) atomic_add(out, v);
}

// reprojection residuals: cost (scal[0] += 1/2 rho) and, with JAC, the corrected rows
template <bool JAC>
__global__ void __launch_bounds__(TPB) g_eval_kernel(GDev d, const double *cam, const double *rcp, const double *instp, const double *pts, int loss,
                                                      double a) {
  const long o = (long)blockIdx.x * TPB + threadIdx.x;
  double cost = 0.0;
  if (o < d.M) {
    double res[3], Jp[9], Ji[18], Jr[18], Jk[48];
    int nres;
    reproj_eval<JAC>(d, cam, rcp, instp, pts, o, res, nres, Jp, Ji, Jr, Jk);
    const double sq = res[0] * res[0] + res[1] * res[1] + res[2] * res[2];
    double rho, rho1;
    loss_eval(loss, a, sq, rho, rho1);
    cost = 0.5 * rho;
    if (JAC) {
      const double wt = sqrt(rho1);  // Triggs corrector for losses with rho'' <= 0
      double *row = d.J + o * kRowJ;
      for (int e = 0; e < 3; e++) row[e] = wt * res[e];
      for (int e = 0; e < 9; e++) row[3 + e] = wt * Jp[e];
      for (int e = 0; e < 18; e++) row[12 + e] = wt * Ji[e];
      for (int e = 0; e < 18; e++) row[30 + e] = wt * Jr[e];
      for (int e = 0; e < 48; e++) row[48 + e] = wt * Jk[e];
    }
  }
  block_accumulate(cost, d.scal);
}

// plain residuals with sigma = 1 (ComputeReprojectionErrors, bundle_adjuster.cc:1196-1208)
__global__ void g_reproj_kernel(GDev d, double *out) {
  const long o = (long)blockIdx.x * TPB + threadIdx.x;
  if (o >= d.M) return;
  double res[3], dummy[1];
  int nres;
  reproj_eval<false>(d, d.cam, d.rc, d.inst, d.pts, o, res, nres, dummy, dummy, dummy, dummy);
  const double sg = (d.o_kind && d.o_kind[o]) ? 0.0 : d.o_sigma[o];  // depth-prior rows are not reprojection errors
  for (int e = 0; e < 3; e++) out[3 * o + e] = res[e] * sg;
}

// ---- dual numbers for the small prior families (what the reference's autodiff does) ----
template <int N>
struct Jet {
  double v, d[N];
};
template <int N>
__device__ __forceinline__ Jet<N> jc(double c) {
  Jet<N> r;
  r.v = c;
  for (int i = 0; i < N; i++) r.d[i] = 0.0;
  return r;
}
template <int N>
__device__ __forceinline__ Jet<N> jv(double c, int k) {
  Jet<N> r = jc<N>(c);
  r.d[k] = 1.0;
  return r;
}
template <int N>
__device__ __forceinline__ Jet<N> operator+(Jet<N> a, const Jet<N> &b) {
  a.v += b.v;
  for (int i = 0; i < N; i++) a.d[i] += b.d[i];
  return a;
}
template <int N>
__device__ __forceinline__ Jet<N> operator-(Jet<N> a, const Jet<N> &b) {
  a.v -= b.v;
  for (int i = 0; i < N; i++) a.d[i] -= b.d[i];
  return a;
}
template <int N>
__device__ __forceinline__ Jet<N> operator*(const Jet<N> &a, const Jet<N> &b) {
  Jet<N> r;
  r.v = a.v * b.v;
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
template <int N>
__device__ __forceinline__ Jet<N> operator*(Jet<N> a, double c) {
  a.v *= c;
  for (int i = 0; i < N; i++) a.d[i] *= c;
  return a;
}
template <int N>
__device__ __forceinline__ Jet<N> operator/(const Jet<N> &a, const Jet<N> &b) {
  Jet<N> r;
  const double ib = 1.0 / b.v;
  r.v = a.v * ib;
  for (int i = 0; i < N; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
template <int N>
__device__ __forceinline__ Jet<N> jsqrt(const Jet<N> &a) {
  Jet<N> r;
  r.v = sqrt(a.v);
  const double h = 0.5 / r.v;
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] * h;
  return r;
}
template <int N>
__device__ __forceinline__ Jet<N> jsin(const Jet<N> &a) {
  Jet<N> r;
  r.v = sin(a.v);
  const double c = cos(a.v);
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] * c;
  return r;
}
template <int N>
__device__ __forceinline__ Jet<N> jcos(const Jet<N> &a) {
  Jet<N> r;
  r.v = cos(a.v);
  const double s = -sin(a.v);
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] * s;
  return r;
}
template <int N>
__device__ __forceinline__ Jet<N> jatan2(const Jet<N> &y, const Jet<N> &x) {
  Jet<N> r;
  r.v = atan2(y.v, x.v);
  const double den = 1.0 / (x.v * x.v + y.v * y.v);
  for (int i = 0; i < N; i++) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * den;
  return r;
}
template <int N>
__device__ __forceinline__ Jet<N> jasin(const Jet<N> &a) {
  Jet<N> r;
  r.v = asin(a.v);
  const double k = 1.0 / sqrt(1.0 - a.v * a.v);
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] * k;
  return r;
}
// DiffBetweenAngles (error_utils.h:88-97)
template <int N>
__device__ __forceinline__ Jet<N> jdiff_angles(const Jet<N> &a, double b) {
  Jet<N> dd = a;
  dd.v -= b;
  if (dd.v > M_PI) dd.v -= 2 * M_PI;
  else if (dd.v < -M_PI) dd.v += 2 * M_PI;
  return dd;
}
template <int N>
__device__ __forceinline__ Jet<N> jlog(const Jet<N> &a) {
  Jet<N> r;
  r.v = log(a.v);
  const double ia = 1.0 / a.v;
  for (int i = 0; i < N; i++) r.d[i] = a.d[i] * ia;
  return r;
}
// ceres::AngleAxisRotatePoint
template <int N>
__device__ void j_rotate(const Jet<N> *aa, const Jet<N> *pt, Jet<N> *out) {
  const Jet<N> th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2.v > kEps) {
    const Jet<N> th = jsqrt(th2), ct = jcos(th), st = jsin(th), ti = jc<N>(1.0) / th;
    const Jet<N> w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
    const Jet<N> wxp[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2], w[0] * pt[1] - w[1] * pt[0]};
    const Jet<N> tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (jc<N>(1.0) - ct);
    for (int i = 0; i < 3; i++) out[i] = pt[i] * ct + wxp[i] * st + w[i] * tmp;
  } else {
    const Jet<N> wxp[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2], aa[0] * pt[1] - aa[1] * pt[0]};
    for (int i = 0; i < 3; i++) out[i] = pt[i] + wxp[i];
  }
}
// ceres::AngleAxisToQuaternion / QuaternionProduct / QuaternionToAngleAxis: MultRotations (error_utils.h:14-24)
template <int N>
__device__ void j_aa_to_quat(const Jet<N> *aa, Jet<N> *q) {
  const Jet<N> th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2.v > 0.0) {
    const Jet<N> th = jsqrt(th2), half = th * 0.5, k = jsin(half) / th;
    q[0] = jcos(half);
    for (int i = 0; i < 3; i++) q[1 + i] = aa[i] * k;
  } else {
    q[0] = jc<N>(1.0);
    for (int i = 0; i < 3; i++) q[1 + i] = aa[i] * 0.5;
  }
}
template <int N>
__device__ void j_quat_to_aa(const Jet<N> *q, Jet<N> *aa) {
  const Jet<N> s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2.v > 0.0) {
    const Jet<N> s = jsqrt(s2);
    const Jet<N> two_theta = (q[0].v < 0.0 ? jatan2(jc<N>(0.0) - s, jc<N>(0.0) - q[0]) : jatan2(s, q[0])) * 2.0;
    const Jet<N> k = two_theta / s;
    for (int i = 0; i < 3; i++) aa[i] = q[1 + i] * k;
  } else {
    for (int i = 0; i < 3; i++) aa[i] = q[1 + i] * 2.0;
  }
}
template <int N>
__device__ void j_mult_rotations(const Jet<N> *r1, const Jet<N> *r2, Jet<N> *out) {
  Jet<N> a[4], b[4], c[4];
  j_aa_to_quat(r1, a);
  j_aa_to_quat(r2, b);
  c[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  c[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  c[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  c[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  j_quat_to_aa(c, out);
}

// adds the normal-equation contribution of one residual block: nres residuals r with Jacobian rows over up to two parameter blocks
// (reduced start index ia / ib, sizes na / nb; an index < 0 = constant block)
template <int N>
__device__ void add_block(const GDev &d, const Jet<N> *r, int nres, double wt, int ia, int na, int ib, int nb) {
  for (int x = 0; x < na + nb; x++) {
    const int gx = x < na ? (ia < 0 ? -1 : ia + x) : (ib < 0 ? -1 : ib + x - na);
    if (gx < 0) continue;
    double gsum = 0.0;
    for (int e = 0; e < nres; e++) gsum += wt * r[e].d[x] * wt * r[e].v;
    atomic_add(d.g + gx, gsum);
    for (int y = 0; y < na + nb; y++) {
      const int gy = y < na ? (ia < 0 ? -1 : ia + y) : (ib < 0 ? -1 : ib + y - na);
      if (gy < 0) continue;
      double h = 0.0;
      for (int e = 0; e < nres; e++) h += wt * r[e].d[x] * wt * r[e].d[y];
      atomic_add(d.H + (long)gx * d.nred + gy, h);
    }
  }
}

// The prior families, one thread per block.  kind 0: camera prior (+ the dual barrier), 1: rig camera pose prior, 2: rig instance
// position prior through the bias, 3: absolute up vector.  mode 0: cost only (scal[0]); 1: cost + normal equations; 2: model change
// scal[1] -= u (r + u / 2) with u = J delta (delta = the reduced step)
__global__ void g_prior_kernel(GDev d, const double *cam, const double *bias, const double *rcp, const double *instp, int mode) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0, mc = 0.0;
  auto finish = [&](auto *r, int nres, double wt, int ia, int na, int ib, int nb) {
    using J = std::remove_pointer_t<decltype(r)>;
    constexpr int N = sizeof(J::d) / sizeof(double);
    if (mode == 1) add_block<N>(d, r, nres, wt, ia, na, ib, nb);
    if (mode == 2)
      for (int e = 0; e < nres; e++) {
        double u = 0.0;
        for (int x = 0; x < na + nb; x++) {
          const int gx = x < na ? (ia < 0 ? -1 : ia + x) : (ib < 0 ? -1 : ib + x - na);
          if (gx >= 0) u += wt * r[e].d[x] * d.delta[gx];
        }
        mc -= u * (wt * r[e].v + 0.5 * u);
      }
  };
  if (t < d.NC) {  // ---- camera prior: (p - prior) / sigma, log(p / prior) / sigma for focal and aspect ratio ----
    const int c = t, model = d.cam_model[c], nk = model_num_params(model);
    int proj, kind, nd, na;
    model_layout(model, proj, kind, nd, na);
    const int ifocal = (proj == 2 ? 1 : 0) + nd;
    Jet<kMaxK> r[kMaxK + 1];
    for (int k = 0; k < nk; k++) {
      const Jet<kMaxK> v = jv<kMaxK>(cam[16 * c + k], k);
      const double scale = 1.0 / fmax(d.cam_sigma[16 * c + k], 2.220446049250313e-16);
      const bool logarithmic = (k == ifocal) || (na == 4 && k == ifocal + 1);
      r[k] = (logarithmic ? jlog(v / jc<kMaxK>(d.cam_prior[16 * c + k])) : v - jc<kMaxK>(d.cam_prior[16 * c + k])) * scale;
    }
    int nres = nk;
    if (model == OSFM_CAMERA_DUAL) {  // ParameterBarrier(0, 1) on the transition
      const Jet<kMaxK> v = jv<kMaxK>(cam[16 * c], 0);
      r[nres++] = jlog(v + jc<kMaxK>(1e-10)) + jlog(jc<kMaxK>(1.0 + 1e-10) - v) + jc<kMaxK>(2.0 * log(0.5));
    }
    for (int e = 0; e < nres; e++) cost += 0.5 * r[e].v * r[e].v;
    finish(r, nres, 1.0, d.cam_idx[c], nk, -1, 0);
  } else if (t < d.NC + d.NR) {  // ---- rig camera pose prior ----
    const int q = t - d.NC;
    if (d.rc_prior && d.rc_sigma) {
      Jet<6> r[6];
      for (int k = 0; k < 6; k++)
        r[k] = (jv<6>(rcp[6 * q + k], k) - jc<6>(d.rc_prior[6 * q + k])) * (1.0 / fmax(d.rc_sigma[6 * q + k], 2.220446049250313e-16));
      for (int e = 0; e < 6; e++) cost += 0.5 * r[e].v * r[e].v;
      finish(r, 6, 1.0, d.rc_idx[q], 6, -1, 0);
    }
  } else if (t < d.NC + d.NR + d.NI) {  // ---- rig instance position prior through the bias: t - (s R(b) gps + t_b) ----
    const int i = t - d.NC - d.NR;
    if (d.gps && d.gps_sigma && d.gps_sigma[3 * i] > 0) {
      const int bc = d.inst_bias_cam[i];
      const double *b = bias + 7 * (long)bc;
      constexpr int N = 13;  // instance 6 | bias 7
      Jet<N> rot[3], tb[3], g[3], rg[3], r[3];
      for (int k = 0; k < 3; k++) {
        rot[k] = jv<N>(b[k], 6 + k);
        tb[k] = jv<N>(b[3 + k], 9 + k);
        g[k] = jc<N>(d.gps[3 * i + k]);
      }
      const Jet<N> sc = jv<N>(b[6], 12);
      j_rotate(rot, g, rg);
      for (int k = 0; k < 3; k++)
        r[k] = (jv<N>(instp[6 * i + 3 + k], 3 + k) - (sc * rg[k] + tb[k])) * (1.0 / fmax(d.gps_sigma[3 * i + k], 2.220446049250313e-16));
      for (int e = 0; e < 3; e++) cost += 0.5 * r[e].v * r[e].v;
      finish(r, 3, 1.0, d.inst_idx[i], 6, d.bias_idx[bc], 7);
    }
  } else if (t < d.NC + d.NR + d.NI + d.S) {  // ---- absolute up vector, CauchyLoss(1) ----
    const int s = t - d.NC - d.NR - d.NI;
    if (d.up && d.up_sigma && d.up_sigma[s] > 0) {
      const int i = d.shot_inst[s], q = d.shot_rc[s];
      constexpr int N = 12;  // instance 6 | rig camera 6
      Jet<N> ri[3], rr[3], R[3], acc[3], z[3], r[3];
      const double *u0 = d.up + 3 * (long)s;
      const double nrm = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
      for (int k = 0; k < 3; k++) {
        ri[k] = jv<N>(instp[6 * i + k], k);
        rr[k] = jv<N>(rcp[6 * q + k], 6 + k);
        acc[k] = jc<N>(u0[k] / nrm);
      }
      j_mult_rotations(ri, rr, R);
      j_rotate(R, acc, z);
      const double is = 1.0 / d.up_sigma[s];
      for (int k = 0; k < 3; k++) r[k] = (z[k] - jc<N>(k == 2 ? 1.0 : 0.0)) * is;
      const double sq = r[0].v * r[0].v + r[1].v * r[1].v + r[2].v * r[2].v;
      double rho, rho1;
      loss_eval(OSFM_LOSS_CAUCHY, 1.0, sq, rho, rho1);
      cost += 0.5 * rho;
      finish(r, 3, sqrt(rho1), d.inst_idx[i], 6, d.rc_idx[q], 6);
    }
  }
  else if (t < d.NC + d.NR + d.NI + 4 * d.S) {  // ---- absolute pan / tilt / roll of a shot, CauchyLoss(1) each ----
    const int u = t - d.NC - d.NR - d.NI - d.S, which = u / d.S, s = u % d.S;
    const double *ang = which == 0 ? d.pan : (which == 1 ? d.tilt : d.roll), *sg = which == 0 ? d.pan_sigma : (which == 1 ? d.tilt_sigma : d.roll_sigma);
    if (ang && sg && sg[s] > 0) {
      const int i = d.shot_inst[s], q = d.shot_rc[s];
      constexpr int N = 12;
      Jet<N> ri[3], rr[3], R[3], r[1];
      for (int k = 0; k < 3; k++) {
        ri[k] = jv<N>(instp[6 * i + k], k);
        rr[k] = jv<N>(rcp[6 * q + k], 6 + k);
      }
      j_mult_rotations(ri, rr, R);
      const Jet<N> ex[3] = {jc<N>(1.0), jc<N>(0.0), jc<N>(0.0)}, ez[3] = {jc<N>(0.0), jc<N>(0.0), jc<N>(1.0)};
      Jet<N> zw[3];
      j_rotate(R, ez, zw);
      const double is = 1.0 / sg[s];
      bool zero = false;
      if (which == 0) {  // PanAngleError
        if (fabs(zw[0].v) < 1e-8 && fabs(zw[1].v) < 1e-8) zero = true;
        else r[0] = jdiff_angles(jatan2(zw[0], zw[1]), ang[s]) * is;
      } else if (which == 1) {  // TiltAngleError
        const Jet<N> l = jsqrt(zw[0] * zw[0] + zw[1] * zw[1]);
        r[0] = jdiff_angles(jc<N>(0.0) - jatan2(zw[2], l), ang[s]) * is;
      } else {  // RollAngleError
        Jet<N> xw[3];
        j_rotate(R, ex, xw);
        Jet<N> a0 = zw[1], a1 = jc<N>(0.0) - zw[0];
        const Jet<N> la = jsqrt(a0 * a0 + a1 * a1);
        if (la.v < 1e-5) {
          zero = true;
        } else {
          a0 = a0 / la;
          a1 = a1 / la;
          // b = Rt_ex x a with a = (a0, a1, 0)
          const Jet<N> b0 = jc<N>(0.0) - xw[2] * a1, b1 = xw[2] * a0, b2 = xw[0] * a1 - xw[1] * a0;
          const Jet<N> sin_roll = zw[0] * b0 + zw[1] * b1 + zw[2] * b2;
          if (sin_roll.v <= -(1.0 - 1e-5)) zero = true;
          else r[0] = jdiff_angles(jasin(sin_roll), ang[s]) * is;
        }
      }
      if (!zero) {
        double rho, rho1;
        loss_eval(OSFM_LOSS_CAUCHY, 1.0, r[0].v * r[0].v, rho, rho1);
        cost += 0.5 * rho;
        finish(r, 1, sqrt(rho1), d.inst_idx[i], 6, d.rc_idx[q], 6);
      }
    }
  }
  block_accumulate(cost, d.scal);
  if (mode == 2) block_accumulate(mc, d.scal + 1);
}

// ---- points: Hpp, g_p from the reprojection rows and the point priors (ground control points) ----
__global__ void g_point_kernel(GDev d, const double *pts) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.P) return;
  double H[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, cost = 0.0;
  for (long o = d.pt_off[p]; o < d.pt_off[p + 1]; o++) {
    const double *row = d.J + o * kRowJ;
    for (int e = 0; e < 3; e++) {
      const double *jp = row + 3 + 3 * e;
      H[0] += jp[0] * jp[0]; H[1] += jp[0] * jp[1]; H[2] += jp[0] * jp[2];
      H[3] += jp[1] * jp[1]; H[4] += jp[1] * jp[2]; H[5] += jp[2] * jp[2];
      for (int a = 0; a < 3; a++) g[a] += jp[a] * row[e];
    }
  }
  if (d.pt_prior && d.pt_prior_sigma && d.pt_prior_sigma[3 * (long)p] > 0) {
    const int n = d.pt_prior_alt[p] ? 3 : 2;
    const int dg[3] = {0, 3, 5};
    for (int a = 0; a < n; a++) {
      const double sc = 1.0 / fmax(d.pt_prior_sigma[3 * (long)p + a], 2.220446049250313e-16);
      const double r = sc * (pts[3 * (long)p + a] - d.pt_prior[3 * (long)p + a]);
      H[dg[a]] += sc * sc;
      g[a] += sc * r;
      cost += 0.5 * r * r;
    }
  }
  for (int e = 0; e < 6; e++) d.Hpp[6 * (long)p + e] = H[e];
  for (int e = 0; e < 3; e++) d.gpt[3 * (long)p + e] = g[e];
  (void)cost;
}

// cost of the point priors (the candidate evaluation has no rows)
__global__ void g_point_prior_cost_kernel(GDev d, const double *pts) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double cost = 0.0;
  if (p < d.P && d.pt_prior && d.pt_prior_sigma && d.pt_prior_sigma[3 * (long)p] > 0) {
    const int n = d.pt_prior_alt[p] ? 3 : 2;
    for (int a = 0; a < n; a++) {
      const double r = (pts[3 * (long)p + a] - d.pt_prior[3 * (long)p + a]) / fmax(d.pt_prior_sigma[3 * (long)p + a], 2.220446049250313e-16);
      cost += 0.5 * r * r;
    }
  }
  block_accumulate(cost, d.scal);
}

// reduced columns of observation o: 6 (instance) + 6 (rig camera) + 16 (camera); cols[x] = reduced index or -1
__device__ __forceinline__ void obs_columns(const GDev &d, long o, int *cols) {
  const int s = d.o_shot[o];
  const int ii = d.inst_idx[d.shot_inst[s]], ri = d.rc_useful[d.shot_rc[s]] ? d.rc_idx[d.shot_rc[s]] : -1, c = d.shot_cam[s];
  const int ci = d.cam_idx[c], nk = model_num_params(d.cam_model[c]);
  for (int x = 0; x < 6; x++) cols[x] = ii < 0 ? -1 : ii + x;
  for (int x = 0; x < 6; x++) cols[6 + x] = ri < 0 ? -1 : ri + x;
  for (int x = 0; x < 16; x++) cols[12 + x] = (ci < 0 || x >= nk) ? -1 : ci + x;
}
// entry (e, x) of the reduced Jacobian row block of observation o (3 x 28)
__device__ __forceinline__ double jred(const double *row, int e, int x) {
  return x < 6 ? row[12 + 6 * e + x] : (x < 12 ? row[30 + 6 * e + (x - 6)] : row[48 + 16 * e + (x - 12)]);
}

// direct terms of the reduced normal equations: H += Jr^T Jr, g += Jr^T r, one thread per observation
__global__ void g_reduced_kernel(GDev d) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= d.M) return;
  const double *row = d.J + o * kRowJ;
  int cols[28];
  obs_columns(d, o, cols);
  for (int x = 0; x < 28; x++) {
    if (cols[x] < 0) continue;
    double g = 0.0;
    for (int e = 0; e < 3; e++) g += jred(row, e, x) * row[e];
    atomic_add(d.g + cols[x], g);
    for (int y = 0; y < 28; y++) {
      if (cols[y] < 0) continue;
      double h = 0.0;
      for (int e = 0; e < 3; e++) h += jred(row, e, x) * jred(row, e, y);
      if (h != 0.0) atomic_add(d.H + (long)cols[x] * d.nred + cols[y], h);
    }
  }
}

__global__ void g_diag_kernel(GDev d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.nred) d.diag[i] = d.H[(long)i * d.nred + i];
}
// Jacobi scaling (once) and the LM diagonal (levenberg_marquardt_strategy.cc): clamp(diag(J^T J) of the scaled Jacobian)
__global__ void g_scale_kernel(GDev d, int init) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int dg[3] = {0, 3, 5};
  if (i < d.nred) {
    if (init) d.sc[i] = 1.0 / (1.0 + sqrt(d.diag[i]));
    d.D[i] = fmin(fmax(d.diag[i] * d.sc[i] * d.sc[i], 1e-6), 1e32);
  }
  if (i < 3L * d.P) {
    const long p = i / 3;
    const int j = (int)(i - 3 * p);
    const bool fixed = d.pt_fixed && d.pt_fixed[p];
    if (init) d.sc_pt[i] = fixed ? 0.0 : 1.0 / (1.0 + sqrt(d.Hpp[6 * p + dg[j]]));
    d.D_pt[i] = fmin(fmax(d.Hpp[6 * p + dg[j]] * d.sc_pt[i] * d.sc_pt[i], 1e-6), 1e32);
  }
}

// Hhat_p = Dp (Dp Hpp Dp + D / radius)^-1 Dp (maps unscaled gradients to unscaled point steps); zero for constant points
__global__ void g_hhat_kernel(GDev d, double radius) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.P) return;
  double *o = d.Hhat + 6 * (long)p;
  if (d.pt_fixed && d.pt_fixed[p]) {
    for (int e = 0; e < 6; e++) o[e] = 0.0;
    return;
  }
  const double *Hs = d.Hpp + 6 * (long)p;
  const double s0 = d.sc_pt[3 * (long)p], s1 = d.sc_pt[3 * (long)p + 1], s2 = d.sc_pt[3 * (long)p + 2];
  const double H00 = Hs[0] * s0 * s0 + d.D_pt[3 * (long)p] / radius, H01 = Hs[1] * s0 * s1, H02 = Hs[2] * s0 * s2;
  const double H11 = Hs[3] * s1 * s1 + d.D_pt[3 * (long)p + 1] / radius, H12 = Hs[4] * s1 * s2;
  const double H22 = Hs[5] * s2 * s2 + d.D_pt[3 * (long)p + 2] / radius;
  const double c00 = H11 * H22 - H12 * H12, c01 = H12 * H02 - H01 * H22, c02 = H01 * H12 - H11 * H02;
  const double det = H00 * c00 + H01 * c01 + H02 * c02, id = 1.0 / det;
  o[0] = c00 * id * s0 * s0;
  o[1] = c01 * id * s0 * s1;
  o[2] = c02 * id * s0 * s2;
  o[3] = (H00 * H22 - H02 * H02) * id * s1 * s1;
  o[4] = (H02 * H01 - H00 * H12) * id * s1 * s2;
  o[5] = (H00 * H11 - H01 * H01) * id * s2 * s2;
}

// Sm = sc (H - sum_p W_p^T Hhat_p W_p) sc + D / radius, rhs = -sc (g - sum_p W_p^T Hhat_p g_p).
// step 1 (this kernel, one thread per entry): Sm = sc H sc (+ D / radius on the diagonal), rhs = -sc g
__global__ void g_schur_init_kernel(GDev d, double radius) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = d.nred;
  if (t < n * n) {
    const long i = t / n, j = t - i * n;
    d.Sm[t] = d.H[t] * d.sc[i] * d.sc[j] + (i == j ? d.D[i] / radius : 0.0);
  }
  if (t < n) d.rhs[t] = -d.sc[t] * d.g[t];
}
// step 2: one workgroup per point: Sm -= sc W^T Hhat W sc over the pairs of its observations, rhs += sc W^T Hhat g_p
__global__ void __launch_bounds__(64) g_schur_point_kernel(GDev d) {
  const int p = blockIdx.x;
  if (d.pt_fixed && d.pt_fixed[p]) return;
  const long o0 = d.pt_off[p], L = d.pt_off[p + 1] - o0;
  const double *hh = d.Hhat + 6 * (long)p;
  const double Hh[9] = {hh[0], hh[1], hh[2], hh[1], hh[3], hh[4], hh[2], hh[4], hh[5]};
  const double *gp = d.gpt + 3 * (long)p;
  const double hg[3] = {Hh[0] * gp[0] + Hh[1] * gp[1] + Hh[2] * gp[2], Hh[3] * gp[0] + Hh[4] * gp[1] + Hh[5] * gp[2],
                        Hh[6] * gp[0] + Hh[7] * gp[1] + Hh[8] * gp[2]};
  // items: (observation a, column x of a): W_a[:, x] = Jp_a^T Jr_a[:, x] (3-vector); T = Hhat W_a[:, x]
  for (long item = threadIdx.x; item < L * 28; item += 64) {
    const long a = item / 28;
    const int x = (int)(item - a * 28);
    int cols[28];
    obs_columns(d, o0 + a, cols);
    if (cols[x] < 0) continue;
    const double *ra = d.J + (o0 + a) * kRowJ;
    double w[3] = {0, 0, 0};
    for (int e = 0; e < 3; e++) {
      const double jx = jred(ra, e, x);
      for (int k = 0; k < 3; k++) w[k] += ra[3 + 3 * e + k] * jx;
    }
    const double sx = d.sc[cols[x]];
    atomic_add(d.rhs + cols[x], sx * (w[0] * hg[0] + w[1] * hg[1] + w[2] * hg[2]));
    const double T[3] = {Hh[0] * w[0] + Hh[1] * w[1] + Hh[2] * w[2], Hh[3] * w[0] + Hh[4] * w[1] + Hh[5] * w[2],
                         Hh[6] * w[0] + Hh[7] * w[1] + Hh[8] * w[2]};
    for (long b = 0; b < L; b++) {
      int colsb[28];
      obs_columns(d, o0 + b, colsb);
      const double *rb = d.J + (o0 + b) * kRowJ;
      for (int y = 0; y < 28; y++) {
        if (colsb[y] < 0) continue;
        double wy[3] = {0, 0, 0};
        for (int e = 0; e < 3; e++) {
          const double jy = jred(rb, e, y);
          if (jy == 0.0) continue;
          for (int k = 0; k < 3; k++) wy[k] += rb[3 + 3 * e + k] * jy;
        }
        const double v = T[0] * wy[0] + T[1] * wy[1] + T[2] * wy[2];
        if (v != 0.0) atomic_add(d.Sm + (long)cols[x] * d.nred + colsb[y], -sx * v * d.sc[colsb[y]]);
      }
    }
  }
}

// delta (unscaled reduced step) = sc * (solution of the scaled system, left in rhs)
__global__ void g_unscale_kernel(GDev d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.nred) d.delta[i] = d.sc[i] * d.rhs[i];
}
// point steps dp = -Hhat (g_p + W_p delta), candidate points; step / parameter norms of the points into scal[2], scal[3]
__global__ void g_point_step_kernel(GDev d) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  double sn = 0.0, xn = 0.0;
  if (p < d.P) {
    double v[3] = {d.gpt[3 * (long)p], d.gpt[3 * (long)p + 1], d.gpt[3 * (long)p + 2]};
    for (long o = d.pt_off[p]; o < d.pt_off[p + 1]; o++) {
      const double *row = d.J + o * kRowJ;
      int cols[28];
      obs_columns(d, o, cols);
      for (int e = 0; e < 3; e++) {
        double u = 0.0;
        for (int x = 0; x < 28; x++)
          if (cols[x] >= 0) u += jred(row, e, x) * d.delta[cols[x]];
        for (int k = 0; k < 3; k++) v[k] += row[3 + 3 * e + k] * u;
      }
    }
    const double *hh = d.Hhat + 6 * (long)p;
    const double dp[3] = {-(hh[0] * v[0] + hh[1] * v[1] + hh[2] * v[2]), -(hh[1] * v[0] + hh[3] * v[1] + hh[4] * v[2]),
                          -(hh[2] * v[0] + hh[4] * v[1] + hh[5] * v[2])};
    for (int k = 0; k < 3; k++) {
      d.dpt[3 * (long)p + k] = dp[k];
      d.pts_n[3 * (long)p + k] = d.pts[3 * (long)p + k] + dp[k];
      sn += dp[k] * dp[k];
      if (!(d.pt_fixed && d.pt_fixed[p])) xn += d.pts[3 * (long)p + k] * d.pts[3 * (long)p + k];
    }
  }
  block_accumulate(sn, d.scal + 2);
  block_accumulate(xn, d.scal + 3);
}
// candidate of the reduced blocks (single block kernel is enough: n_r is small) + their norms
__global__ void g_candidate_kernel(GDev d) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double sn = 0.0, xn = 0.0;
  auto upd = [&](const double *cur, double *nxt, int n, int idx) {  // constant blocks are not part of the problem Ceres sees
    for (int k = 0; k < n; k++) {
      const double dl = idx < 0 ? 0.0 : d.delta[idx + k];
      nxt[k] = cur[k] + dl;
      sn += dl * dl;
      if (idx >= 0) xn += cur[k] * cur[k];
    }
  };
  if (t < d.NC) {
    const int nk = model_num_params(d.cam_model[t]);
    upd(d.cam + 16 * t, d.cam_n + 16 * t, nk, d.cam_idx[t]);
    for (int k = nk; k < 16; k++) d.cam_n[16 * t + k] = d.cam[16 * t + k];
    upd(d.bias + 7 * t, d.bias_n + 7 * t, 7, d.bias_idx[t]);
  }
  if (t < d.NR) upd(d.rc + 6 * t, d.rc_n + 6 * t, 6, d.rc_idx[t]);
  if (t < d.NI) upd(d.inst + 6 * t, d.inst_n + 6 * t, 6, d.inst_idx[t]);
  block_accumulate(sn, d.scal + 2);
  block_accumulate(xn, d.scal + 3);
}
// model cost change of the reprojection rows and the point priors: scal[1] -= u (r + u / 2), u = J delta
__global__ void g_model_change_kernel(GDev d) {
  const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  double mc = 0.0;
  if (o < d.M) {
    const double *row = d.J + o * kRowJ;
    int cols[28];
    obs_columns(d, o, cols);
    const int p = d.o_point[o];
    for (int e = 0; e < 3; e++) {
      double u = 0.0;
      for (int x = 0; x < 28; x++)
        if (cols[x] >= 0) u += jred(row, e, x) * d.delta[cols[x]];
      for (int k = 0; k < 3; k++) u += row[3 + 3 * e + k] * d.dpt[3 * (long)p + k];
      mc -= u * (row[e] + 0.5 * u);
    }
  }
  if (o < d.P && d.pt_prior && d.pt_prior_sigma && d.pt_prior_sigma[3 * o] > 0) {
    const int n = d.pt_prior_alt[o] ? 3 : 2;
    for (int a = 0; a < n; a++) {
      const double sc = 1.0 / fmax(d.pt_prior_sigma[3 * o + a], 2.220446049250313e-16);
      const double r = sc * (d.pts[3 * o + a] - d.pt_prior[3 * o + a]), u = sc * d.dpt[3 * o + a];
      mc -= u * (r + 0.5 * u);
    }
  }
  block_accumulate(mc, d.scal + 1);
}
__global__ void g_absmax_kernel(const double *a, long n, const double *b, long m, double *out) {
  double v = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n + m; i += (long)gridDim.x * blockDim.x) v = fmax(v, fabs(i < n ? a[i] : b[i - n]));
  for (int k = 32; k >= 1; k >>= 1) v = fmax(v, __shfl_xor(v, k));
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(v));  // non-negative doubles order as integers
}

struct Arena {  // one device allocation, released on scope exit
  std::vector<void *> blocks;
  hipError_t err = hipSuccess;
  osfm_ctx *ctx = nullptr;  // set: an out-of-memory allocation drops the context's block cache and retries
  ~Arena() {
    for (void *b : blocks) (void)hipFree(b);
  }
  template <class T>
  T *alloc(size_t n) {
    void *p = nullptr;
    const hipError_t e = osfm_malloc_retry(ctx, &p, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) {
      err = e;
      return nullptr;
    }
    blocks.push_back(p);
    return (T *)p;
  }
  template <class T>
  T *upload(const T *h, size_t n, hipStream_t st) {
    T *p = alloc<T>(n);
    if (p && n && h) {
      const hipError_t e = hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, st);
      if (e != hipSuccess) err = e;
    }
    return p;
  }
};

inline unsigned nblk(long n, int tpb = TPB) { return (unsigned)((n + tpb - 1) / tpb > 0 ? (n + tpb - 1) / tpb : 1); }

}  // namespace

extern "C" int osfm_bundle_solve(osfm_ctx *ctx, osfm_bundle_problem *P, const osfm_ba_options *O, osfm_ba_report *Rp) {
  OSFM_REQUIRE(ctx && P && O && Rp, OSFM_E_INVALID, "osfm_bundle_solve: null argument");
  memset(Rp, 0, sizeof(*Rp));
  const int NC = P->n_cameras, NR = P->n_rig_cameras, NI = P->n_rig_instances, S = P->n_shots, NP = P->n_points;
  const long M0 = P->n_obs;  // reprojection observations; M below also counts the depth-prior rows
  OSFM_REQUIRE(NC > 0 && NR > 0 && NI > 0 && S > 0 && NP >= 0 && M0 >= 0, OSFM_E_INVALID, "osfm_bundle_solve: empty problem");
  // depth priors become rows of their own (kind 1 / 2) behind the observation they belong to: same parameter blocks, one residual
  std::vector<long> depth_obs;
  if (P->obs_depth && P->obs_depth_sigma)
    for (long o = 0; o < M0; o++)
      if (P->obs_depth_sigma[o] > 0) {
        OSFM_REQUIRE(std::isfinite(P->obs_depth[o]), OSFM_E_INVALID, "observation %ld (shot %d) has non-finite depth prior", o, P->obs_shot ? P->obs_shot[o] : -1);
        depth_obs.push_back(o);
      }
  const long M = M0 + (long)depth_obs.size();
  OSFM_REQUIRE(P->cam_model && P->cam_params && P->cam_prior && P->cam_sigma && P->cam_fixed && P->rig_camera_pose && P->rig_camera_fixed &&
                   P->rig_instance_pose && P->shot_rig_instance && P->shot_rig_camera && P->shot_camera && (NP == 0 || P->points) &&
                   (M == 0 || (P->obs_shot && P->obs_point && P->obs_xy && P->obs_sigma)),
               OSFM_E_INVALID, "osfm_bundle_solve: null array");
  for (int c = 0; c < NC; c++)
    OSFM_REQUIRE(P->cam_model[c] >= OSFM_CAMERA_PERSPECTIVE && P->cam_model[c] <= OSFM_CAMERA_SPHERICAL, OSFM_E_INVALID, "camera %d: model %d", c,
                 P->cam_model[c]);
  for (int s = 0; s < S; s++)
    OSFM_REQUIRE(P->shot_rig_instance[s] >= 0 && P->shot_rig_instance[s] < NI && P->shot_rig_camera[s] >= 0 && P->shot_rig_camera[s] < NR &&
                     P->shot_camera[s] >= 0 && P->shot_camera[s] < NC,
                 OSFM_E_INVALID, "shot %d references a missing rig instance / rig camera / camera", s);
  for (long o = 0; o < M0; o++)
    OSFM_REQUIRE(P->obs_shot[o] >= 0 && P->obs_shot[o] < S && P->obs_point[o] >= 0 && P->obs_point[o] < NP && P->obs_sigma[o] > 0, OSFM_E_INVALID,
                 "observation %ld is out of range", o);
  if (P->rig_instance_gps && P->rig_instance_gps_sigma)
    for (int i = 0; i < NI; i++)
      OSFM_REQUIRE(!(P->rig_instance_gps_sigma[3 * i] > 0) || (P->rig_instance_bias_camera && P->rig_instance_bias_camera[i] >= 0 &&
                                                               P->rig_instance_bias_camera[i] < NC),
                   OSFM_E_INVALID, "rig instance %d has a position prior but no bias reference camera", i);
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const auto t_start = std::chrono::steady_clock::now();

  // ---- reduced indices: instances, rig cameras, cameras, biases ----
  std::vector<int> inst_idx(NI), rc_idx(NR), cam_idx(NC), bias_idx(NC);
  std::vector<unsigned char> rc_useful(NR);
  int nred = 0;
  for (int i = 0; i < NI; i++) {
    const bool fixed = P->rig_instance_fixed && P->rig_instance_fixed[i];
    inst_idx[i] = fixed ? -1 : nred;
    if (!fixed) nred += 6;
  }
  for (int q = 0; q < NR; q++) {
    const bool fixed = P->rig_camera_fixed[q] != 0;
    bool zero = true;
    for (int k = 0; k < 6; k++) zero = zero && P->rig_camera_pose[6 * q + k] == 0.0;
    rc_useful[q] = !(fixed && zero);  // IsRigCameraUseful, bundle_adjuster.cc:17-20
    rc_idx[q] = fixed ? -1 : nred;
    if (!fixed) nred += 6;
  }
  for (int c = 0; c < NC; c++) {
    const int nk = model_num_params(P->cam_model[c]);
    const bool fixed = P->cam_fixed[c] != 0 || nk == 0;
    cam_idx[c] = fixed ? -1 : nred;
    if (!fixed) nred += nk;
  }
  std::vector<double> bias((size_t)NC * 7, 0.0);
  for (int c = 0; c < NC; c++) {
    if (P->bias)
      for (int k = 0; k < 7; k++) bias[7 * c + k] = P->bias[7 * c + k];
    else
      bias[7 * c + 6] = 1.0;
    const bool fixed = !P->bias || !P->bias_fixed || P->bias_fixed[c] != 0;
    bias_idx[c] = fixed ? -1 : nred;
    if (!fixed) nred += 7;
  }
  OSFM_REQUIRE(nred > 0 || NP > 0, OSFM_E_INVALID, "osfm_bundle_solve: nothing to optimise");
  OSFM_REQUIRE((double)nred * nred * 16.0 < 200e9, OSFM_E_UNSUPPORTED,
               "osfm_bundle_solve: %d reduced unknowns need a dense system beyond the device memory; use osfm_ba_solve for sequences of this size", nred);

  // ---- observations point-major ----
  // row -> (source observation, kind); rows M0 .. M - 1 are the depth priors
  auto src_of = [&](long o) { return o < M0 ? o : depth_obs[(size_t)(o - M0)]; };
  std::vector<long> pt_off((size_t)NP + 1, 0);
  for (long o = 0; o < M; o++) pt_off[(size_t)P->obs_point[src_of(o)] + 1]++;
  for (int p = 0; p < NP; p++) pt_off[(size_t)p + 1] += pt_off[(size_t)p];
  std::vector<long> fill(pt_off.begin(), pt_off.end() - 1), perm((size_t)M);
  for (long o = 0; o < M; o++) perm[(size_t)fill[(size_t)P->obs_point[src_of(o)]]++] = o;
  std::vector<int> o_shot((size_t)M), o_point((size_t)M);
  std::vector<double> o_x((size_t)M), o_y((size_t)M), o_sigma((size_t)M);
  std::vector<unsigned char> o_kind((size_t)M, 0);
  for (long k = 0; k < M; k++) {
    const long o = perm[(size_t)k], so = src_of(o);
    o_shot[(size_t)k] = P->obs_shot[so];
    o_point[(size_t)k] = P->obs_point[so];
    if (o < M0) {
      o_x[(size_t)k] = P->obs_xy[2 * o];
      o_y[(size_t)k] = P->obs_xy[2 * o + 1];
      o_sigma[(size_t)k] = P->obs_sigma[o];
    } else {
      o_x[(size_t)k] = P->obs_depth[so];
      o_y[(size_t)k] = 0.0;
      o_sigma[(size_t)k] = P->obs_depth_sigma[so];
      o_kind[(size_t)k] = (!P->obs_depth_radial || P->obs_depth_radial[so]) ? 2 : 1;
    }
  }

  Arena A;
  A.ctx = ctx;
  GDev d;
  memset(&d, 0, sizeof(d));
  d.NC = NC; d.NR = NR; d.NI = NI; d.S = S; d.P = NP; d.M = M; d.nred = nred;
  d.cam = A.upload(P->cam_params, (size_t)NC * 16, st);
  d.bias = A.upload(bias.data(), (size_t)NC * 7, st);
  d.rc = A.upload(P->rig_camera_pose, (size_t)NR * 6, st);
  d.inst = A.upload(P->rig_instance_pose, (size_t)NI * 6, st);
  d.pts = A.upload(P->points, (size_t)NP * 3, st);
  d.cam_n = A.alloc<double>((size_t)NC * 16); d.bias_n = A.alloc<double>((size_t)NC * 7); d.rc_n = A.alloc<double>((size_t)NR * 6);
  d.inst_n = A.alloc<double>((size_t)NI * 6); d.pts_n = A.alloc<double>((size_t)NP * 3);
  d.cam_model = A.upload(P->cam_model, (size_t)NC, st);
  d.cam_prior = A.upload(P->cam_prior, (size_t)NC * 16, st);
  d.cam_sigma = A.upload(P->cam_sigma, (size_t)NC * 16, st);
  if (P->rig_camera_prior && P->rig_camera_sigma) {
    d.rc_prior = A.upload(P->rig_camera_prior, (size_t)NR * 6, st);
    d.rc_sigma = A.upload(P->rig_camera_sigma, (size_t)NR * 6, st);
  }
  if (P->rig_instance_gps && P->rig_instance_gps_sigma) {
    d.gps = A.upload(P->rig_instance_gps, (size_t)NI * 3, st);
    d.gps_sigma = A.upload(P->rig_instance_gps_sigma, (size_t)NI * 3, st);
    d.inst_bias_cam = A.upload(P->rig_instance_bias_camera, (size_t)NI, st);
  }
  if (P->shot_up && P->shot_up_sigma) {
    d.up = A.upload(P->shot_up, (size_t)S * 3, st);
    d.up_sigma = A.upload(P->shot_up_sigma, (size_t)S, st);
  }
  if (P->shot_pan && P->shot_pan_sigma) { d.pan = A.upload(P->shot_pan, (size_t)S, st); d.pan_sigma = A.upload(P->shot_pan_sigma, (size_t)S, st); }
  if (P->shot_tilt && P->shot_tilt_sigma) { d.tilt = A.upload(P->shot_tilt, (size_t)S, st); d.tilt_sigma = A.upload(P->shot_tilt_sigma, (size_t)S, st); }
  if (P->shot_roll && P->shot_roll_sigma) { d.roll = A.upload(P->shot_roll, (size_t)S, st); d.roll_sigma = A.upload(P->shot_roll_sigma, (size_t)S, st); }
  if (P->point_prior && P->point_prior_sigma && NP > 0) {
    d.pt_prior = A.upload(P->point_prior, (size_t)NP * 3, st);
    d.pt_prior_sigma = A.upload(P->point_prior_sigma, (size_t)NP * 3, st);
    std::vector<unsigned char> alt((size_t)NP, 1);
    if (P->point_prior_has_altitude) alt.assign(P->point_prior_has_altitude, P->point_prior_has_altitude + NP);
    d.pt_prior_alt = A.upload(alt.data(), (size_t)NP, st);
    OSFM_HIP(hipStreamSynchronize(st));  // `alt` goes out of scope
  }
  d.cam_idx = A.upload(cam_idx.data(), (size_t)NC, st);
  d.bias_idx = A.upload(bias_idx.data(), (size_t)NC, st);
  d.rc_idx = A.upload(rc_idx.data(), (size_t)NR, st);
  d.inst_idx = A.upload(inst_idx.data(), (size_t)NI, st);
  d.rc_useful = A.upload(rc_useful.data(), (size_t)NR, st);
  if (P->point_fixed && NP > 0) d.pt_fixed = A.upload(P->point_fixed, (size_t)NP, st);
  d.shot_inst = A.upload(P->shot_rig_instance, (size_t)S, st);
  d.shot_rc = A.upload(P->shot_rig_camera, (size_t)S, st);
  d.shot_cam = A.upload(P->shot_camera, (size_t)S, st);
  d.o_shot = A.upload(o_shot.data(), (size_t)M, st);
  d.o_point = A.upload(o_point.data(), (size_t)M, st);
  d.o_x = A.upload(o_x.data(), (size_t)M, st);
  d.o_y = A.upload(o_y.data(), (size_t)M, st);
  d.o_sigma = A.upload(o_sigma.data(), (size_t)M, st);
  d.o_kind = depth_obs.empty() ? nullptr : A.upload(o_kind.data(), (size_t)M, st);
  d.pt_off = A.upload(pt_off.data(), (size_t)NP + 1, st);
  d.J = A.alloc<double>((size_t)M * kRowJ);
  d.Hpp = A.alloc<double>((size_t)NP * 6); d.gpt = A.alloc<double>((size_t)NP * 3); d.Hhat = A.alloc<double>((size_t)NP * 6);
  d.sc_pt = A.alloc<double>((size_t)NP * 3); d.D_pt = A.alloc<double>((size_t)NP * 3); d.dpt = A.alloc<double>((size_t)NP * 3);
  const size_t n2 = (size_t)nred * nred;
  d.H = A.alloc<double>(n2); d.Sm = A.alloc<double>(n2);
  d.g = A.alloc<double>(nred); d.diag = A.alloc<double>(nred); d.sc = A.alloc<double>(nred); d.D = A.alloc<double>(nred);
  d.rhs = A.alloc<double>(nred); d.delta = A.alloc<double>(nred);
  d.scal = A.alloc<double>(16);
  int *d_info = A.alloc<int>(1);
  OSFM_REQUIRE(A.err == hipSuccess, OSFM_E_NOMEM, "osfm_bundle_solve: device allocation / upload failed: %s", hipGetErrorString(A.err));
  OSFM_HIP(hipStreamSynchronize(st));

  rocblas_handle blas = nullptr;
  OSFM_REQUIRE(rocblas_create_handle(&blas) == rocblas_status_success, OSFM_E_HIP, "rocblas_create_handle failed");
  struct BlasGuard {
    rocblas_handle h;
    ~BlasGuard() { rocblas_destroy_handle(h); }
  } blas_guard{blas};
  rocblas_set_stream(blas, st);

  const int nprior = NC + NR + NI + 4 * S;
  std::vector<double> hs(16);
  // cost at (cam, bias, rc, inst, pts); jac: also fills the rows and the normal equations
  auto evaluate = [&](const double *cam, const double *bs, const double *rc, const double *inst, const double *pts, bool jac, double *cost) -> int {
    OSFM_HIP(hipMemsetAsync(d.scal, 0, 16 * sizeof(double), st));
    if (jac) {
      OSFM_HIP(hipMemsetAsync(d.H, 0, n2 * sizeof(double), st));
      OSFM_HIP(hipMemsetAsync(d.g, 0, (size_t)nred * sizeof(double), st));
      if (M > 0) hipLaunchKernelGGL(g_eval_kernel<true>, dim3(nblk(M)), dim3(TPB), 0, st, d, cam, rc, inst, pts, O->loss, O->loss_threshold);
      if (NP > 0) hipLaunchKernelGGL(g_point_kernel, dim3(nblk(NP)), dim3(TPB), 0, st, d, pts);
      if (M > 0 && nred > 0) hipLaunchKernelGGL(g_reduced_kernel, dim3(nblk(M)), dim3(TPB), 0, st, d);
    } else if (M > 0) {
      hipLaunchKernelGGL(g_eval_kernel<false>, dim3(nblk(M)), dim3(TPB), 0, st, d, cam, rc, inst, pts, O->loss, O->loss_threshold);
    }
    if (NP > 0) hipLaunchKernelGGL(g_point_prior_cost_kernel, dim3(nblk(NP)), dim3(TPB), 0, st, d, pts);
    hipLaunchKernelGGL(g_prior_kernel, dim3(nblk(nprior, 64)), dim3(64), 0, st, d, cam, bs, rc, inst, jac ? 1 : 0);
    if (jac && nred > 0) hipLaunchKernelGGL(g_diag_kernel, dim3(nblk(nred)), dim3(TPB), 0, st, d);
    OSFM_HIP(hipGetLastError());
    OSFM_HIP(hipMemcpyAsync(hs.data(), d.scal, sizeof(double), hipMemcpyDeviceToHost, st));
    OSFM_HIP(hipStreamSynchronize(st));
    *cost = hs[0];
    return OSFM_OK;
  };

  double cost = 0.0;
  int rc = evaluate(d.cam, d.bias, d.rc, d.inst, d.pts, true, &cost);
  if (rc != OSFM_OK) return rc;
  Rp->initial_cost = cost;
  Rp->cost_history[0] = cost;
  Rp->seconds_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  const auto t_run = std::chrono::steady_clock::now();

  double radius = O->initial_radius, decrease_factor = 2.0;
  bool need_prepare = true, have_scale = false;
  int iter = 0, n_invalid = 0;
  double gmax = 0.0, lin_seconds = 0.0;
  Rp->termination = 0;
  for (;;) {
    if (need_prepare) {
      hipLaunchKernelGGL(g_scale_kernel, dim3(nblk(std::max<long>(nred, 3L * NP))), dim3(TPB), 0, st, d, have_scale ? 0 : 1);
      have_scale = true;
      OSFM_HIP(hipMemsetAsync(d.scal + 4, 0, sizeof(double), st));
      hipLaunchKernelGGL(g_absmax_kernel, dim3(64), dim3(256), 0, st, d.g, (long)nred, d.gpt, 3L * NP, d.scal + 4);
      OSFM_HIP(hipMemcpyAsync(hs.data(), d.scal + 4, sizeof(double), hipMemcpyDeviceToHost, st));
      OSFM_HIP(hipStreamSynchronize(st));
      gmax = hs[0];
      need_prepare = false;
    }
    if (iter >= O->max_iterations) { Rp->termination = 0; break; }
    if (gmax <= O->gradient_tolerance) { Rp->termination = 2; break; }
    if (radius < 1e-32) { Rp->termination = 4; break; }
    iter++;
    if (iter < 256) Rp->cost_history[iter] = cost;
    const auto t_lin = std::chrono::steady_clock::now();
    bool bad = false;
    if (NP > 0) hipLaunchKernelGGL(g_hhat_kernel, dim3(nblk(NP)), dim3(TPB), 0, st, d, radius);
    if (nred > 0) {
      hipLaunchKernelGGL(g_schur_init_kernel, dim3(nblk((long)n2)), dim3(TPB), 0, st, d, radius);
      if (NP > 0) hipLaunchKernelGGL(g_schur_point_kernel, dim3(NP), dim3(64), 0, st, d);
      OSFM_HIP(hipGetLastError());
      // dense Cholesky of the reduced system (the exact solve SPARSE_SCHUR performs), solution left in rhs
      if (rocsolver_dpotrf(blas, rocblas_fill_lower, nred, d.Sm, nred, d_info) != rocblas_status_success) bad = true;
      int info = 0;
      OSFM_HIP(hipMemcpyAsync(&info, d_info, sizeof(int), hipMemcpyDeviceToHost, st));
      OSFM_HIP(hipStreamSynchronize(st));
      if (info != 0) bad = true;
      if (!bad && rocsolver_dpotrs(blas, rocblas_fill_lower, nred, 1, d.Sm, nred, d.rhs, nred) != rocblas_status_success) bad = true;
      hipLaunchKernelGGL(g_unscale_kernel, dim3(nblk(nred)), dim3(TPB), 0, st, d);
    }
    OSFM_HIP(hipMemsetAsync(d.scal + 1, 0, 3 * sizeof(double), st));
    if (NP > 0) hipLaunchKernelGGL(g_point_step_kernel, dim3(nblk(NP)), dim3(TPB), 0, st, d);
    hipLaunchKernelGGL(g_candidate_kernel, dim3(nblk(std::max(std::max(NC, NR), NI))), dim3(TPB), 0, st, d);
    if (std::max<long>(M, NP) > 0) hipLaunchKernelGGL(g_model_change_kernel, dim3(nblk(std::max<long>(M, NP))), dim3(TPB), 0, st, d);
    hipLaunchKernelGGL(g_prior_kernel, dim3(nblk(nprior, 64)), dim3(64), 0, st, d, d.cam, d.bias, d.rc, d.inst, 2);
    OSFM_HIP(hipGetLastError());
    OSFM_HIP(hipMemcpyAsync(hs.data(), d.scal, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    OSFM_HIP(hipStreamSynchronize(st));
    lin_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_lin).count();
    const double model_change = hs[1], step_sq = hs[2], x_sq = hs[3];
    if (bad || !(model_change > 0) || !std::isfinite(model_change)) {  // HandleInvalidStep + StepIsInvalid
      radius *= 0.5;
      if (++n_invalid >= 5) { Rp->termination = -1; break; }
      continue;
    }
    n_invalid = 0;
    double cost_n = 0.0;
    rc = evaluate(d.cam_n, d.bias_n, d.rc_n, d.inst_n, d.pts_n, false, &cost_n);
    if (rc != OSFM_OK) return rc;
    const double step_norm = std::sqrt(step_sq), x_norm = std::sqrt(x_sq);
    if (step_norm <= O->parameter_tolerance * (x_norm + O->parameter_tolerance)) { Rp->termination = 3; break; }
    const double cost_change = cost - cost_n;
    if (std::fabs(cost_change) <= O->function_tolerance * cost) { Rp->termination = 1; break; }
    const double rho = cost_change / model_change;
    if (O->verbose) fprintf(stderr, "[osfm_bundle] it %d cost %.9e -> %.9e rho %.3f radius %.3e\n", iter, cost, cost_n, rho, radius);
    if (rho > 1e-3) {  // StepAccepted
      std::swap(d.cam, d.cam_n); std::swap(d.bias, d.bias_n); std::swap(d.rc, d.rc_n); std::swap(d.inst, d.inst_n); std::swap(d.pts, d.pts_n);
      const double t = 2.0 * rho - 1.0;
      radius = std::fmin(1e16, radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t));
      decrease_factor = 2.0;
      Rp->successful_steps++;
      rc = evaluate(d.cam, d.bias, d.rc, d.inst, d.pts, true, &cost);
      if (rc != OSFM_OK) return rc;
      need_prepare = true;
    } else {  // StepRejected
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
    }
    if (iter < 256) Rp->cost_history[iter] = cost;
  }
  Rp->iterations = iter;
  Rp->final_cost = cost;
  Rp->seconds_linear_solver = lin_seconds;
  OSFM_HIP(hipStreamSynchronize(st));
  const auto t_tear = std::chrono::steady_clock::now();
  Rp->seconds_run = std::chrono::duration<double>(t_tear - t_run).count();
  // ---- results ----
  OSFM_HIP(hipMemcpyAsync(P->cam_params, d.cam, (size_t)NC * 16 * sizeof(double), hipMemcpyDeviceToHost, st));
  if (P->bias) OSFM_HIP(hipMemcpyAsync(P->bias, d.bias, (size_t)NC * 7 * sizeof(double), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipMemcpyAsync(P->rig_camera_pose, d.rc, (size_t)NR * 6 * sizeof(double), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipMemcpyAsync(P->rig_instance_pose, d.inst, (size_t)NI * 6 * sizeof(double), hipMemcpyDeviceToHost, st));
  if (NP > 0) OSFM_HIP(hipMemcpyAsync(P->points, d.pts, (size_t)NP * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
  if (P->reproj_err && M > 0) {
    double *d_err = A.alloc<double>((size_t)M * 3);
    OSFM_REQUIRE(d_err != nullptr, OSFM_E_NOMEM, "osfm_bundle_solve: out of device memory");
    hipLaunchKernelGGL(g_reproj_kernel, dim3(nblk(M)), dim3(TPB), 0, st, d, d_err);
    std::vector<double> e((size_t)M * 3);
    OSFM_HIP(hipMemcpyAsync(e.data(), d_err, (size_t)M * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    OSFM_HIP(hipStreamSynchronize(st));
    for (long k = 0; k < M; k++)
      if (perm[(size_t)k] < M0)  // the depth-prior rows have no reprojection error
        for (int a = 0; a < 3; a++) P->reproj_err[3 * perm[(size_t)k] + a] = e[(size_t)(3 * k + a)];
  }
  OSFM_HIP(hipStreamSynchronize(st));
  Rp->seconds_teardown = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tear).count();
  Rp->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  for (int i = 0; i < NC * 16; i++) OSFM_REQUIRE(std::isfinite(P->cam_params[i]), OSFM_E_NUMERIC, "camera has either NaN or INF values");
  for (int i = 0; i < NI * 6; i++) OSFM_REQUIRE(std::isfinite(P->rig_instance_pose[i]), OSFM_E_NUMERIC, "rig instance has either NaN or INF values");
  for (long i = 0; i < (long)NP * 3; i++) OSFM_REQUIRE(std::isfinite(P->points[i]), OSFM_E_NUMERIC, "point has either NaN or INF values");
  return OSFM_OK;
}
