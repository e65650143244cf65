// ba_math.h -- device maths shared by the two bundle-adjustment solvers (ba.hip: the streaming Schur-PCG solver of the headline
// configuration; ba_general.hip: the general solver for every camera model, rigs, biases and control points).
// Same formulas, same order as the CPU statement of the reference's functors (geometry/transformations_functions.h,
// camera_projections_functions.h, camera_distortions_functions.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/osfm_mi355.h"

namespace osfm_ba {

constexpr int TPB = 256;
constexpr double kEps = 2.220446049250313e-16;

// ------------------------------------------------------------------------------------------
// device maths (same formulas, same order as the CPU statement of the reference's functors)
// ------------------------------------------------------------------------------------------
static __device__ void rot_and_derivs(const double *r, double *R, double *dR /*[3][9]*/) {
  const double a[3] = {-r[0], -r[1], -r[2]};
  const double th2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
  if (!(th2 > kEps)) {
    for (int i = 0; i < 9; i++) R[i] = K[i];
    R[0] = R[4] = R[8] = 1.0;
    for (int i = 0; i < 27; i++) dR[i] = 0;
    dR[0 * 9 + 5] = -1; dR[0 * 9 + 7] = 1;
    dR[1 * 9 + 2] = 1;  dR[1 * 9 + 6] = -1;
    dR[2 * 9 + 1] = -1; dR[2 * 9 + 3] = 1;
    return;
  }
  const double th = sqrt(th2), s = sin(th), c = cos(th);
  const double sh = sin(0.5 * th);
  const double A = s / th, B = 2.0 * sh * sh / th2;
  const double Ap = (c - A) / th2, Bp = (A - 2.0 * B) / th2;
  double K2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double v = 0;
      for (int m = 0; m < 3; m++) v += K[3 * i + m] * K[3 * m + j];
      K2[3 * i + j] = v;
    }
  for (int i = 0; i < 9; i++) R[i] = A * K[i] + B * K2[i];
  R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
  for (int k = 0; k < 3; k++) {
    double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (k == 0) { E[5] = -1; E[7] = 1; }
    if (k == 1) { E[2] = 1; E[6] = -1; }
    if (k == 2) { E[1] = -1; E[3] = 1; }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double ek_k = 0, k_ek = 0;
        for (int m = 0; m < 3; m++) {
          ek_k += E[3 * i + m] * K[3 * m + j];
          k_ek += K[3 * i + m] * E[3 * m + j];
        }
        dR[9 * k + 3 * i + j] = a[k] * Ap * K[3 * i + j] + A * E[3 * i + j] + a[k] * Bp * K2[3 * i + j] + B * (ek_k + k_ek);
      }
  }
}

// PROJ stage of the camera (camera_projections_functions.h): PerspectiveProjection (:88-117) or
// FisheyeProjection (:9-85: theta / r * (x, y), theta = atan2(r, z); perspective below r = 1e-8).
template <bool JAC>
__device__ __forceinline__ void project_stage(int model, const double *Xc, double &u, double &v, double *jp) {
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  const double r2 = x * x + y * y, r = sqrt(r2);
  if (model == OSFM_CAMERA_FISHEYE && !(r < 1e-8)) {
    const double theta = atan2(r, z);
    u = theta / r * x;
    v = theta / r * y;
    if (JAC) {
      const double R2 = r2 + z * z, x2 = x * x, y2 = y * y, z2 = z * z;
      const double inv_denom = 1.0 / (r2 * R2 * r);
      jp[0] = (x2 * y2 * theta + y2 * y2 * theta + y2 * z2 * theta + x2 * z * r) * inv_denom;
      jp[1] = x * (y * z * r - y * theta * R2) * inv_denom;
      jp[2] = -x / R2;
      jp[3] = y * (x * z * r - x * theta * R2) * inv_denom;
      jp[4] = (x2 * y2 * theta + x2 * x2 * theta + x2 * z2 * theta + y2 * z * r) * inv_denom;
      jp[5] = -y / R2;
    }
    return;
  }
  const double iz = 1.0 / z;
  u = x * iz;
  v = y * iz;
  if (JAC) {
    jp[0] = iz; jp[1] = 0.0; jp[2] = -x * iz * iz;
    jp[3] = 0.0; jp[4] = iz; jp[5] = -y * iz * iz;
  }
}

// ---- the other 2-D camera models, for CONSTANT cameras (no intrinsics Jacobian needed) ----------------
// ProjectGeneric<PROJ, DISTO, AFF> (camera_instances.h:127-160), parameters [PROJ][DISTO][AFF] in the
// reference's native order.  The 2x2 Jacobian of the distortion stage is taken by forward-mode duals
// (what the reference's autodiff twin does with the same formulas), the projection stage analytically.
struct D2 {
  double v, a, b;  // value, d/dx, d/dy
};
__device__ __forceinline__ D2 d2c(double c) { return D2{c, 0.0, 0.0}; }
__device__ __forceinline__ D2 operator+(D2 p, D2 q) { return D2{p.v + q.v, p.a + q.a, p.b + q.b}; }
__device__ __forceinline__ D2 operator*(D2 p, D2 q) { return D2{p.v * q.v, p.a * q.v + p.v * q.a, p.b * q.v + p.v * q.b}; }
__device__ __forceinline__ D2 operator*(D2 p, double c) { return D2{p.v * c, p.a * c, p.b * c}; }

__device__ __forceinline__ void model_layout(int model, int &proj, int &kind, int &nd, int &na) {
  switch (model) {
    case OSFM_CAMERA_PERSPECTIVE: proj = 0; kind = 1; nd = 2; na = 1; break;
    case OSFM_CAMERA_FISHEYE: proj = 1; kind = 1; nd = 2; na = 1; break;
    case OSFM_CAMERA_BROWN: proj = 0; kind = 3; nd = 5; na = 4; break;
    case OSFM_CAMERA_FISHEYE_OPENCV: proj = 1; kind = 2; nd = 4; na = 4; break;
    case OSFM_CAMERA_FISHEYE62: proj = 1; kind = 4; nd = 8; na = 4; break;
    case OSFM_CAMERA_FISHEYE624: proj = 1; kind = 5; nd = 12; na = 4; break;
    case OSFM_CAMERA_DUAL: proj = 2; kind = 1; nd = 2; na = 1; break;
    case OSFM_CAMERA_RADIAL: proj = 0; kind = 1; nd = 2; na = 4; break;
    default: proj = 0; kind = 0; nd = 1; na = 4; break;  // SIMPLE_RADIAL
  }
}

static __device__ void distort_d2(int kind, const double *k, D2 x, D2 y, D2 &ox, D2 &oy) {
  const D2 r2 = x * x + y * y;
  D2 rad, tx = d2c(0.0), ty = d2c(0.0);
  switch (kind) {
    case 0: rad = d2c(1.0) + r2 * k[0]; break;
    case 1: rad = d2c(1.0) + r2 * (d2c(k[0]) + r2 * k[1]); break;
    case 2: rad = d2c(1.0) + r2 * (d2c(k[0]) + r2 * (d2c(k[1]) + r2 * (d2c(k[2]) + r2 * k[3]))); break;
    case 3: rad = d2c(1.0) + r2 * (d2c(k[0]) + r2 * (d2c(k[1]) + r2 * k[2])); break;
    default: {  // six radial coefficients, Horner as RadialDistortion (camera_distortions_functions.h:481-485)
      D2 acc = d2c(k[4]) + r2 * k[5];
      acc = d2c(k[3]) + r2 * acc;
      acc = d2c(k[2]) + r2 * acc;
      acc = d2c(k[1]) + r2 * acc;
      acc = d2c(k[0]) + r2 * acc;
      rad = d2c(1.0) + r2 * acc;
    } break;
  }
  if (kind >= 3) {  // tangential: 2 p1 x y + p2 (r2 + 2 x^2), 2 p2 x y + p1 (r2 + 2 y^2)
    const double p1 = kind == 3 ? k[3] : k[6], p2 = kind == 3 ? k[4] : k[7];
    const D2 xy = x * y;
    tx = xy * (2.0 * p1) + (r2 + (x * x) * 2.0) * p2;
    ty = xy * (2.0 * p2) + (r2 + (y * y) * 2.0) * p1;
  }
  if (kind == 5) {  // thin prism: s0 r2 + s1 r2^2, s2 r2 + s3 r2^2
    const D2 r4 = r2 * r2;
    tx = tx + (r2 * k[8] + r4 * k[9]);
    ty = ty + (r2 * k[10] + r4 * k[11]);
  }
  ox = x * rad + tx;
  oy = y * rad + ty;
}

// projection of a camera-frame point by a constant camera of model >= 2: out (2), J (2x3 w.r.t. Xc)
template <bool JAC>
__device__ void project_generic(int model, const double *par, const double *Xc, double *out, double *J) {
  int proj, kind, nd, na;
  model_layout(model, proj, kind, nd, na);
  const double *kd = par + (proj == 2 ? 1 : 0), *ka = kd + nd;
  double u, v, jp[6];
  if (proj == 2) {  // DualProjection: t * perspective + (1 - t) * fisheye (camera_projections_functions.h:122-134)
    double ua, va, ja[6], ub, vb, jb[6];
    project_stage<true>(OSFM_CAMERA_PERSPECTIVE, Xc, ua, va, ja);
    project_stage<true>(OSFM_CAMERA_FISHEYE, Xc, ub, vb, jb);
    const double t = par[0];
    u = t * ua + (1.0 - t) * ub;
    v = t * va + (1.0 - t) * vb;
    for (int i = 0; i < 6; i++) jp[i] = t * ja[i] + (1.0 - t) * jb[i];
  } else {
    project_stage<true>(proj, Xc, u, v, jp);
  }
  D2 dx, dy;
  distort_d2(kind, kd, D2{u, 1.0, 0.0}, D2{v, 0.0, 1.0}, dx, dy);
  const double fx = ka[0], fy = na == 4 ? ka[0] * ka[1] : ka[0];
  const double cx = na == 4 ? ka[2] : 0.0, cy = na == 4 ? ka[3] : 0.0;
  out[0] = fx * dx.v + cx;
  out[1] = fy * dy.v + cy;
  if (JAC)
    for (int j = 0; j < 3; j++) {
      J[j] = fx * (dx.a * jp[j] + dx.b * jp[3 + j]);
      J[3 + j] = fy * (dy.a * jp[j] + dy.b * jp[3 + j]);
    }
}


// number of native parameters of a camera model ([PROJ][DISTO][AFF], camera_instances.h:127-160); SPHERICAL has none to optimise
__device__ __host__ inline int model_num_params(int model) {
  switch (model) {
    case OSFM_CAMERA_PERSPECTIVE:
    case OSFM_CAMERA_FISHEYE: return 3;
    case OSFM_CAMERA_BROWN: return 9;
    case OSFM_CAMERA_FISHEYE_OPENCV: return 8;
    case OSFM_CAMERA_FISHEYE62: return 12;
    case OSFM_CAMERA_FISHEYE624: return 16;
    case OSFM_CAMERA_DUAL: return 4;
    case OSFM_CAMERA_RADIAL: return 6;
    case OSFM_CAMERA_SIMPLE_RADIAL: return 5;
    default: return 0;
  }
}

// Projection of a camera-frame point by ANY 2-D model with every derivative the bundle adjustment needs: out (2), JX (2 x 3 w.r.t.
// Xc) and JK (2 x 16 row-major w.r.t. the native parameters, unused columns zero) -- the ForwardDerivatives<T, true> of
// ProjectGeneric<PROJ, DISTO, AFF> (camera_instances.h:127-160; parameter columns: camera_projections_functions.h:137-171 for the
// dual transition, the distortion classes of camera_distortions_functions.h, transformations_functions.h:22-40,60-72 for the
// affine / uniform scale).
template <bool JAC>
__device__ void project_full(int model, const double *par, const double *Xc, double *out, double *JX, double *JK) {
  int proj, kind, nd, na;
  model_layout(model, proj, kind, nd, na);
  const int np0 = proj == 2 ? 1 : 0;
  const double *kd = par + np0, *ka = kd + nd;
  double u, v, jp[6], dtu = 0.0, dtv = 0.0;
  if (proj == 2) {
    double ua, va, ja[6], ub, vb, jb[6];
    project_stage<true>(OSFM_CAMERA_PERSPECTIVE, Xc, ua, va, ja);
    project_stage<true>(OSFM_CAMERA_FISHEYE, Xc, ub, vb, jb);
    const double t = par[0];
    u = t * ua + (1.0 - t) * ub;
    v = t * va + (1.0 - t) * vb;
    for (int i = 0; i < 6; i++) jp[i] = t * ja[i] + (1.0 - t) * jb[i];
    dtu = ua - ub;
    dtv = va - vb;
  } else {
    project_stage<true>(proj, Xc, u, v, jp);
  }
  D2 dx, dy;
  distort_d2(kind, kd, D2{u, 1.0, 0.0}, D2{v, 0.0, 1.0}, dx, dy);
  const double fx = ka[0], fy = na == 4 ? ka[0] * ka[1] : ka[0];
  const double cx = na == 4 ? ka[2] : 0.0, cy = na == 4 ? ka[3] : 0.0;
  out[0] = fx * dx.v + cx;
  out[1] = fy * dy.v + cy;
  if (!JAC) return;
  for (int j = 0; j < 3; j++) {
    JX[j] = fx * (dx.a * jp[j] + dx.b * jp[3 + j]);
    JX[3 + j] = fy * (dy.a * jp[j] + dy.b * jp[3 + j]);
  }
  for (int i = 0; i < 32; i++) JK[i] = 0.0;
  if (proj == 2) {  // transition: through the distortion's 2 x 2 Jacobian
    JK[0] = fx * (dx.a * dtu + dx.b * dtv);
    JK[16] = fy * (dy.a * dtu + dy.b * dtv);
  }
  const double r2 = u * u + v * v;
  const int nrad = kind == 0 ? 1 : kind == 1 ? 2 : kind == 2 ? 4 : kind == 3 ? 3 : 6;
  double pw = r2;
  for (int i = 0; i < nrad; i++) {  // radial coefficient i multiplies r2^(i + 1) (x, y)
    JK[np0 + i] = fx * u * pw;
    JK[16 + np0 + i] = fy * v * pw;
    pw *= r2;
  }
  if (kind >= 3) {  // tangential p1, p2
    const int ip = np0 + (kind == 3 ? 3 : 6);
    JK[ip] = fx * (2.0 * u * v);
    JK[16 + ip] = fy * (r2 + 2.0 * v * v);
    JK[ip + 1] = fx * (r2 + 2.0 * u * u);
    JK[16 + ip + 1] = fy * (2.0 * u * v);
  }
  if (kind == 5) {  // thin prism s0 .. s3
    JK[np0 + 8] = fx * r2;
    JK[np0 + 9] = fx * r2 * r2;
    JK[16 + np0 + 10] = fy * r2;
    JK[16 + np0 + 11] = fy * r2 * r2;
  }
  const int ia = np0 + nd;
  JK[ia] = dx.v;  // focal
  JK[16 + ia] = (na == 4 ? ka[1] : 1.0) * dy.v;
  if (na == 4) {
    JK[16 + ia + 1] = ka[0] * dy.v;  // aspect ratio
    JK[ia + 2] = 1.0;                // cx
    JK[16 + ia + 3] = 1.0;           // cy
  }
}

}  // namespace osfm_ba
