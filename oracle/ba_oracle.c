/*
 * ba_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never on the product path).
 *
 * CPU restatement of OpenSfM's global bundle adjustment hot path:
 *
 *   opensfm/reconstruction.py:69-86          bundle()
 *   opensfm/src/sfm/src/ba_helpers.cc:581-763  BAHelpers::Bundle   (what is added to the adjuster)
 *   opensfm/src/bundle/src/bundle_adjuster.cc:595-1121 BundleAdjuster::Run (the ceres::Problem)
 *   opensfm/src/bundle/error/projection_errors.h:59-208  ReprojectionError2DAnalytic<C>::Evaluate
 *   opensfm/src/geometry/transformations_functions.h:107-263  PoseFunctor (angle-axis cam->world
 *        rotation r, camera ORIGIN t:  Xc = R(-r) (X - t))
 *   opensfm/src/geometry/camera_projections_functions.h:88-117  PerspectiveProjection
 *   opensfm/src/geometry/camera_distortions_functions.h:103-202 Disto24
 *   opensfm/src/geometry/transformations_functions.h:51-79      UniformScale
 *   opensfm/src/bundle/error/prior_error.h:19-114 + bundle_adjuster.cc:568-593: camera prior,
 *        linear for k1,k2 and log(f/f0) for the focal;  bundle_adjuster.cc:745-778: position prior
 *
 * The optimiser itself is Ceres (ceres-solver 2.1, conda.yml:10), NOT under /root/reference.
 * Its published trust-region Levenberg-Marquardt is restated here with the defaults the reference
 * leaves untouched (bundle_adjuster.cc:1104-1113 sets only solver type, threads, max iterations):
 *   initial_trust_region_radius 1e4, max 1e16, min 1e-32, min_relative_decrease 1e-3,
 *   min/max_lm_diagonal 1e-6/1e32, jacobi_scaling (1/(1+||col||), computed once),
 *   function_tolerance 1e-6, gradient_tolerance 1e-10, parameter_tolerance 1e-8,
 *   cost = 1/2 sum rho(||r||^2), robust loss through the corrector (rho'' <= 0: scale r and J by
 *   sqrt(rho')), SPARSE_SCHUR = exact elimination of the point blocks + direct factorisation of
 *   the reduced camera system (here: skyline Cholesky).
 * PARITY STATUS: residual/Jacobian maths is pinned by the reference's own test vectors
 * (bundle/test/reprojection_errors_test.cc, geometry/test/camera_functions_test.cc) via
 * tests/golden/reprojection_golden.json; the LM trajectory is "parity unpinned" against real Ceres
 * (absent), the reference only pins BA results loosely (test_bundle.py:116-165).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int32_t n_cameras, n_shots, n_points;
  int64_t n_obs;
  double *cam_params;       /* n_cameras x 3: k1, k2, focal (camera.cc:9-17) -- in/out */
  const double *cam_prior;  /* n_cameras x 3 */
  const double *cam_sigma;  /* n_cameras x 3: sd of k1, k2, focal(log) */
  const uint8_t *cam_fixed; /* n_cameras */
  double *shot_pose;        /* n_shots x 6: rx ry rz tx ty tz -- in/out */
  const int32_t *shot_camera;
  const uint8_t *shot_fixed;     /* may be NULL */
  const double *shot_gps;        /* n_shots x 3 or NULL */
  const double *shot_gps_sigma;  /* n_shots (<= 0: no prior) or NULL */
  double *points;                /* n_points x 3 -- in/out */
  const uint8_t *point_fixed;    /* may be NULL */
  const int32_t *obs_shot, *obs_point;
  const double *obs_xy;    /* n_obs x 2 */
  const double *obs_sigma; /* n_obs */
  double *reproj_err;      /* n_obs x 2 out (sigma = 1), may be NULL */
  /* absolute up-vector prior (bundle_adjuster.cc:955-970, absolute_motion_errors.h:12-39):
     r = (R(rot) * up - e_z) / sd, CauchyLoss(1) per shot; NULL or sd <= 0: none */
  const double *shot_up;       /* n_shots x 3 (normalised here) or NULL */
  const double *shot_up_sigma; /* n_shots or NULL */
  const int32_t *cam_model;    /* n_cameras or NULL: 0 PERSPECTIVE, 1 FISHEYE (camera_instances.h:183-190); 2.. see below */
  /* models 2..8 (BROWN, FISHEYE_OPENCV, FISHEYE62, FISHEYE624, DUAL, RADIAL, SIMPLE_RADIAL): CONSTANT cameras only
     (cam_fixed = 1, as BundleLocal / BundleShotPoses always have them, ba_helpers.cc:137,415); their parameters in
     the reference's native order [projection][distortion][affine] (camera_instances.h:100-160), 16 per camera */
  const double *cam_ext;       /* n_cameras x 16 or NULL */
} ba_problem;

typedef struct {
  int32_t loss;          /* 0 Trivial, 1 SoftLOne, 2 Huber, 3 Cauchy */
  double loss_threshold; /* a */
  int32_t max_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_radius;
  int32_t verbose;
} ba_options;

typedef struct {
  int32_t iterations;        /* LM iterations performed (successful + unsuccessful) */
  int32_t successful_steps;
  int32_t termination;       /* 0 max iters, 1 function tol, 2 gradient tol, 3 parameter tol, -1 failure */
  double initial_cost, final_cost;
  double rmse_normalized_initial, rmse_normalized_final; /* sqrt(mean |pi - o|^2) over observations */
  double seconds_total, seconds_linear_solver;
  double cost_history[256];
} ba_report;

/* ---------------------------------------------------------------------------------------- */
static void rot_and_derivs(const double *r, double R[9], double dR[3][9]) {
  /* R = R(a), a = -r ; dR[k] = d R / d a_k.  Closed form of what the reference obtains by
   * forward-mode autodiff of AngleAxisToRotation (transformations_functions.h:151-173,216-262). */
  const double a[3] = {-r[0], -r[1], -r[2]};
  const double th2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
  if (!(th2 > 2.220446049250313e-16)) {
    for (int i = 0; i < 9; i++) R[i] = K[i];
    R[0] = R[4] = R[8] = 1.0;
    for (int k = 0; k < 3; k++) {
      for (int i = 0; i < 9; i++) dR[k][i] = 0;
    }
    dR[0][5] = -1; dR[0][7] = 1;
    dR[1][2] = 1;  dR[1][6] = -1;
    dR[2][1] = -1; dR[2][3] = 1;
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
    double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* [e_k]x */
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
        dR[k][3 * i + j] = a[k] * Ap * K[3 * i + j] + A * E[3 * i + j] + a[k] * Bp * K2[3 * i + j] + B * (ek_k + k_ek);
      }
  }
}

/* One observation: residual (pi - o)/sigma and Jacobians 2x3 (point), 2x6 (pose), 2x3 (k1,k2,f),
 * already multiplied by 1/sigma (projection_errors.h:150-205).  R, dR: of the shot. */
/* PROJ stage (camera_projections_functions.h): PerspectiveProjection :88-117 or FisheyeProjection :9-85
 * (theta / r * (x, y), theta = atan2(r, z), r = |(x, y)|; perspective below r = 1e-8).  uv (2), jp (2x3). */
static void project_stage(int model, const double *Xc, double *uv, double *jp) {
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  const double r2 = x * x + y * y, r = sqrt(r2);
  if (model == 1 && !(r < 1e-8)) {
    const double theta = atan2(r, z);
    uv[0] = theta / r * x;
    uv[1] = theta / r * y;
    const double R2 = r2 + z * z, x2 = x * x, y2 = y * y, z2 = z * z;
    const double inv_denom = 1.0 / (r2 * R2 * r);
    jp[0] = (x2 * y2 * theta + y2 * y2 * theta + y2 * z2 * theta + x2 * z * r) * inv_denom;
    jp[1] = x * (y * z * r - y * theta * R2) * inv_denom;
    jp[2] = -x / R2;
    jp[3] = y * (x * z * r - x * theta * R2) * inv_denom;
    jp[4] = (x2 * y2 * theta + x2 * x2 * theta + x2 * z2 * theta + y2 * z * r) * inv_denom;
    jp[5] = -y / R2;
    return;
  }
  const double iz = 1.0 / z;
  uv[0] = x * iz;
  uv[1] = y * iz;
  jp[0] = iz; jp[1] = 0.0; jp[2] = -x * iz * iz;
  jp[3] = 0.0; jp[4] = iz; jp[5] = -y * iz * iz;
}

/* ---- the other 2-D camera models, for CONSTANT cameras (no intrinsics Jacobian needed) ----------------
 * ProjectGeneric<PROJ, DISTO, AFF> (camera_instances.h:127-160); parameters [PROJ][DISTO][AFF].
 * The 2x2 Jacobian of the distortion stage is taken by forward-mode duals (the reference's autodiff twin of
 * the same formulas, camera_distortions_functions.h), the projection stage analytically as above. */
typedef struct { double v, a, b; } d2; /* value, d/dx, d/dy */
static inline d2 d2c(double c) { d2 r = {c, 0, 0}; return r; }
static inline d2 d2add(d2 p, d2 q) { d2 r = {p.v + q.v, p.a + q.a, p.b + q.b}; return r; }
static inline d2 d2mul(d2 p, d2 q) { d2 r = {p.v * q.v, p.a * q.v + p.v * q.a, p.b * q.v + p.v * q.b}; return r; }
static inline d2 d2scale(d2 p, double c) { d2 r = {p.v * c, p.a * c, p.b * c}; return r; }

/* model -> (projection: 0 perspective / 1 fisheye / 2 dual, #distortion params and kind, affine: 1 uniform / 4 affine) */
static void model_layout(int model, int *proj, int *kind, int *nd, int *na) {
  switch (model) {
    case 2: *proj = 0; *kind = 3; *nd = 5;  *na = 4; break; /* BROWN: DistoBrown k1 k2 k3 p1 p2 */
    case 3: *proj = 1; *kind = 2; *nd = 4;  *na = 4; break; /* FISHEYE_OPENCV: Disto2468 */
    case 4: *proj = 1; *kind = 4; *nd = 8;  *na = 4; break; /* FISHEYE62: Disto62 k1..k6 p1 p2 */
    case 5: *proj = 1; *kind = 5; *nd = 12; *na = 4; break; /* FISHEYE624: + s0..s3 */
    case 6: *proj = 2; *kind = 1; *nd = 2;  *na = 1; break; /* DUAL: transition | Disto24 | focal */
    case 7: *proj = 0; *kind = 1; *nd = 2;  *na = 4; break; /* RADIAL: Disto24 | Affine */
    default: *proj = 0; *kind = 0; *nd = 1; *na = 4; break; /* 8 SIMPLE_RADIAL: Disto2 | Affine */
  }
}

static void distort_d2(int kind, const double *k, d2 x, d2 y, d2 *ox, d2 *oy) {
  const d2 r2 = d2add(d2mul(x, x), d2mul(y, y));
  d2 rad, tx = d2c(0), ty = d2c(0);
  switch (kind) {
    case 0: rad = d2add(d2c(1.0), d2scale(r2, k[0])); break;
    case 1: rad = d2add(d2c(1.0), d2mul(r2, d2add(d2c(k[0]), d2scale(r2, k[1])))); break;
    case 2: rad = d2add(d2c(1.0), d2mul(r2, d2add(d2c(k[0]), d2mul(r2, d2add(d2c(k[1]), d2mul(r2, d2add(d2c(k[2]), d2scale(r2, k[3])))))))); break;
    case 3: rad = d2add(d2c(1.0), d2mul(r2, d2add(d2c(k[0]), d2mul(r2, d2add(d2c(k[1]), d2scale(r2, k[2])))))); break;
    default: { /* 4, 5: six radial coefficients */
      d2 acc = d2add(d2c(k[4]), d2scale(r2, k[5])); /* Horner, as RadialDistortion (camera_distortions_functions.h:481-485) */
      acc = d2add(d2c(k[3]), d2mul(r2, acc));
      acc = d2add(d2c(k[2]), d2mul(r2, acc));
      acc = d2add(d2c(k[1]), d2mul(r2, acc));
      acc = d2add(d2c(k[0]), d2mul(r2, acc));
      rad = d2add(d2c(1.0), d2mul(r2, acc));
    } break;
  }
  if (kind >= 3) { /* tangential: 2 p1 x y + p2 (r2 + 2 x^2), 2 p2 x y + p1 (r2 + 2 y^2) */
    const double p1 = kind == 3 ? k[3] : k[6], p2 = kind == 3 ? k[4] : k[7];
    const d2 xy = d2mul(x, y);
    tx = d2add(d2scale(xy, 2.0 * p1), d2scale(d2add(r2, d2scale(d2mul(x, x), 2.0)), p2));
    ty = d2add(d2scale(xy, 2.0 * p2), d2scale(d2add(r2, d2scale(d2mul(y, y), 2.0)), p1));
  }
  if (kind == 5) { /* thin prism: s0 r2 + s1 r2^2, s2 r2 + s3 r2^2 */
    const d2 r4 = d2mul(r2, r2);
    tx = d2add(tx, d2add(d2scale(r2, k[8]), d2scale(r4, k[9])));
    ty = d2add(ty, d2add(d2scale(r2, k[10]), d2scale(r4, k[11])));
  }
  *ox = d2add(d2mul(x, rad), tx);
  *oy = d2add(d2mul(y, rad), ty);
}

/* projection of a camera-frame point by a constant camera of model 2..8: out (2), J (2x3 w.r.t. Xc) */
static void project_generic(int model, const double *par, const double *Xc, double *out, double *J) {
  int proj, kind, nd, na;
  model_layout(model, &proj, &kind, &nd, &na);
  const double *kd = par + (proj == 2 ? 1 : 0), *ka = kd + nd;
  double uv[2], jp[6];
  if (proj == 2) { /* DualProjection: t * perspective + (1 - t) * fisheye */
    double a[2], ja[6], b[2], jb[6];
    project_stage(0, Xc, a, ja);
    project_stage(1, Xc, b, jb);
    const double t = par[0];
    for (int i = 0; i < 2; i++) uv[i] = t * a[i] + (1.0 - t) * b[i];
    for (int i = 0; i < 6; i++) jp[i] = t * ja[i] + (1.0 - t) * jb[i];
  } else {
    project_stage(proj, Xc, uv, jp);
  }
  d2 x = {uv[0], 1, 0}, y = {uv[1], 0, 1}, dx, dy;
  distort_d2(kind, kd, x, y, &dx, &dy);
  const double fx = ka[0], fy = na == 4 ? ka[0] * ka[1] : ka[0];
  const double cx = na == 4 ? ka[2] : 0.0, cy = na == 4 ? ka[3] : 0.0;
  out[0] = fx * dx.v + cx;
  out[1] = fy * dy.v + cy;
  for (int j = 0; j < 3; j++) {
    J[j] = fx * (dx.a * jp[j] + dx.b * jp[3 + j]);
    J[3 + j] = fy * (dy.a * jp[j] + dy.b * jp[3 + j]);
  }
}

static void project_obs(int model, const double *X, const double *pose, const double *R, const double (*dR)[9],
                        const double *cam, const double *obs, double inv_sigma, double *res,
                        double *Jp, double *Jc, double *Jk) {
  const double p[3] = {X[0] - pose[3], X[1] - pose[4], X[2] - pose[5]};
  double Xc[3];
  for (int i = 0; i < 3; i++) Xc[i] = R[3 * i] * p[0] + R[3 * i + 1] * p[1] + R[3 * i + 2] * p[2];
  if (model >= 2) { /* constant camera of another 2-D model: cam points at its 16 native parameters */
    double out[2], J[6], M[6];
    project_generic(model, cam, Xc, out, J);
    res[0] = inv_sigma * (out[0] - obs[0]);
    res[1] = inv_sigma * (out[1] - obs[1]);
    if (!Jp) return;
    for (int j = 0; j < 6; j++) M[j] = inv_sigma * J[j];
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) {
        const double mr = M[3 * i] * R[j] + M[3 * i + 1] * R[3 + j] + M[3 * i + 2] * R[6 + j];
        Jp[3 * i + j] = mr;
        Jc[6 * i + 3 + j] = -mr;
      }
    for (int k = 0; k < 3; k++) {
      double q[3];
      for (int i = 0; i < 3; i++) q[i] = dR[k][3 * i] * p[0] + dR[k][3 * i + 1] * p[1] + dR[k][3 * i + 2] * p[2];
      for (int i = 0; i < 2; i++) Jc[6 * i + k] = -(M[3 * i] * q[0] + M[3 * i + 1] * q[1] + M[3 * i + 2] * q[2]);
    }
    for (int i = 0; i < 6; i++) Jk[i] = 0.0;
    return;
  }
  const double k1 = cam[0], k2 = cam[1], f = cam[2];
  double uv[2], jp[6];
  project_stage(model, Xc, uv, jp);
  const double u = uv[0], v = uv[1];
  const double r2 = u * u + v * v;
  const double d = 1.0 + r2 * (k1 + k2 * r2);
  res[0] = inv_sigma * (f * d * u - obs[0]);
  res[1] = inv_sigma * (f * d * v - obs[1]);
  if (!Jp) return;
  /* d pi / d Xc = f * Jd * Jproj */
  const double x2 = u * u, y2 = v * v, x4 = x2 * x2, y4 = y2 * y2;
  const double jd00 = 5.0 * k2 * x4 + 3.0 * k1 * x2 + 6.0 * k2 * x2 * y2 + k2 * y4 + k1 * y2 + 1.0;
  const double jd01 = u * (2.0 * k1 * v + 4.0 * k2 * v * r2);
  const double jd10 = v * (2.0 * k1 * u + 4.0 * k2 * u * r2);
  const double jd11 = 5.0 * k2 * y4 + 3.0 * k1 * y2 + 6.0 * k2 * y2 * x2 + k2 * x4 + k1 * x2 + 1.0;
  double M[6]; /* 2x3 = f * Jd * Jproj * inv_sigma */
  for (int j = 0; j < 3; j++) {
    M[j] = inv_sigma * f * (jd00 * jp[j] + jd01 * jp[3 + j]);
    M[3 + j] = inv_sigma * f * (jd10 * jp[j] + jd11 * jp[3 + j]);
  }
  /* point: M * R ; origin t: -M * R ; rotation r_k: M * d(R(a) p)/d r_k = -M * dR[k] p */
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) {
      const double mr = M[3 * i] * R[j] + M[3 * i + 1] * R[3 + j] + M[3 * i + 2] * R[6 + j];
      Jp[3 * i + j] = mr;
      Jc[6 * i + 3 + j] = -mr;
    }
  for (int k = 0; k < 3; k++) {
    double q[3];
    for (int i = 0; i < 3; i++) q[i] = dR[k][3 * i] * p[0] + dR[k][3 * i + 1] * p[1] + dR[k][3 * i + 2] * p[2];
    for (int i = 0; i < 2; i++) Jc[6 * i + k] = -(M[3 * i] * q[0] + M[3 * i + 1] * q[1] + M[3 * i + 2] * q[2]);
  }
  Jk[0] = inv_sigma * f * r2 * u;
  Jk[1] = inv_sigma * f * r2 * r2 * u;
  Jk[2] = inv_sigma * d * u;
  Jk[3] = inv_sigma * f * r2 * v;
  Jk[4] = inv_sigma * f * r2 * r2 * v;
  Jk[5] = inv_sigma * d * v;
}

/* rho(s), rho'(s) for the losses of bundle_adjuster.cc:414-429 (ceres/loss_function.cc). */
static void loss_eval(int loss, double a, double s, double *rho, double *rho1) {
  const double b = a * a;
  switch (loss) {
    case 1: { /* SoftLOne */
      const double sum = 1.0 + s / b, tmp = sqrt(sum);
      *rho = 2.0 * b * (tmp - 1.0);
      *rho1 = 1.0 / tmp;
    } break;
    case 2: /* Huber */
      if (s > b) {
        const double r = sqrt(s);
        *rho = 2.0 * a * r - b;
        *rho1 = a / r;
      } else {
        *rho = s;
        *rho1 = 1.0;
      }
      break;
    case 3: { /* Cauchy */
      const double sum = 1.0 + s / b;
      *rho = b * log(sum);
      *rho1 = 1.0 / sum;
    } break;
    default:
      *rho = s;
      *rho1 = 1.0;
  }
}

/* exposed for the golden-vector tests */
void oracle_ba_project(const double *X, const double *pose, const double *cam, const double *obs,
                       double sigma, double *res, double *Jp, double *Jc, double *Jk, int model) {
  double R[9], dR[3][9];
  rot_and_derivs(pose, R, dR);
  project_obs(model, X, pose, R, (const double (*)[9])dR, cam, obs, 1.0 / sigma, res, Jp, Jc, Jk);
}
void oracle_ba_loss(int loss, double a, double s, double *out2) { loss_eval(loss, a, s, &out2[0], &out2[1]); }

/* GROUNDWORK (no product counterpart yet: the GPU solver keeps the 4-16 parameter cameras constant): the Jacobian of the
 * reprojection residual w.r.t. EVERY intrinsic parameter of a camera of model 2..8, in the native order
 * [projection][distortion][affine] -- what ForwardDerivatives<T, DERIV_PARAMS = true> composes in the reference
 * (camera_projections_functions.h:137-171 dual transition; camera_distortions_functions.h: the parameter columns of each
 * distortion's ForwardDerivatives; transformations_functions.h:22-40 Affine, :60-72 UniformScale).  Jk: 2 x 16 row-major
 * (unused columns zero), already multiplied by 1 / sigma.  Pinned by the mpmath golden vectors (tests/test_oracle_ba.py). */
void oracle_ba_project_intrinsics(int model, const double *X, const double *pose, const double *par, const double *obs, double sigma,
                                  double *res, double *Jk) {
  double R[9], dR[3][9];
  rot_and_derivs(pose, R, dR);
  const double p[3] = {X[0] - pose[3], X[1] - pose[4], X[2] - pose[5]};
  double Xc[3];
  for (int i = 0; i < 3; i++) Xc[i] = R[3 * i] * p[0] + R[3 * i + 1] * p[1] + R[3 * i + 2] * p[2];
  int proj, kind, nd, na;
  model_layout(model, &proj, &kind, &nd, &na);
  const int np0 = proj == 2 ? 1 : 0;
  const double *kd = par + np0, *ka = kd + nd;
  double uv[2], jp[6], dt[2] = {0, 0};
  if (proj == 2) {
    double a[2], ja[6], b[2], jb[6];
    project_stage(0, Xc, a, ja);
    project_stage(1, Xc, b, jb);
    for (int i = 0; i < 2; i++) {
      uv[i] = par[0] * a[i] + (1.0 - par[0]) * b[i];
      dt[i] = a[i] - b[i]; /* d uv / d transition */
    }
  } else {
    project_stage(proj, Xc, uv, jp);
  }
  d2 x = {uv[0], 1, 0}, y = {uv[1], 0, 1}, dx, dy;
  distort_d2(kind, kd, x, y, &dx, &dy);
  const double u = uv[0], v = uv[1], r2 = u * u + v * v;
  const double fx = ka[0], fy = na == 4 ? ka[0] * ka[1] : ka[0];
  const double cx = na == 4 ? ka[2] : 0.0, cy = na == 4 ? ka[3] : 0.0;
  const double is = 1.0 / sigma;
  res[0] = is * (fx * dx.v + cx - obs[0]);
  res[1] = is * (fy * dy.v + cy - obs[1]);
  for (int i = 0; i < 32; i++) Jk[i] = 0.0;
#define OSFM_SET(col, a, b) { Jk[(col)] = is * fx * (a); Jk[16 + (col)] = is * fy * (b); }
  if (proj == 2) OSFM_SET(0, dx.a * dt[0] + dx.b * dt[1], dy.a * dt[0] + dy.b * dt[1]);
  /* distortion parameters: derivatives of (du, dv) */
  const int nrad = kind == 0 ? 1 : kind == 1 ? 2 : kind == 2 ? 4 : kind == 3 ? 3 : 6;
  double pw = r2;
  for (int i = 0; i < nrad; i++) { /* radial coefficient k_{i+1} multiplies r2^(i+1) */
    OSFM_SET(np0 + i, u * pw, v * pw);
    pw *= r2;
  }
  if (kind >= 3) {
    const int ip = np0 + (kind == 3 ? 3 : 6);
    OSFM_SET(ip, 2.0 * u * v, r2 + 2.0 * v * v);     /* p1 */
    OSFM_SET(ip + 1, r2 + 2.0 * u * u, 2.0 * u * v); /* p2 */
  }
  if (kind == 5) {
    OSFM_SET(np0 + 8, r2, 0.0);
    OSFM_SET(np0 + 9, r2 * r2, 0.0);
    OSFM_SET(np0 + 10, 0.0, r2);
    OSFM_SET(np0 + 11, 0.0, r2 * r2);
  }
#undef OSFM_SET
  /* affine parameters */
  const int ia = np0 + nd;
  Jk[ia] = is * dx.v;                          /* focal */
  Jk[16 + ia] = is * (na == 4 ? ka[1] : 1.0) * dy.v;
  if (na == 4) {
    Jk[16 + ia + 1] = is * ka[0] * dy.v;       /* aspect ratio */
    Jk[ia + 2] = is;                           /* cx */
    Jk[16 + ia + 3] = is;                      /* cy */
  }
}

/* ---------------------------------------------------------------------------------------- */
typedef struct {
  const ba_problem *P;
  const ba_options *O;
  int ncv;        /* variable cameras */
  int nsv;        /* variable shots */
  int *cam_var;   /* camera -> reduced index or -1 */
  int *shot_var;  /* shot   -> reduced index or -1 */
  int nred;       /* reduced unknowns: 6*nsv + 3*ncv (shots first, cameras last) */
  /* per observation storage */
  double *res;    /* 2 (corrected) */
  double *Jp, *Jc, *Jk; /* 6, 12, 6 (corrected, UNscaled) */
  double *up_r, *up_J;  /* per shot: corrected up-vector residual (3) and its 3x3 Jacobian w.r.t. the rotation */
  /* point-major observation lists */
  int64_t *pt_off; int64_t *pt_obs;
} ba_ctx;

/* up-vector residual of one shot: r = (Rw^T u - e_z)/sd with Rw = R(-rot) (world -> camera), i.e. R(rot) u;
 * J[i][k] = d r_i / d rot_k = -(dRw_k^T u)_i / sd  (dRw_k = d Rw / d a_k, a = -rot) */
static int up_residual(const ba_problem *P, int s, const double *R, const double (*dR)[9], double r[3], double J[9]) {
  if (!(P->shot_up && P->shot_up_sigma) || !(P->shot_up_sigma[s] > 0)) return 0;
  const double *u0 = P->shot_up + 3 * (size_t)s;
  const double nrm = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
  const double u[3] = {u0[0] / nrm, u0[1] / nrm, u0[2] / nrm};
  const double isd = 1.0 / P->shot_up_sigma[s];
  for (int i = 0; i < 3; i++) {
    const double z = R[i] * u[0] + R[3 + i] * u[1] + R[6 + i] * u[2]; /* (Rw^T u)_i */
    r[i] = isd * (z - (i == 2 ? 1.0 : 0.0));
    if (J)
      for (int k = 0; k < 3; k++) J[3 * i + k] = -isd * (dR[k][i] * u[0] + dR[k][3 + i] * u[1] + dR[k][6 + i] * u[2]);
  }
  return 1;
}

/* exposed for the tests: up-vector residual and Jacobian of one shot */
void oracle_ba_up(const double *pose, const double *up, double sigma, double *r, double *J) {
  double R[9], dR[3][9];
  rot_and_derivs(pose, R, dR);
  ba_problem P;
  memset(&P, 0, sizeof(P));
  P.shot_up = up;
  P.shot_up_sigma = &sigma;
  up_residual(&P, 0, R, (const double (*)[9])dR, r, J);
}

static double eval_cost(const ba_ctx *C, const double *cams, const double *poses, const double *pts,
                        int with_jac, double *sumsq_out) {
  const ba_problem *P = C->P;
  double *Rall = (double *)malloc(sizeof(double) * 36 * (size_t)P->n_shots);
#pragma omp parallel for schedule(static)
  for (int s = 0; s < P->n_shots; s++) rot_and_derivs(poses + 6 * s, Rall + 36 * (size_t)s, (double (*)[9])(Rall + 36 * (size_t)s + 9));
  double cost = 0.0, sumsq = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : cost, sumsq)
  for (int64_t o = 0; o < P->n_obs; o++) {
    const int s = P->obs_shot[o], p = P->obs_point[o];
    const double *R = Rall + 36 * (size_t)s;
    double r[2], Jp[6], Jc[12], Jk[6];
    const double isg = 1.0 / P->obs_sigma[o];
    const int cmodel = P->cam_model ? P->cam_model[P->shot_camera[s]] : 0;
    project_obs(cmodel, pts + 3 * (size_t)p, poses + 6 * (size_t)s, R,
                (const double (*)[9])(R + 9), cmodel >= 2 ? P->cam_ext + 16 * (size_t)P->shot_camera[s] : cams + 3 * (size_t)P->shot_camera[s], P->obs_xy + 2 * o, isg, r, with_jac ? Jp : NULL, Jc, Jk);
    const double sq = r[0] * r[0] + r[1] * r[1];
    double rho, rho1;
    loss_eval(C->O->loss, C->O->loss_threshold, sq, &rho, &rho1);
    cost += 0.5 * rho;
    sumsq += sq * P->obs_sigma[o] * P->obs_sigma[o];
    if (with_jac) {
      const double w = sqrt(rho1); /* corrector, rho'' <= 0 */
      C->res[2 * o] = w * r[0];
      C->res[2 * o + 1] = w * r[1];
      for (int i = 0; i < 6; i++) C->Jp[6 * o + i] = w * Jp[i];
      for (int i = 0; i < 12; i++) C->Jc[12 * o + i] = w * Jc[i];
      for (int i = 0; i < 6; i++) C->Jk[6 * o + i] = w * Jk[i];
    }
  }
  /* camera priors (bundle_adjuster.cc:568-593): (k - k0)/s, (k2 - k20)/s, log(f/f0)/s ; no loss */
  for (int c = 0; c < P->n_cameras; c++) {
    if (P->cam_fixed[c]) continue;
    const double *v = cams + 3 * c, *pr = P->cam_prior + 3 * c, *sg = P->cam_sigma + 3 * c;
    const double e0 = (v[0] - pr[0]) / fmax(sg[0], 2.220446049250313e-16);
    const double e1 = (v[1] - pr[1]) / fmax(sg[1], 2.220446049250313e-16);
    const double e2 = log(v[2] / pr[2]) / fmax(sg[2], 2.220446049250313e-16);
    cost += 0.5 * (e0 * e0 + e1 * e1 + e2 * e2);
  }
  /* position priors (bundle_adjuster.cc:745-778, identity bias, unit scale) */
  if (P->shot_gps && P->shot_gps_sigma)
    for (int s = 0; s < P->n_shots; s++) {
      if (!(P->shot_gps_sigma[s] > 0) || C->shot_var[s] < 0) continue;
      for (int i = 0; i < 3; i++) {
        const double e = (poses[6 * s + 3 + i] - P->shot_gps[3 * s + i]) / P->shot_gps_sigma[s];
        cost += 0.5 * e * e;
      }
    }
  /* up-vector priors: 3 residuals per shot under one CauchyLoss(1) */
  if (P->shot_up && P->shot_up_sigma)
    for (int s = 0; s < P->n_shots; s++) {
      if (C->shot_var[s] < 0) continue;
      double r[3], J[9];
      const double *R = Rall + 36 * (size_t)s;
      if (!up_residual(P, s, R, (const double (*)[9])(R + 9), r, J)) continue;
      const double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      double rho, rho1;
      loss_eval(3, 1.0, sq, &rho, &rho1);
      cost += 0.5 * rho;
      if (with_jac) {
        const double w = sqrt(rho1);
        for (int i = 0; i < 3; i++) C->up_r[3 * s + i] = w * r[i];
        for (int i = 0; i < 9; i++) C->up_J[9 * s + i] = w * J[i];
      }
    }
  free(Rall);
  if (sumsq_out) *sumsq_out = sumsq;
  return cost;
}

/* skyline (row profile) symmetric positive definite solve: A = L L^T, rows stored from first[i] */
typedef struct {
  int n;
  int *first;
  int64_t *off; /* off[i] = start of row i ; row i holds columns first[i]..i */
  double *v;
} skyline;

static int skyline_cholesky(skyline *A) {
  const int n = A->n;
  for (int i = 0; i < n; i++) {
    double *ri = A->v + A->off[i];
    const int fi = A->first[i];
    for (int j = fi; j <= i; j++) {
      const double *rj = A->v + A->off[j];
      const int fj = A->first[j];
      const int k0 = fi > fj ? fi : fj;
      double s = ri[j - fi];
      for (int k = k0; k < j; k++) s -= ri[k - fi] * rj[k - fj];
      if (j < i) {
        ri[j - fi] = s / rj[j - fj];
      } else {
        if (!(s > 0)) return -1;
        ri[i - fi] = sqrt(s);
      }
    }
  }
  return 0;
}
/* The same factor, right-looking and on all cores (round 4): column k is finished, then every row below it subtracts L[i][k] L[j][k] from its
 * entries (j ascending) -- entry (i, j) loses the products of k = max(first[i], first[j]) .. j - 1 one by one in ascending k, from A[i][j], and
 * is divided by L[j][j] when column j comes up: the sequence of operations of skyline_cholesky above, so the factor is the same bit for bit
 * (tests/test_oracle_ba.py compares the two).  Two barriers per column: pays when the rows are wide (a block survey's half-width of ~600
 * unknowns: the serial factor is most of the oracle's time there), not on a 60-wide sequence band.  Opt-in (oracle_ba_set_parallel(2)). */
static int skyline_cholesky_parallel(skyline *A) {
  const int n = A->n;
  /* last row that has an entry in column k (first[] need not be monotone) */
  int *last = (int *)malloc((size_t)n * sizeof(int));
  if (!last) return -1;
  for (int k = 0; k < n; k++) last[k] = k;
  {
    /* last[k] = max row i with first[i] <= k: a running maximum over the rows bucketed by their first column */
    int *best = (int *)calloc((size_t)n + 1, sizeof(int)); /* best[f] = max row with first == f */
    if (!best) {
      free(last);
      return -1;
    }
    for (int f = 0; f <= n; f++) best[f] = -1;
    for (int i = 0; i < n; i++)
      if (i > best[A->first[i]]) best[A->first[i]] = i;
    int run = -1;
    for (int k = 0; k < n; k++) {
      run = best[k] > run ? best[k] : run;
      last[k] = run > k ? run : k;
    }
    free(best);
  }
  /* column k of L, gathered (the rows are stored row by row: reading L[j][k] down a column is a stride of a row length) */
  double *colk = (double *)malloc((size_t)n * sizeof(double));
  if (!colk) {
    free(last);
    return -1;
  }
  int monotone = 1; /* first[] nondecreasing (a band): every row between k and i then reaches column k, and the update is one dense axpy */
  for (int i = 1; i < n; i++)
    if (A->first[i] < A->first[i - 1]) monotone = 0;
  int bad = 0;
  int nt = 1;
#ifdef _OPENMP
  nt = omp_get_max_threads() < 16 ? omp_get_max_threads() : 16; /* two barriers per column: more threads cost more than they bring */
#endif
#pragma omp parallel num_threads(nt)
  {
    for (int k = 0; k < n; k++) {
#pragma omp single
      {
        double *rk = A->v + A->off[k];
        const double s = rk[k - A->first[k]];
        if (!(s > 0)) bad = 1;
        else rk[k - A->first[k]] = sqrt(s);
      } /* implicit barrier */
      if (bad) break;
      const double dkk = A->v[A->off[k] + (k - A->first[k])];
      const int hi = last[k];
#pragma omp for schedule(static)
      for (int i = k + 1; i <= hi; i++) {
        const int fi = A->first[i];
        double l = 0.0;
        if (fi <= k) {
          double *ri = A->v + A->off[i];
          l = ri[k - fi] / dkk;
          ri[k - fi] = l;
        }
        colk[i] = l;
      } /* implicit barrier: column k of L is complete */
#pragma omp for schedule(dynamic, 8)
      for (int i = k + 1; i <= hi; i++) {
        const int fi = A->first[i];
        if (fi > k) continue;
        double *ri = A->v + A->off[i] - fi;
        const double lik = colk[i];
        if (monotone) {
          for (int j = k + 1; j <= i; j++) ri[j] -= lik * colk[j];
        } else {
          for (int j = k + 1; j <= i; j++)
            if (A->first[j] <= k) ri[j] -= lik * colk[j];
        }
      } /* implicit barrier */
    }
  }
  free(colk);
  free(last);
  return bad ? -1 : 0;
}
static void skyline_solve(const skyline *A, double *b) {
  const int n = A->n;
  for (int i = 0; i < n; i++) {
    const double *ri = A->v + A->off[i];
    const int fi = A->first[i];
    double s = b[i];
    for (int k = fi; k < i; k++) s -= ri[k - fi] * b[k];
    b[i] = s / ri[i - fi];
  }
  for (int i = n - 1; i >= 0; i--) {
    const double *ri = A->v + A->off[i];
    const int fi = A->first[i];
    b[i] /= ri[i - fi];
    for (int k = fi; k < i; k++) b[k] -= ri[k - fi] * b[i];
  }
}

static double now_s(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}

/* 1 (default): the Schur elimination runs on all cores (entry-owner partition, see below); 0: the serial loops (kept: they define the
 * summation order the parallel path reproduces, and tests compare the two) */
static int g_ba_parallel = 1;
void oracle_ba_set_parallel(int on) { g_ba_parallel = on; } /* 2: also force the parallel skyline factor (tests) */
void oracle_set_num_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

int oracle_ba_solve(ba_problem *P, const ba_options *O, ba_report *Rp) {
  memset(Rp, 0, sizeof(*Rp));
  const double t_start = now_s();
  ba_ctx C;
  memset(&C, 0, sizeof(C));
  C.P = P;
  C.O = O;
  const int S = P->n_shots, NP = P->n_points, NC = P->n_cameras;
  const int64_t M = P->n_obs;
  C.cam_var = (int *)malloc(sizeof(int) * (size_t)(NC + 1));
  C.shot_var = (int *)malloc(sizeof(int) * (size_t)(S + 1));
  for (int s = 0; s < S; s++) C.shot_var[s] = (P->shot_fixed && P->shot_fixed[s]) ? -1 : C.nsv++;
  for (int c = 0; c < NC; c++) C.cam_var[c] = P->cam_fixed[c] ? -1 : C.ncv++;
  C.nred = 6 * C.nsv + 3 * C.ncv;
  const int nred = C.nred, cam0 = 6 * C.nsv;
  C.res = (double *)malloc(sizeof(double) * 2 * (size_t)M);
  C.Jp = (double *)malloc(sizeof(double) * 6 * (size_t)M);
  C.Jc = (double *)malloc(sizeof(double) * 12 * (size_t)M);
  C.Jk = (double *)malloc(sizeof(double) * 6 * (size_t)M);
  C.up_r = (double *)calloc(3 * (size_t)P->n_shots + 1, sizeof(double));
  C.up_J = (double *)calloc(9 * (size_t)P->n_shots + 1, sizeof(double));
  /* point-major lists */
  C.pt_off = (int64_t *)calloc((size_t)NP + 1, sizeof(int64_t));
  C.pt_obs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(M > 0 ? M : 1));
  for (int64_t o = 0; o < M; o++) C.pt_off[P->obs_point[o] + 1]++;
  for (int p = 0; p < NP; p++) C.pt_off[p + 1] += C.pt_off[p];
  {
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(NP + 1));
    memcpy(fill, C.pt_off, sizeof(int64_t) * (size_t)(NP + 1));
    for (int64_t o = 0; o < M; o++) C.pt_obs[fill[P->obs_point[o]]++] = o;
    free(fill);
  }
  /* skyline profile of the reduced system: shot i couples with the smallest co-visible shot */
  skyline A;
  A.n = nred;
  A.first = (int *)malloc(sizeof(int) * (size_t)(nred + 1));
  A.off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nred + 2));
  {
    int *minshot = (int *)malloc(sizeof(int) * (size_t)(C.nsv + 1));
    for (int i = 0; i < C.nsv; i++) minshot[i] = i;
    for (int p = 0; p < NP; p++) {
      int mn = 1 << 30;
      for (int64_t q = C.pt_off[p]; q < C.pt_off[p + 1]; q++) {
        const int sv = C.shot_var[P->obs_shot[C.pt_obs[q]]];
        if (sv >= 0 && sv < mn) mn = sv;
      }
      for (int64_t q = C.pt_off[p]; q < C.pt_off[p + 1]; q++) {
        const int sv = C.shot_var[P->obs_shot[C.pt_obs[q]]];
        if (sv >= 0 && mn < minshot[sv]) minshot[sv] = mn;
      }
    }
    int64_t tot = 0;
    for (int i = 0; i < C.nsv; i++)
      for (int d = 0; d < 6; d++) {
        A.first[6 * i + d] = 6 * minshot[i];
        A.off[6 * i + d] = tot;
        tot += (6 * i + d) - 6 * minshot[i] + 1;
      }
    for (int r = cam0; r < nred; r++) {
      A.first[r] = 0;
      A.off[r] = tot;
      tot += r + 1;
    }
    A.off[nred] = tot;
    A.v = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    free(minshot);
  }
#define SKY(i, j) A.v[A.off[(i)] + ((j)-A.first[(i)])] /* j <= i */

  double *cams = (double *)malloc(sizeof(double) * 3 * (size_t)(NC + 1));
  double *poses = (double *)malloc(sizeof(double) * 6 * (size_t)(S + 1));
  double *pts = (double *)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
  memcpy(cams, P->cam_params, sizeof(double) * 3 * (size_t)NC);
  memcpy(poses, P->shot_pose, sizeof(double) * 6 * (size_t)S);
  memcpy(pts, P->points, sizeof(double) * 3 * (size_t)NP);
  double *cams_n = (double *)malloc(sizeof(double) * 3 * (size_t)(NC + 1));
  double *poses_n = (double *)malloc(sizeof(double) * 6 * (size_t)(S + 1));
  double *pts_n = (double *)malloc(sizeof(double) * 3 * (size_t)(NP + 1));

  /* scaling (jacobi) and work vectors */
  double *sc_red = (double *)malloc(sizeof(double) * (size_t)(nred + 1));
  double *sc_pt = (double *)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
  double *g_red = (double *)malloc(sizeof(double) * (size_t)(nred + 1));
  double *g_pt = (double *)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
  double *diag_red = (double *)malloc(sizeof(double) * (size_t)(nred + 1));
  double *diag_pt = (double *)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
  double *Hpp_inv = (double *)malloc(sizeof(double) * 9 * (size_t)(NP + 1));
  double *d_red = (double *)malloc(sizeof(double) * (size_t)(nred + 1));
  double *d_pt = (double *)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
  double *rhs = (double *)malloc(sizeof(double) * (size_t)(nred + 1));

  double sumsq = 0;
  double cost = eval_cost(&C, cams, poses, pts, 1, &sumsq);
  Rp->initial_cost = cost;
  Rp->rmse_normalized_initial = sqrt(sumsq / (double)(M > 0 ? M : 1));
  Rp->cost_history[0] = cost;
  double radius = O->initial_radius > 0 ? O->initial_radius : 1e4;
  double decrease_factor = 2.0;
  int have_scale = 0;
  int iter = 0;
  Rp->termination = 0;
  const int max_it = O->max_iterations;
  const double eps = 2.220446049250313e-16;

  double *Dred = (double *)malloc(sizeof(double) * (size_t)(nred + 1));
  double *Dpt = (double *)malloc(sizeof(double) * 3 * (size_t)(NP + 1));
  int need_prepare = 1, n_invalid = 0;
  double gmax = 0;
  for (;;) {
    if (need_prepare) {
    /* ---- gradient + column norms of the (unscaled, corrected) Jacobian ---- */
    /* priors are rows of the Jacobian too */
    for (int i = 0; i < nred; i++) g_red[i] = 0, diag_red[i] = 0;
    for (int i = 0; i < 3 * NP; i++) g_pt[i] = 0, diag_pt[i] = 0;
    for (int64_t o = 0; o < M; o++) {
      const int s = P->obs_shot[o], p = P->obs_point[o];
      const int sv = C.shot_var[s], cv = C.cam_var[P->shot_camera[s]];
      const double *r = C.res + 2 * o, *Jp = C.Jp + 6 * o, *Jc = C.Jc + 12 * o, *Jk = C.Jk + 6 * o;
      if (!(P->point_fixed && P->point_fixed[p]))
        for (int j = 0; j < 3; j++) {
          g_pt[3 * p + j] += Jp[j] * r[0] + Jp[3 + j] * r[1];
          diag_pt[3 * p + j] += Jp[j] * Jp[j] + Jp[3 + j] * Jp[3 + j];
        }
      if (sv >= 0)
        for (int j = 0; j < 6; j++) {
          g_red[6 * sv + j] += Jc[j] * r[0] + Jc[6 + j] * r[1];
          diag_red[6 * sv + j] += Jc[j] * Jc[j] + Jc[6 + j] * Jc[6 + j];
        }
      if (cv >= 0)
        for (int j = 0; j < 3; j++) {
          g_red[cam0 + 3 * cv + j] += Jk[j] * r[0] + Jk[3 + j] * r[1];
          diag_red[cam0 + 3 * cv + j] += Jk[j] * Jk[j] + Jk[3 + j] * Jk[3 + j];
        }
    }
    for (int c = 0; c < NC; c++) {
      const int cv = C.cam_var[c];
      if (cv < 0) continue;
      const double *v = cams + 3 * c, *pr = P->cam_prior + 3 * c, *sg = P->cam_sigma + 3 * c;
      const double w0 = 1.0 / fmax(sg[0], eps), w1 = 1.0 / fmax(sg[1], eps), w2 = 1.0 / fmax(sg[2], eps);
      const double e[3] = {(v[0] - pr[0]) * w0, (v[1] - pr[1]) * w1, log(v[2] / pr[2]) * w2};
      const double j[3] = {w0, w1, w2 / v[2]};
      for (int k = 0; k < 3; k++) {
        g_red[cam0 + 3 * cv + k] += j[k] * e[k];
        diag_red[cam0 + 3 * cv + k] += j[k] * j[k];
      }
    }
    if (P->shot_gps && P->shot_gps_sigma)
      for (int s = 0; s < S; s++) {
        const int sv = C.shot_var[s];
        if (sv < 0 || !(P->shot_gps_sigma[s] > 0)) continue;
        const double w = 1.0 / P->shot_gps_sigma[s];
        for (int k = 0; k < 3; k++) {
          g_red[6 * sv + 3 + k] += w * w * (poses[6 * s + 3 + k] - P->shot_gps[3 * s + k]);
          diag_red[6 * sv + 3 + k] += w * w;
        }
      }
    if (P->shot_up && P->shot_up_sigma)
      for (int s = 0; s < S; s++) {
        const int sv = C.shot_var[s];
        if (sv < 0 || !(P->shot_up_sigma[s] > 0)) continue;
        const double *r = C.up_r + 3 * s, *J = C.up_J + 9 * s;
        for (int k = 0; k < 3; k++)
          for (int i = 0; i < 3; i++) {
            g_red[6 * sv + k] += J[3 * i + k] * r[i];
            diag_red[6 * sv + k] += J[3 * i + k] * J[3 * i + k];
          }
      }
    if (!have_scale) { /* jacobi scaling, once (ceres trust_region_minimizer: jacobian_scaling_) */
      for (int i = 0; i < nred; i++) sc_red[i] = 1.0 / (1.0 + sqrt(diag_red[i]));
      for (int i = 0; i < 3 * NP; i++) sc_pt[i] = 1.0 / (1.0 + sqrt(diag_pt[i]));
      have_scale = 1;
    }
      gmax = 0;
      for (int i = 0; i < nred; i++) gmax = fmax(gmax, fabs(g_red[i]));
      for (int i = 0; i < 3 * NP; i++) gmax = fmax(gmax, fabs(g_pt[i]));
      /* diagonal of the SCALED J^T J, clamped (levenberg_marquardt_strategy.cc) */
      for (int i = 0; i < nred; i++) Dred[i] = fmin(fmax(diag_red[i] * sc_red[i] * sc_red[i], 1e-6), 1e32);
      for (int i = 0; i < 3 * NP; i++) Dpt[i] = fmin(fmax(diag_pt[i] * sc_pt[i] * sc_pt[i], 1e-6), 1e32);
      need_prepare = 0;
    }
    /* FinalizeIterationAndCheckIfMinimizerCanContinue (trust_region_minimizer.cc) */
    if (iter >= max_it) { Rp->termination = 0; break; }
    if (gmax <= O->gradient_tolerance) { Rp->termination = 2; break; }
    if (radius < 1e-32) { Rp->termination = 4; break; }
    iter++;
    if (iter < 256) Rp->cost_history[iter] = cost; /* defined on every exit of this iteration (tolerances, invalid step) */
    {
      const double t_lin = now_s();
      /* ---- Schur complement in the scaled space ---- */
      /* All cores ("OpenMP Jacobian + block Schur + Cholesky"): every entry of the reduced system has ONE owner thread
       * -- the thread that owns the shot of its row (shot rows) or of its column (camera rows) -- which adds the contributions in the
       * order of the serial loops below, so those entries come out bit-identical to the serial elimination for any number of threads.
       * Only the small camera x camera block and the camera rows of the right-hand side, which every observation touches, are summed
       * from partial sums over a FIXED number of chunks (in chunk order: independent of the thread count as well). */
      const int ncc = 3 * C.ncv;
      if (g_ba_parallel && ncc <= 24 && C.nsv > 0) {
        enum { KCH = 256 };
        const int ncc2 = ncc * ncc;
        double *part1 = (double *)calloc((size_t)KCH * (size_t)(ncc2 + 1), sizeof(double));            /* camera-side blocks */
        double *part2 = (double *)calloc((size_t)KCH * (size_t)(ncc2 + ncc + 1), sizeof(double));      /* elimination: block + rhs */
        for (int64_t i = 0; i < A.off[nred]; i++) A.v[i] = 0;
        for (int i = 0; i < nred; i++) rhs[i] = -g_red[i] * sc_red[i];
#pragma omp parallel for schedule(static)
        for (int p = 0; p < NP; p++) {
          double *Hi = Hpp_inv + 9 * (size_t)p;
          if (P->point_fixed && P->point_fixed[p]) {
            for (int i = 0; i < 9; i++) Hi[i] = 0;
            continue;
          }
          double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int64_t q = C.pt_off[p]; q < C.pt_off[p + 1]; q++) {
            const double *Jp = C.Jp + 6 * C.pt_obs[q];
            for (int i = 0; i < 3; i++)
              for (int j = 0; j < 3; j++) H[3 * i + j] += (Jp[i] * Jp[j] + Jp[3 + i] * Jp[3 + j]) * sc_pt[3 * p + i] * sc_pt[3 * p + j];
          }
          for (int i = 0; i < 3; i++) H[4 * i] += Dpt[3 * p + i] / radius;
          const double c00 = H[4] * H[8] - H[5] * H[7], c01 = H[5] * H[6] - H[3] * H[8], c02 = H[3] * H[7] - H[4] * H[6];
          const double det = H[0] * c00 + H[1] * c01 + H[2] * c02, id = 1.0 / det;
          Hi[0] = c00 * id; Hi[1] = c01 * id; Hi[2] = c02 * id;
          Hi[3] = Hi[1]; Hi[4] = (H[0] * H[8] - H[2] * H[6]) * id; Hi[5] = (H[2] * H[3] - H[0] * H[5]) * id;
          Hi[6] = Hi[2]; Hi[7] = Hi[5]; Hi[8] = (H[0] * H[4] - H[1] * H[3]) * id;
        }
#pragma omp parallel
        {
          const int nt = omp_get_num_threads(), tn = omp_get_thread_num();
          const int lo = (int)((int64_t)C.nsv * tn / nt), hi = (int)((int64_t)C.nsv * (tn + 1) / nt); /* shot variables this thread owns */
          /* camera-side blocks from observations: the shot block and the (camera, shot) block belong to the shot's owner */
          for (int64_t o = 0; o < M; o++) {
            const int s = P->obs_shot[o];
            const int sv = C.shot_var[s], cv = C.cam_var[P->shot_camera[s]];
            if (sv < lo || sv >= hi) continue;
            const double *Jc = C.Jc + 12 * o, *Jk = C.Jk + 6 * o;
            for (int i = 0; i < 6; i++)
              for (int j = 0; j <= i; j++)
                SKY(6 * sv + i, 6 * sv + j) += (Jc[i] * Jc[j] + Jc[6 + i] * Jc[6 + j]) * sc_red[6 * sv + i] * sc_red[6 * sv + j];
            if (cv >= 0) {
              const int b = cam0 + 3 * cv;
              for (int i = 0; i < 3; i++)
                for (int j = 0; j < 6; j++)
                  SKY(b + i, 6 * sv + j) += (Jk[i] * Jc[j] + Jk[3 + i] * Jc[6 + j]) * sc_red[b + i] * sc_red[6 * sv + j];
            }
          }
          for (int s = 0; s < S; s++) {
            const int sv = C.shot_var[s];
            if (sv < lo || sv >= hi) continue;
            if (P->shot_gps && P->shot_gps_sigma && P->shot_gps_sigma[s] > 0) {
              const double w = 1.0 / P->shot_gps_sigma[s];
              for (int k = 0; k < 3; k++) SKY(6 * sv + 3 + k, 6 * sv + 3 + k) += w * w * sc_red[6 * sv + 3 + k] * sc_red[6 * sv + 3 + k];
            }
          }
          for (int s = 0; s < S; s++) {
            const int sv = C.shot_var[s];
            if (sv < lo || sv >= hi) continue;
            if (P->shot_up && P->shot_up_sigma && P->shot_up_sigma[s] > 0) {
              const double *J = C.up_J + 9 * s;
              for (int i = 0; i < 3; i++)
                for (int j = 0; j <= i; j++)
                  SKY(6 * sv + i, 6 * sv + j) += (J[i] * J[j] + J[3 + i] * J[3 + j] + J[6 + i] * J[6 + j]) * sc_red[6 * sv + i] * sc_red[6 * sv + j];
            }
          }
          for (int i = 6 * lo; i < 6 * hi; i++) SKY(i, i) += Dred[i] / radius;
          /* eliminate points: entries with a shot row belong to the owner of that shot, (camera, shot) entries to the owner of the
           * shot of the COLUMN; (p, qa, qb) are walked in the serial order */
          for (int p = 0; p < NP; p++) {
            if (P->point_fixed && P->point_fixed[p]) continue;
            const int64_t q0 = C.pt_off[p], q1 = C.pt_off[p + 1];
            int mine = 0;
            for (int64_t q = q0; q < q1 && !mine; q++) {
              const int sv = C.shot_var[P->obs_shot[C.pt_obs[q]]];
              mine = sv >= lo && sv < hi;
            }
            if (!mine) continue;
            const double *Hi = Hpp_inv + 9 * (size_t)p;
            double gp[3], Hg[3];
            for (int i = 0; i < 3; i++) gp[i] = -g_pt[3 * p + i] * sc_pt[3 * p + i];
            for (int i = 0; i < 3; i++) Hg[i] = Hi[3 * i] * gp[0] + Hi[3 * i + 1] * gp[1] + Hi[3 * i + 2] * gp[2];
            for (int64_t qa = q0; qa < q1; qa++) {
              const int64_t oa = C.pt_obs[qa];
              const int sa = P->obs_shot[oa];
              const int sva = C.shot_var[sa], cva = C.cam_var[P->shot_camera[sa]];
              const int own_a = sva >= lo && sva < hi;
              const double *Jpa = C.Jp + 6 * oa, *Jca = C.Jc + 12 * oa, *Jka = C.Jk + 6 * oa;
              double Wa[9][3], WH[9][3];
              int ia[9], na = 0;
              if (sva >= 0)
                for (int i = 0; i < 6; i++) {
                  for (int j = 0; j < 3; j++) Wa[na][j] = (Jca[i] * Jpa[j] + Jca[6 + i] * Jpa[3 + j]) * sc_red[6 * sva + i] * sc_pt[3 * p + j];
                  ia[na++] = 6 * sva + i;
                }
              if (cva >= 0)
                for (int i = 0; i < 3; i++) {
                  for (int j = 0; j < 3; j++) Wa[na][j] = (Jka[i] * Jpa[j] + Jka[3 + i] * Jpa[3 + j]) * sc_red[cam0 + 3 * cva + i] * sc_pt[3 * p + j];
                  ia[na++] = cam0 + 3 * cva + i;
                }
              for (int i = 0; i < na; i++) {
                for (int j = 0; j < 3; j++) WH[i][j] = Wa[i][0] * Hi[j] + Wa[i][1] * Hi[3 + j] + Wa[i][2] * Hi[6 + j];
                if (ia[i] < cam0 && own_a) rhs[ia[i]] -= Wa[i][0] * Hg[0] + Wa[i][1] * Hg[1] + Wa[i][2] * Hg[2];
              }
              for (int64_t qb = q0; qb < q1; qb++) {
                const int64_t ob = C.pt_obs[qb];
                const int sb = P->obs_shot[ob];
                const int svb = C.shot_var[sb];
                if (svb < 0) continue; /* only shot columns here: camera columns are the camera x camera block below */
                const int own_b = svb >= lo && svb < hi;
                if (!own_a && !own_b) continue;
                const double *Jpb = C.Jp + 6 * ob, *Jcb = C.Jc + 12 * ob;
                double Wb[6][3];
                for (int i = 0; i < 6; i++)
                  for (int j = 0; j < 3; j++) Wb[i][j] = (Jcb[i] * Jpb[j] + Jcb[6 + i] * Jpb[3 + j]) * sc_red[6 * svb + i] * sc_pt[3 * p + j];
                for (int i = 0; i < na; i++) {
                  const int cam_row = ia[i] >= cam0;
                  if (cam_row ? !own_b : !own_a) continue;
                  for (int j = 0; j < 6; j++)
                    if (6 * svb + j <= ia[i]) SKY(ia[i], 6 * svb + j) -= WH[i][0] * Wb[j][0] + WH[i][1] * Wb[j][1] + WH[i][2] * Wb[j][2];
                }
              }
            }
          }
        }
        if (ncc > 0) {
          /* camera x camera block and the camera rows of the right-hand side: fixed chunks, summed in chunk order */
#pragma omp parallel for schedule(dynamic, 1)
          for (int ch = 0; ch < KCH; ch++) {
            double *a1 = part1 + (size_t)ch * (size_t)(ncc2 + 1);
            for (int64_t o = M * ch / KCH; o < M * (ch + 1) / KCH; o++) {
              const int cv = C.cam_var[P->shot_camera[P->obs_shot[o]]];
              if (cv < 0) continue;
              const double *Jk = C.Jk + 6 * o;
              const int b = cam0 + 3 * cv;
              for (int i = 0; i < 3; i++)
                for (int j = 0; j <= i; j++)
                  a1[(3 * cv + i) * ncc + 3 * cv + j] += (Jk[i] * Jk[j] + Jk[3 + i] * Jk[3 + j]) * sc_red[b + i] * sc_red[b + j];
            }
            double *a2 = part2 + (size_t)ch * (size_t)(ncc2 + ncc + 1);
            for (int p = (int)((int64_t)NP * ch / KCH); p < (int)((int64_t)NP * (ch + 1) / KCH); p++) {
              if (P->point_fixed && P->point_fixed[p]) continue;
              const double *Hi = Hpp_inv + 9 * (size_t)p;
              double gp[3], Hg[3];
              for (int i = 0; i < 3; i++) gp[i] = -g_pt[3 * p + i] * sc_pt[3 * p + i];
              for (int i = 0; i < 3; i++) Hg[i] = Hi[3 * i] * gp[0] + Hi[3 * i + 1] * gp[1] + Hi[3 * i + 2] * gp[2];
              const int64_t q0 = C.pt_off[p], q1 = C.pt_off[p + 1];
              for (int64_t qa = q0; qa < q1; qa++) {
                const int64_t oa = C.pt_obs[qa];
                const int cva = C.cam_var[P->shot_camera[P->obs_shot[oa]]];
                if (cva < 0) continue;
                const double *Jpa = C.Jp + 6 * oa, *Jka = C.Jk + 6 * oa;
                double Wa[3][3], WH[3][3];
                for (int i = 0; i < 3; i++) {
                  for (int j = 0; j < 3; j++) Wa[i][j] = (Jka[i] * Jpa[j] + Jka[3 + i] * Jpa[3 + j]) * sc_red[cam0 + 3 * cva + i] * sc_pt[3 * p + j];
                  for (int j = 0; j < 3; j++) WH[i][j] = Wa[i][0] * Hi[j] + Wa[i][1] * Hi[3 + j] + Wa[i][2] * Hi[6 + j];
                  a2[ncc2 + 3 * cva + i] += Wa[i][0] * Hg[0] + Wa[i][1] * Hg[1] + Wa[i][2] * Hg[2];
                }
                for (int64_t qb = q0; qb < q1; qb++) {
                  const int64_t ob = C.pt_obs[qb];
                  const int cvb = C.cam_var[P->shot_camera[P->obs_shot[ob]]];
                  if (cvb < 0 || cvb > cva) continue;
                  const double *Jpb = C.Jp + 6 * ob, *Jkb = C.Jk + 6 * ob;
                  for (int j = 0; j < 3; j++) {
                    double wb[3];
                    for (int c = 0; c < 3; c++) wb[c] = (Jkb[j] * Jpb[c] + Jkb[3 + j] * Jpb[3 + c]) * sc_red[cam0 + 3 * cvb + j] * sc_pt[3 * p + c];
                    for (int i = 0; i < 3; i++)
                      if (3 * cvb + j <= 3 * cva + i) a2[(3 * cva + i) * ncc + 3 * cvb + j] += WH[i][0] * wb[0] + WH[i][1] * wb[1] + WH[i][2] * wb[2];
                  }
                }
              }
            }
          }
          for (int ch = 0; ch < KCH; ch++)
            for (int i = 0; i < ncc; i++)
              for (int j = 0; j <= i; j++) SKY(cam0 + i, cam0 + j) += part1[(size_t)ch * (size_t)(ncc2 + 1) + (size_t)i * ncc + j];
          for (int c = 0; c < NC; c++) {
            const int cv = C.cam_var[c];
            if (cv < 0) continue;
            const double *v = cams + 3 * c, *sg = P->cam_sigma + 3 * c;
            const double j[3] = {1.0 / fmax(sg[0], eps), 1.0 / fmax(sg[1], eps), 1.0 / fmax(sg[2], eps) / v[2]};
            for (int k = 0; k < 3; k++) SKY(cam0 + 3 * cv + k, cam0 + 3 * cv + k) += j[k] * j[k] * sc_red[cam0 + 3 * cv + k] * sc_red[cam0 + 3 * cv + k];
          }
          for (int i = cam0; i < nred; i++) SKY(i, i) += Dred[i] / radius;
          for (int ch = 0; ch < KCH; ch++) {
            const double *a2 = part2 + (size_t)ch * (size_t)(ncc2 + ncc + 1);
            for (int i = 0; i < ncc; i++) {
              for (int j = 0; j <= i; j++) SKY(cam0 + i, cam0 + j) -= a2[(size_t)i * ncc + j];
              rhs[cam0 + i] -= a2[ncc2 + i];
            }
          }
        }
        free(part1);
        free(part2);
      } else {
      for (int64_t i = 0; i < A.off[nred]; i++) A.v[i] = 0;
      for (int i = 0; i < nred; i++) rhs[i] = -g_red[i] * sc_red[i];
      /* camera-side blocks from observations */
      for (int64_t o = 0; o < M; o++) {
        const int s = P->obs_shot[o];
        const int sv = C.shot_var[s], cv = C.cam_var[P->shot_camera[s]];
        const double *Jc = C.Jc + 12 * o, *Jk = C.Jk + 6 * o;
        if (sv >= 0)
          for (int i = 0; i < 6; i++)
            for (int j = 0; j <= i; j++)
              SKY(6 * sv + i, 6 * sv + j) += (Jc[i] * Jc[j] + Jc[6 + i] * Jc[6 + j]) * sc_red[6 * sv + i] * sc_red[6 * sv + j];
        if (cv >= 0) {
          const int b = cam0 + 3 * cv;
          for (int i = 0; i < 3; i++)
            for (int j = 0; j <= i; j++) SKY(b + i, b + j) += (Jk[i] * Jk[j] + Jk[3 + i] * Jk[3 + j]) * sc_red[b + i] * sc_red[b + j];
          if (sv >= 0)
            for (int i = 0; i < 3; i++)
              for (int j = 0; j < 6; j++)
                SKY(b + i, 6 * sv + j) += (Jk[i] * Jc[j] + Jk[3 + i] * Jc[6 + j]) * sc_red[b + i] * sc_red[6 * sv + j];
        }
      }
      for (int c = 0; c < NC; c++) {
        const int cv = C.cam_var[c];
        if (cv < 0) continue;
        const double *v = cams + 3 * c, *sg = P->cam_sigma + 3 * c;
        const double j[3] = {1.0 / fmax(sg[0], eps), 1.0 / fmax(sg[1], eps), 1.0 / fmax(sg[2], eps) / v[2]};
        for (int k = 0; k < 3; k++) SKY(cam0 + 3 * cv + k, cam0 + 3 * cv + k) += j[k] * j[k] * sc_red[cam0 + 3 * cv + k] * sc_red[cam0 + 3 * cv + k];
      }
      if (P->shot_gps && P->shot_gps_sigma)
        for (int s = 0; s < S; s++) {
          const int sv = C.shot_var[s];
          if (sv < 0 || !(P->shot_gps_sigma[s] > 0)) continue;
          const double w = 1.0 / P->shot_gps_sigma[s];
          for (int k = 0; k < 3; k++) SKY(6 * sv + 3 + k, 6 * sv + 3 + k) += w * w * sc_red[6 * sv + 3 + k] * sc_red[6 * sv + 3 + k];
        }
      if (P->shot_up && P->shot_up_sigma)
        for (int s = 0; s < S; s++) {
          const int sv = C.shot_var[s];
          if (sv < 0 || !(P->shot_up_sigma[s] > 0)) continue;
          const double *J = C.up_J + 9 * s;
          for (int i = 0; i < 3; i++)
            for (int j = 0; j <= i; j++)
              SKY(6 * sv + i, 6 * sv + j) += (J[i] * J[j] + J[3 + i] * J[3 + j] + J[6 + i] * J[6 + j]) * sc_red[6 * sv + i] * sc_red[6 * sv + j];
        }
      for (int i = 0; i < nred; i++) SKY(i, i) += Dred[i] / radius;
      /* eliminate points */
      for (int p = 0; p < NP; p++) {
        double *Hi = Hpp_inv + 9 * (size_t)p;
        if (P->point_fixed && P->point_fixed[p]) {
          for (int i = 0; i < 9; i++) Hi[i] = 0;
          continue;
        }
        double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int64_t q = C.pt_off[p]; q < C.pt_off[p + 1]; q++) {
          const double *Jp = C.Jp + 6 * C.pt_obs[q];
          for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) H[3 * i + j] += (Jp[i] * Jp[j] + Jp[3 + i] * Jp[3 + j]) * sc_pt[3 * p + i] * sc_pt[3 * p + j];
        }
        for (int i = 0; i < 3; i++) H[4 * i] += Dpt[3 * p + i] / radius;
        /* 3x3 symmetric inverse */
        const double c00 = H[4] * H[8] - H[5] * H[7], c01 = H[5] * H[6] - H[3] * H[8], c02 = H[3] * H[7] - H[4] * H[6];
        const double det = H[0] * c00 + H[1] * c01 + H[2] * c02, id = 1.0 / det;
        Hi[0] = c00 * id; Hi[1] = c01 * id; Hi[2] = c02 * id;
        Hi[3] = Hi[1]; Hi[4] = (H[0] * H[8] - H[2] * H[6]) * id; Hi[5] = (H[2] * H[3] - H[0] * H[5]) * id;
        Hi[6] = Hi[2]; Hi[7] = Hi[5]; Hi[8] = (H[0] * H[4] - H[1] * H[3]) * id;
        /* W_o = Jred_o^T Jp_o (scaled), for all observations of p ; S -= W_a Hi W_b^T ; rhs -= W_a Hi (-g_p) */
        const int64_t q0 = C.pt_off[p], q1 = C.pt_off[p + 1];
        double gp[3];
        for (int i = 0; i < 3; i++) gp[i] = -g_pt[3 * p + i] * sc_pt[3 * p + i];
        double Hg[3];
        for (int i = 0; i < 3; i++) Hg[i] = Hi[3 * i] * gp[0] + Hi[3 * i + 1] * gp[1] + Hi[3 * i + 2] * gp[2];
        for (int64_t qa = q0; qa < q1; qa++) {
          const int64_t oa = C.pt_obs[qa];
          const int sa = P->obs_shot[oa];
          const int sva = C.shot_var[sa], cva = C.cam_var[P->shot_camera[sa]];
          /* rows of W_a: 6 (shot) + 3 (camera), each 3 wide */
          double Wa[9][3];
          int ia[9], na = 0;
          const double *Jpa = C.Jp + 6 * oa, *Jca = C.Jc + 12 * oa, *Jka = C.Jk + 6 * oa;
          if (sva >= 0)
            for (int i = 0; i < 6; i++) {
              for (int j = 0; j < 3; j++) Wa[na][j] = (Jca[i] * Jpa[j] + Jca[6 + i] * Jpa[3 + j]) * sc_red[6 * sva + i] * sc_pt[3 * p + j];
              ia[na++] = 6 * sva + i;
            }
          if (cva >= 0)
            for (int i = 0; i < 3; i++) {
              for (int j = 0; j < 3; j++) Wa[na][j] = (Jka[i] * Jpa[j] + Jka[3 + i] * Jpa[3 + j]) * sc_red[cam0 + 3 * cva + i] * sc_pt[3 * p + j];
              ia[na++] = cam0 + 3 * cva + i;
            }
          double WH[9][3];
          for (int i = 0; i < na; i++) {
            for (int j = 0; j < 3; j++) WH[i][j] = Wa[i][0] * Hi[j] + Wa[i][1] * Hi[3 + j] + Wa[i][2] * Hi[6 + j];
            rhs[ia[i]] -= Wa[i][0] * Hg[0] + Wa[i][1] * Hg[1] + Wa[i][2] * Hg[2];
          }
          for (int64_t qb = q0; qb < q1; qb++) {
            const int64_t ob = C.pt_obs[qb];
            const int sb = P->obs_shot[ob];
            const int svb = C.shot_var[sb], cvb = C.cam_var[P->shot_camera[sb]];
            const double *Jpb = C.Jp + 6 * ob, *Jcb = C.Jc + 12 * ob, *Jkb = C.Jk + 6 * ob;
            double Wb[9][3];
            int ib[9], nb = 0;
            if (svb >= 0)
              for (int i = 0; i < 6; i++) {
                for (int j = 0; j < 3; j++) Wb[nb][j] = (Jcb[i] * Jpb[j] + Jcb[6 + i] * Jpb[3 + j]) * sc_red[6 * svb + i] * sc_pt[3 * p + j];
                ib[nb++] = 6 * svb + i;
              }
            if (cvb >= 0)
              for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) Wb[nb][j] = (Jkb[i] * Jpb[j] + Jkb[3 + i] * Jpb[3 + j]) * sc_red[cam0 + 3 * cvb + i] * sc_pt[3 * p + j];
                ib[nb++] = cam0 + 3 * cvb + i;
              }
            for (int i = 0; i < na; i++)
              for (int j = 0; j < nb; j++)
                if (ib[j] <= ia[i]) {
                  /* camera rows appear once per observation: the (cam,cam) and (cam,shot) sums over
                   * (a,b) pairs are exactly the Schur products */
                  SKY(ia[i], ib[j]) -= WH[i][0] * Wb[j][0] + WH[i][1] * Wb[j][1] + WH[i][2] * Wb[j][2];
                }
          }
        }
      }
      }
      /* factor + solve */
      double *dx = d_red;
      for (int i = 0; i < nred; i++) dx[i] = rhs[i];
      int bad = 0;
      if (nred > 0) {
        /* g_ba_parallel == 2 (opt-in): the right-looking factor on up to 16 cores.  It is NOT the default: with hundreds of OpenMP threads --
         * the GPU box's host, or several test workers at once -- its two barriers per column cost more than the factor itself (round 4: a run
         * of the GPU suite with it on by default did not finish), and nothing on a GPU box may depend on an unmeasured path. */
        bad = g_ba_parallel == 2 ? skyline_cholesky_parallel(&A) : skyline_cholesky(&A);
      }
      if (!bad && nred > 0) skyline_solve(&A, dx);
      Rp->seconds_linear_solver += now_s() - t_lin;
      if (bad) { /* factorisation failed: invalid step */
        radius *= 0.5;
        if (++n_invalid >= 5) { Rp->termination = -1; break; }
        continue;
      }
      /* back-substitute points: dp = Hi ( -g_p - sum_o W_o^T dx_o ) */
      for (int p = 0; p < NP; p++) {
        const double *Hi = Hpp_inv + 9 * (size_t)p;
        double b[3];
        for (int i = 0; i < 3; i++) b[i] = -g_pt[3 * p + i] * sc_pt[3 * p + i];
        for (int64_t q = C.pt_off[p]; q < C.pt_off[p + 1]; q++) {
          const int64_t o = C.pt_obs[q];
          const int s = P->obs_shot[o];
          const int sv = C.shot_var[s], cv = C.cam_var[P->shot_camera[s]];
          const double *Jp = C.Jp + 6 * o, *Jc = C.Jc + 12 * o, *Jk = C.Jk + 6 * o;
          double t0 = 0, t1 = 0; /* J_red dx (2-vector) */
          if (sv >= 0)
            for (int i = 0; i < 6; i++) {
              t0 += Jc[i] * sc_red[6 * sv + i] * dx[6 * sv + i];
              t1 += Jc[6 + i] * sc_red[6 * sv + i] * dx[6 * sv + i];
            }
          if (cv >= 0)
            for (int i = 0; i < 3; i++) {
              t0 += Jk[i] * sc_red[cam0 + 3 * cv + i] * dx[cam0 + 3 * cv + i];
              t1 += Jk[3 + i] * sc_red[cam0 + 3 * cv + i] * dx[cam0 + 3 * cv + i];
            }
          for (int j = 0; j < 3; j++) b[j] -= (Jp[j] * t0 + Jp[3 + j] * t1) * sc_pt[3 * p + j];
        }
        for (int i = 0; i < 3; i++) d_pt[3 * p + i] = Hi[3 * i] * b[0] + Hi[3 * i + 1] * b[1] + Hi[3 * i + 2] * b[2];
      }
      /* model cost change = -m^T (r + m/2), m = J delta (unscaled J, unscaled delta) */
      double model_change = 0;
      for (int64_t o = 0; o < M; o++) {
        const int s = P->obs_shot[o], p = P->obs_point[o];
        const int sv = C.shot_var[s], cv = C.cam_var[P->shot_camera[s]];
        const double *Jp = C.Jp + 6 * o, *Jc = C.Jc + 12 * o, *Jk = C.Jk + 6 * o, *r = C.res + 2 * o;
        double m0 = 0, m1 = 0;
        for (int j = 0; j < 3; j++) {
          const double d = d_pt[3 * p + j] * sc_pt[3 * p + j];
          m0 += Jp[j] * d;
          m1 += Jp[3 + j] * d;
        }
        if (sv >= 0)
          for (int j = 0; j < 6; j++) {
            const double d = dx[6 * sv + j] * sc_red[6 * sv + j];
            m0 += Jc[j] * d;
            m1 += Jc[6 + j] * d;
          }
        if (cv >= 0)
          for (int j = 0; j < 3; j++) {
            const double d = dx[cam0 + 3 * cv + j] * sc_red[cam0 + 3 * cv + j];
            m0 += Jk[j] * d;
            m1 += Jk[3 + j] * d;
          }
        model_change -= m0 * (r[0] + 0.5 * m0) + m1 * (r[1] + 0.5 * m1);
      }
      for (int c = 0; c < NC; c++) {
        const int cv = C.cam_var[c];
        if (cv < 0) continue;
        const double *v = cams + 3 * c, *pr = P->cam_prior + 3 * c, *sg = P->cam_sigma + 3 * c;
        const double w[3] = {1.0 / fmax(sg[0], eps), 1.0 / fmax(sg[1], eps), 1.0 / fmax(sg[2], eps)};
        const double e[3] = {(v[0] - pr[0]) * w[0], (v[1] - pr[1]) * w[1], log(v[2] / pr[2]) * w[2]};
        const double j[3] = {w[0], w[1], w[2] / v[2]};
        for (int k = 0; k < 3; k++) {
          const double m = j[k] * dx[cam0 + 3 * cv + k] * sc_red[cam0 + 3 * cv + k];
          model_change -= m * (e[k] + 0.5 * m);
        }
      }
      if (P->shot_gps && P->shot_gps_sigma)
        for (int s = 0; s < S; s++) {
          const int sv = C.shot_var[s];
          if (sv < 0 || !(P->shot_gps_sigma[s] > 0)) continue;
          const double w = 1.0 / P->shot_gps_sigma[s];
          for (int k = 0; k < 3; k++) {
            const double m = w * dx[6 * sv + 3 + k] * sc_red[6 * sv + 3 + k];
            const double e = w * (poses[6 * s + 3 + k] - P->shot_gps[3 * s + k]);
            model_change -= m * (e + 0.5 * m);
          }
        }
      if (P->shot_up && P->shot_up_sigma)
        for (int s = 0; s < S; s++) {
          const int sv = C.shot_var[s];
          if (sv < 0 || !(P->shot_up_sigma[s] > 0)) continue;
          const double *r = C.up_r + 3 * s, *J = C.up_J + 9 * s;
          for (int i = 0; i < 3; i++) {
            double m = 0;
            for (int k = 0; k < 3; k++) m += J[3 * i + k] * dx[6 * sv + k] * sc_red[6 * sv + k];
            model_change -= m * (r[i] + 0.5 * m);
          }
        }
      /* candidate */
      double step_sq = 0, x_sq = 0;
      memcpy(cams_n, cams, sizeof(double) * 3 * (size_t)NC);
      memcpy(poses_n, poses, sizeof(double) * 6 * (size_t)S);
      memcpy(pts_n, pts, sizeof(double) * 3 * (size_t)NP);
      for (int c = 0; c < NC; c++) {
        const int cv = C.cam_var[c];
        for (int k = 0; k < 3; k++) {
          if (cv >= 0) {
            const double d = dx[cam0 + 3 * cv + k] * sc_red[cam0 + 3 * cv + k];
            cams_n[3 * c + k] += d;
            step_sq += d * d;
            x_sq += cams[3 * c + k] * cams[3 * c + k];
          }
        }
      }
      for (int s = 0; s < S; s++) {
        const int sv = C.shot_var[s];
        if (sv < 0) continue;
        for (int k = 0; k < 6; k++) {
          const double d = dx[6 * sv + k] * sc_red[6 * sv + k];
          poses_n[6 * s + k] += d;
          step_sq += d * d;
          x_sq += poses[6 * s + k] * poses[6 * s + k];
        }
      }
      for (int p = 0; p < NP; p++) {
        if (P->point_fixed && P->point_fixed[p]) continue;
        for (int k = 0; k < 3; k++) {
          const double d = d_pt[3 * p + k] * sc_pt[3 * p + k];
          pts_n[3 * p + k] += d;
          step_sq += d * d;
          x_sq += pts[3 * p + k] * pts[3 * p + k];
        }
      }
      if (!(model_change > 0)) { /* HandleInvalidStep + LevenbergMarquardtStrategy::StepIsInvalid */
        radius *= 0.5;
        if (++n_invalid >= 5) { Rp->termination = -1; break; }
        continue;
      }
      n_invalid = 0;
      const double cost_n = eval_cost(&C, cams_n, poses_n, pts_n, 0, NULL);
      const double step_norm = sqrt(step_sq), x_norm = sqrt(x_sq);
      /* ParameterToleranceReached / FunctionToleranceReached: before accept/reject, candidate dropped */
      if (step_norm <= O->parameter_tolerance * (x_norm + O->parameter_tolerance)) { Rp->termination = 3; break; }
      const double cost_change = cost - cost_n;
      if (fabs(cost_change) <= O->function_tolerance * cost) { Rp->termination = 1; break; }
      const double rho = cost_change / model_change;
      if (O->verbose) fprintf(stderr, "[ba_oracle] it %d cost %.9e -> %.9e rho %.3f radius %.3e\n", iter, cost, cost_n, rho, radius);
      if (rho > 1e-3) { /* StepAccepted */
        memcpy(cams, cams_n, sizeof(double) * 3 * (size_t)NC);
        memcpy(poses, poses_n, sizeof(double) * 6 * (size_t)S);
        memcpy(pts, pts_n, sizeof(double) * 3 * (size_t)NP);
        const double t = 2.0 * rho - 1.0;
        radius = radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
        radius = fmin(1e16, radius);
        decrease_factor = 2.0;
        Rp->successful_steps++;
        cost = eval_cost(&C, cams, poses, pts, 1, &sumsq);
        need_prepare = 1;
      } else { /* StepRejected */
        radius = radius / decrease_factor;
        decrease_factor *= 2.0;
      }
      if (iter < 256) Rp->cost_history[iter] = cost;
    }
  }
  free(Dred);
  free(Dpt);
  Rp->iterations = iter;
  Rp->final_cost = cost;
  /* outputs */
  memcpy(P->cam_params, cams, sizeof(double) * 3 * (size_t)NC);
  memcpy(P->shot_pose, poses, sizeof(double) * 6 * (size_t)S);
  memcpy(P->points, pts, sizeof(double) * 3 * (size_t)NP);
  /* ComputeReprojectionErrors (bundle_adjuster.cc:1196-1208,531-566): residual with sigma = 1 */
  {
    double ss = 0;
    for (int64_t o = 0; o < M; o++) {
      const int s = P->obs_shot[o], p = P->obs_point[o];
      double R[9], dR[3][9], r[2];
      rot_and_derivs(poses + 6 * s, R, dR);
      const int cmodel = P->cam_model ? P->cam_model[P->shot_camera[s]] : 0;
      project_obs(cmodel, pts + 3 * (size_t)p, poses + 6 * (size_t)s, R, (const double (*)[9])dR,
                  cmodel >= 2 ? P->cam_ext + 16 * (size_t)P->shot_camera[s] : cams + 3 * (size_t)P->shot_camera[s], P->obs_xy + 2 * o, 1.0, r,
                  NULL, NULL, NULL);
      if (P->reproj_err) {
        P->reproj_err[2 * o] = r[0];
        P->reproj_err[2 * o + 1] = r[1];
      }
      ss += r[0] * r[0] + r[1] * r[1];
    }
    Rp->rmse_normalized_final = sqrt(ss / (double)(M > 0 ? M : 1));
  }
  Rp->seconds_total = now_s() - t_start;
  free(C.cam_var); free(C.shot_var); free(C.res); free(C.Jp); free(C.Jc); free(C.Jk); free(C.up_r); free(C.up_J); free(C.pt_off); free(C.pt_obs);
  free(A.first); free(A.off); free(A.v);
  free(cams); free(poses); free(pts); free(cams_n); free(poses_n); free(pts_n);
  free(sc_red); free(sc_pt); free(g_red); free(g_pt); free(diag_red); free(diag_pt); free(Hpp_inv); free(d_red); free(d_pt); free(rhs);
  return 0;
}
