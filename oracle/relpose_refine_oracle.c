/* relpose_refine_oracle.c -- CPU restatement of the relative-pose refinement of robust_match_calibrated.
 *
 * TEST INFRASTRUCTURE ONLY, GROUNDWORK (no product counterpart yet; see relpose_oracle.c).
 * reference: RelativePoseRefinement + RelativePoseCost (opensfm/src/geometry/relative_pose.h:86-183):
 *   - 100 correspondences picked with std::srand(42); index = float(std::rand()) / RAND_MAX * count
 *     (glibc's TYPE_3 additive-feedback generator, restated below; known answers from this box's libc in the tests);
 *   - residual_i = 1 - (px.x + py.y)/2 with the midpoint triangulation of the two rays, residual_100 = 1 - |c|;
 *   - parameters: angle-axis of R and the centre c = -R^T t of the second camera;
 *   - ceres::TinySolver (Levenberg-Marquardt on the normal equations, Jacobi scaling from the first Jacobian,
 *     LM diagonal clamped to [1e-6, 1e32], rho > 0 accepts, u *= max(1/3, 1 - (2 rho - 1)^3), else u *= v, v *= 2),
 *     restated from the published header (ceres/tiny_solver.h, Ceres 2.1: third-party, not in /root/reference);
 *   - derivatives: forward-mode duals over the 6 parameters (what TinySolverAutoDiffFunction does with Jets).
 * Parity: unpinned vs the reference binary; pinned by known answers of rand(), by finite differences of the
 * Jacobian and by the refinement lowering its own cost / improving a perturbed pose (tests/test_oracle_relpose.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- glibc rand() (random_r.c, TYPE_3: x^31 + x^3 + 1) ---- */
typedef struct { int32_t r[344 + 1024]; int n; } glibc_rand_t;
static void glibc_srand(glibc_rand_t *g, uint32_t seed) {
  int32_t *r = g->r;
  if (seed == 0) seed = 1;
  r[0] = (int32_t)seed;
  for (int i = 1; i < 31; i++) {
    const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
    int64_t word = 16807 * lo - 2836 * hi;
    if (word < 0) word += 2147483647;
    r[i] = (int32_t)word;
  }
  for (int i = 31; i < 34; i++) r[i] = r[i - 31];
  for (int i = 34; i < 344; i++) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
  g->n = 344;
}
static int glibc_rand(glibc_rand_t *g) {
  int32_t *r = g->r;
  if (g->n >= 344 + 1024) { /* keep the last 34 values and continue */
    memmove(r, r + g->n - 34, 34 * sizeof(int32_t));
    g->n = 34;
  }
  r[g->n] = (int32_t)((uint32_t)r[g->n - 31] + (uint32_t)r[g->n - 3]);
  const int out = (int)(((uint32_t)r[g->n]) >> 1);
  g->n++;
  return out;
}
void oracle_glibc_rand(uint32_t seed, int count, int *out) {
  glibc_rand_t g;
  glibc_srand(&g, seed);
  for (int i = 0; i < count; i++) out[i] = glibc_rand(&g);
}

/* ---- forward-mode duals over 6 parameters ---- */
#define NPAR 6
typedef struct { double v, d[NPAR]; } d6;
static d6 dc(double c) { d6 r; r.v = c; for (int i = 0; i < NPAR; i++) r.d[i] = 0; return r; }
static d6 dvar(double c, int k) { d6 r = dc(c); r.d[k] = 1.0; return r; }
static d6 dadd(d6 a, d6 b) { for (int i = 0; i < NPAR; i++) a.d[i] += b.d[i]; a.v += b.v; return a; }
static d6 dsub(d6 a, d6 b) { for (int i = 0; i < NPAR; i++) a.d[i] -= b.d[i]; a.v -= b.v; return a; }
static d6 dneg(d6 a) { for (int i = 0; i < NPAR; i++) a.d[i] = -a.d[i]; a.v = -a.v; return a; }
static d6 dmul(d6 a, d6 b) { d6 r; r.v = a.v * b.v; for (int i = 0; i < NPAR; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
static d6 dmulc(d6 a, double c) { for (int i = 0; i < NPAR; i++) a.d[i] *= c; a.v *= c; return a; }
static d6 ddiv(d6 a, d6 b) { d6 r; const double ib = 1.0 / b.v; r.v = a.v * ib; for (int i = 0; i < NPAR; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
static d6 dsqrt(d6 a) { d6 r; r.v = sqrt(a.v); const double h = 0.5 / r.v; for (int i = 0; i < NPAR; i++) r.d[i] = a.d[i] * h; return r; }
static d6 dsin(d6 a) { d6 r; r.v = sin(a.v); const double c = cos(a.v); for (int i = 0; i < NPAR; i++) r.d[i] = a.d[i] * c; return r; }
static d6 dcos(d6 a) { d6 r; r.v = cos(a.v); const double s = -sin(a.v); for (int i = 0; i < NPAR; i++) r.d[i] = a.d[i] * s; return r; }
static d6 ddot3(const d6 *a, const d6 *b) { return dadd(dadd(dmul(a[0], b[0]), dmul(a[1], b[1])), dmul(a[2], b[2])); }

/* ceres::AngleAxisRotatePoint (ceres/rotation.h) */
static void aa_rotate(const d6 *aa, const d6 *pt, d6 *out) {
  const d6 theta2 = ddot3(aa, aa);
  if (theta2.v > 2.220446049250313e-16) {
    const d6 theta = dsqrt(theta2), ct = dcos(theta), st = dsin(theta), ti = ddiv(dc(1.0), theta);
    const d6 w[3] = {dmul(aa[0], ti), dmul(aa[1], ti), dmul(aa[2], ti)};
    const d6 wxp[3] = {dsub(dmul(w[1], pt[2]), dmul(w[2], pt[1])), dsub(dmul(w[2], pt[0]), dmul(w[0], pt[2])), dsub(dmul(w[0], pt[1]), dmul(w[1], pt[0]))};
    const d6 tmp = dmul(ddot3(w, pt), dsub(dc(1.0), ct));
    for (int i = 0; i < 3; i++) out[i] = dadd(dadd(dmul(pt[i], ct), dmul(wxp[i], st)), dmul(w[i], tmp));
  } else {
    const d6 wxp[3] = {dsub(dmul(aa[1], pt[2]), dmul(aa[2], pt[1])), dsub(dmul(aa[2], pt[0]), dmul(aa[0], pt[2])), dsub(dmul(aa[0], pt[1]), dmul(aa[1], pt[0]))};
    for (int i = 0; i < 3; i++) out[i] = dadd(pt[i], wxp[i]);
  }
}

#define MAX_ERRORS 100
typedef struct {
  const double *b1, *b2;
  int picked[MAX_ERRORS];
} refine_cost;

/* RelativePoseCost::operator(): residuals[101]; jac (101 x 6, row-major) when not NULL */
static void cost_eval(const refine_cost *C, const double *par, double *res, double *jac) {
  d6 rot[3], tr[3], rot_t[3];
  for (int i = 0; i < 3; i++) {
    rot[i] = dvar(par[i], i);
    tr[i] = dvar(par[3 + i], 3 + i);
    rot_t[i] = dneg(rot[i]);
  }
  for (int i = 0; i < MAX_ERRORS; i++) {
    const double *xs = C->b1 + 3 * C->picked[i], *ys = C->b2 + 3 * C->picked[i];
    const d6 x[3] = {dc(xs[0]), dc(xs[1]), dc(xs[2])}, y[3] = {dc(ys[0]), dc(ys[1]), dc(ys[2])};
    d6 ry[3];
    aa_rotate(rot_t, y, ry);
    /* TriangulateTwoBearingsMidpointSolve(centers = (0, tr), bearings = (x, ry)) */
    const d6 b0 = ddot3(tr, x), b1v = ddot3(tr, ry);
    const d6 a00 = ddot3(x, x), a10 = ddot3(x, ry), a01 = dneg(a10), a11 = dneg(ddot3(ry, ry));
    const d6 det = dsub(dmul(a00, a11), dmul(a01, a10));
    d6 r;
    if (-1e-10 < det.v && det.v < 1e-10) {
      r = dc(1.0);
    } else {
      const d6 l0 = ddiv(dsub(dmul(a11, b0), dmul(a01, b1v)), det), l1 = ddiv(dsub(dmul(a00, b1v), dmul(a10, b0)), det);
      d6 pt[3], yc[3], py[3];
      for (int a = 0; a < 3; a++) pt[a] = dmulc(dadd(dmul(l0, x[a]), dadd(tr[a], dmul(l1, ry[a]))), 0.5);
      const d6 npt = dsqrt(ddot3(pt, pt));
      for (int a = 0; a < 3; a++) yc[a] = dsub(pt[a], tr[a]);
      aa_rotate(rot, yc, py);
      const d6 npy = dsqrt(ddot3(py, py));
      const d6 s = dadd(ddiv(ddot3(pt, x), npt), ddiv(ddot3(py, y), npy));
      r = dsub(dc(1.0), dmulc(s, 0.5));
    }
    res[i] = r.v;
    if (jac)
      for (int k = 0; k < NPAR; k++) jac[NPAR * i + k] = r.d[k];
  }
  const d6 last = dsub(dc(1.0), dsqrt(ddot3(tr, tr)));
  res[MAX_ERRORS] = last.v;
  if (jac)
    for (int k = 0; k < NPAR; k++) jac[NPAR * MAX_ERRORS + k] = last.d[k];
}
void oracle_relpose_cost(const double *b1, const double *b2, int n, const double *par, double *res, double *jac) {
  refine_cost C;
  C.b1 = b1;
  C.b2 = b2;
  glibc_rand_t g;
  glibc_srand(&g, 42u);
  for (int i = 0; i < MAX_ERRORS; i++) {
    int idx = (int)(((float)glibc_rand(&g) / (float)2147483647) * (float)n);
    if (idx >= n) idx = n - 1; /* rand() == RAND_MAX would index one past the end in the reference */
    C.picked[i] = idx;
  }
  cost_eval(&C, par, res, jac);
}

/* ceres rotation.h: RotationMatrixToAngleAxis (through the quaternion) and AngleAxisToRotationMatrix; R row-major here */
static void rotmat_to_aa(const double *R, double *aa) {
  double q[4];
  const double trace = R[0] + R[4] + R[8];
  if (trace >= 0.0) {
    double t = sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t;
    q[2] = (R[2] - R[6]) * t;
    q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j + 1] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k + 1] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > 0.0) {
    const double s = sqrt(s2), c = q[0];
    const double two_theta = 2.0 * (c < 0.0 ? atan2(-s, -c) : atan2(s, c));
    const double k = two_theta / s;
    for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * k;
  } else {
    for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * 2.0;
  }
}
static void aa_to_rotmat(const double *aa, double *R) {
  const double th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2 > 2.220446049250313e-16) {
    const double th = sqrt(th2), wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th, c = cos(th), s = sin(th);
    R[0] = c + wx * wx * (1 - c);      R[1] = wx * wy * (1 - c) - wz * s; R[2] = wy * s + wx * wz * (1 - c);
    R[3] = wz * s + wx * wy * (1 - c); R[4] = c + wy * wy * (1 - c);      R[5] = -wx * s + wy * wz * (1 - c);
    R[6] = -wy * s + wx * wz * (1 - c); R[7] = wx * s + wy * wz * (1 - c); R[8] = c + wz * wz * (1 - c);
  } else {
    R[0] = 1; R[1] = -aa[2]; R[2] = aa[1];
    R[3] = aa[2]; R[4] = 1; R[5] = -aa[0];
    R[6] = -aa[1]; R[7] = aa[0]; R[8] = 1;
  }
}

/* 6 x 6 SPD-ish solve by LDL^T without pivoting (Eigen::LDLT pivots; for these well-scaled systems the result agrees to rounding) */
static int ldlt_solve6(const double *A, const double *b, double *x) {
  double L[36], D[6];
  memset(L, 0, sizeof(L));
  for (int j = 0; j < 6; j++) {
    double d = A[7 * j];
    for (int k = 0; k < j; k++) d -= L[6 * j + k] * L[6 * j + k] * D[k];
    if (d == 0.0 || !isfinite(d)) return 0;
    D[j] = d;
    L[7 * j] = 1.0;
    for (int i = j + 1; i < 6; i++) {
      double v = A[6 * i + j];
      for (int k = 0; k < j; k++) v -= L[6 * i + k] * L[6 * j + k] * D[k];
      L[6 * i + j] = v / d;
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) {
    double v = b[i];
    for (int k = 0; k < i; k++) v -= L[6 * i + k] * y[k];
    y[i] = v;
  }
  for (int i = 0; i < 6; i++) y[i] /= D[i];
  for (int i = 5; i >= 0; i--) {
    double v = y[i];
    for (int k = i + 1; k < 6; k++) v -= L[6 * k + i] * x[k];
    x[i] = v;
  }
  return 1;
}

/* RT (3 x 4 row-major, x2 ~ R x1 + t) refined in place; returns the number of TinySolver iterations.
 * costs[2] = initial / final cost (sum of squares / 2). */
int oracle_relative_pose_refinement(double *RT, const double *b1, const double *b2, int n, int iterations, double *costs) {
  refine_cost C;
  C.b1 = b1;
  C.b2 = b2;
  {
    glibc_rand_t g;
    glibc_srand(&g, 42u);
    for (int i = 0; i < MAX_ERRORS; i++) {
      int idx = (int)(((float)glibc_rand(&g) / (float)2147483647) * (float)n);
      if (idx >= n) idx = n - 1;
      C.picked[i] = idx;
    }
  }
  double R[9], x[6];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) R[3 * a + b] = RT[4 * a + b];
  rotmat_to_aa(R, x);
  for (int a = 0; a < 3; a++) x[3 + a] = -(R[a] * RT[3] + R[3 + a] * RT[7] + R[6 + a] * RT[11]); /* -R^T t */
  enum { NR = MAX_ERRORS + 1 };
  double err[NR], J[NR * 6], fnew[NR], scal[6], jtj[36], g[6], cost = 0;
  int have_scale = 0, it = 0;
#define UPDATE()                                                                              \
  {                                                                                           \
    cost_eval(&C, x, err, J);                                                                 \
    for (int i = 0; i < NR; i++) err[i] = -err[i];                                            \
    if (!have_scale) {                                                                        \
      for (int k = 0; k < 6; k++) {                                                           \
        double s = 0;                                                                         \
        for (int i = 0; i < NR; i++) s += J[6 * i + k] * J[6 * i + k];                        \
        scal[k] = 1.0 / (1.0 + sqrt(s));                                                      \
      }                                                                                       \
      have_scale = 1;                                                                         \
    }                                                                                         \
    for (int i = 0; i < NR; i++)                                                              \
      for (int k = 0; k < 6; k++) J[6 * i + k] *= scal[k];                                    \
    for (int a = 0; a < 6; a++) {                                                             \
      g[a] = 0;                                                                               \
      for (int i = 0; i < NR; i++) g[a] += J[6 * i + a] * err[i];                             \
      for (int b = 0; b < 6; b++) {                                                           \
        double s = 0;                                                                         \
        for (int i = 0; i < NR; i++) s += J[6 * i + a] * J[6 * i + b];                        \
        jtj[6 * a + b] = s;                                                                   \
      }                                                                                       \
    }                                                                                         \
    cost = 0;                                                                                 \
    for (int i = 0; i < NR; i++) cost += err[i] * err[i];                                     \
    cost *= 0.5;                                                                              \
  }
  UPDATE()
  if (costs) costs[0] = cost;
  double gmax = 0;
  for (int k = 0; k < 6; k++) gmax = fmax(gmax, fabs(g[k]));
  double u = 1.0 / 1e4, v = 2.0;
  if (!(gmax < 1e-10) && !(cost < 2.220446049250313e-16)) {
    for (it = 1; it < iterations; it++) {
      double reg[36], step[6], dx[6], xn[6];
      memcpy(reg, jtj, sizeof(reg));
      for (int k = 0; k < 6; k++) reg[7 * k] += u * fmin(fmax(jtj[7 * k], 1e-6), 1e32);
      if (!ldlt_solve6(reg, g, step)) { u *= v; v *= 2; continue; }
      double dxn = 0, xnorm = 0;
      for (int k = 0; k < 6; k++) {
        dx[k] = scal[k] * step[k];
        dxn += dx[k] * dx[k];
        xnorm += x[k] * x[k];
      }
      if (sqrt(dxn) < 1e-8 * (sqrt(xnorm) + 1e-8)) break; /* RELATIVE_STEP_SIZE_TOO_SMALL */
      for (int k = 0; k < 6; k++) xn[k] = x[k] + dx[k];
      cost_eval(&C, xn, fnew, NULL);
      double fn2 = 0;
      for (int i = 0; i < NR; i++) fn2 += fnew[i] * fnew[i];
      const double cost_change = 2.0 * cost - fn2;
      double mc = 0;
      for (int a = 0; a < 6; a++) {
        double s = 2.0 * g[a];
        for (int b = 0; b < 6; b++) s -= jtj[6 * a + b] * step[b];
        mc += step[a] * s;
      }
      const double rho = cost_change / mc;
      if (rho > 0) {
        memcpy(x, xn, sizeof(xn));
        UPDATE()
        gmax = 0;
        for (int k = 0; k < 6; k++) gmax = fmax(gmax, fabs(g[k]));
        if (gmax < 1e-10 || cost < 2.220446049250313e-16) { it++; break; }
        const double tmp = 2.0 * rho - 1.0;
        u = u * fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp);
        v = 2.0;
        continue;
      }
      u *= v;
      v *= 2.0;
    }
  }
#undef UPDATE
  if (costs) costs[1] = cost;
  aa_to_rotmat(x, R);
  for (int a = 0; a < 3; a++) {
    for (int b = 0; b < 3; b++) RT[4 * a + b] = R[3 * a + b];
    RT[4 * a + 3] = -(R[3 * a] * x[3] + R[3 * a + 1] * x[4] + R[3 * a + 2] * x[5]);
  }
  return it;
}

/* ---- Stage 6: robust_match_calibrated on bearings, end to end (opensfm/matching.py:871-903 with
 * multiview.relative_pose_ransac / relative_pose_optimize_nonlinear, opensfm/multiview.py:494-553).
 * The small matrix conversions are written out (left-to-right sums) so that another implementation can match bits.
 * Returns the number of inliers (0 = the reference returns an empty array); mask[n]; R, t = second camera in the first. */
int oracle_ransac_relative_pose(const double *b1, const double *b2, int n, double threshold_angle, int iterations, double probability,
                                int use_lo, int lo_iterations, double *model, double *lo_model, int *inliers, int *iters_run);
void oracle_inliers_bearings(const double *b1, const double *b2, int n, const double *R, const double *t, double threshold, uint8_t *mask);
int oracle_robust_match_calibrated(const double *b1, const double *b2, int n, double threshold, int iterations, double probability,
                                   int use_lo, int lo_iterations, int refine_iterations, double *R, double *t, uint8_t *mask,
                                   double *ransac_models /* 24 */, int *ransac_info /* score, iterations */) {
  memset(mask, 0, (size_t)(n > 0 ? n : 0));
  memset(R, 0, 9 * sizeof(double));
  memset(t, 0, 3 * sizeof(double));
  if (ransac_models) memset(ransac_models, 0, 24 * sizeof(double));
  if (ransac_info) ransac_info[0] = ransac_info[1] = 0;
  if (n < 8) return 0;
  double model[12], lo[12];
  int *inl = (int *)__builtin_alloca(sizeof(int) * (size_t)n), iters = 0;
  const int score = oracle_ransac_relative_pose(b1, b2, n, threshold, iterations, probability, use_lo, lo_iterations, model, lo, inl, &iters);
  if (ransac_models) { memcpy(ransac_models, model, sizeof(model)); memcpy(ransac_models + 12, lo, sizeof(lo)); }
  if (ransac_info) { ransac_info[0] = score; ransac_info[1] = iters; }
  for (int a = 0; a < 3; a++) {
    for (int b = 0; b < 3; b++) R[3 * a + b] = lo[4 * b + a];
    t[a] = -(lo[a] * lo[3] + lo[4 + a] * lo[7] + lo[8 + a] * lo[11]);
  }
  double *s1 = (double *)__builtin_alloca(sizeof(double) * 3 * (size_t)n), *s2 = (double *)__builtin_alloca(sizeof(double) * 3 * (size_t)n);
  const double relax[3] = {4.0, 2.0, 1.0};
  for (int stage = 0; stage < 3; stage++) {
    oracle_inliers_bearings(b1, b2, n, R, t, relax[stage] * threshold, mask);
    int cnt = 0;
    for (int i = 0; i < n; i++)
      if (mask[i]) {
        memcpy(s1 + 3 * cnt, b1 + 3 * i, 3 * sizeof(double));
        memcpy(s2 + 3 * cnt, b2 + 3 * i, 3 * sizeof(double));
        cnt++;
      }
    if (cnt < 8) { memset(mask, 0, (size_t)n); return 0; }
    double RT[12];
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) RT[4 * a + b] = R[3 * b + a];
      RT[4 * a + 3] = -(R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2]);
    }
    oracle_relative_pose_refinement(RT, s1, s2, cnt, refine_iterations, NULL);
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) R[3 * a + b] = RT[4 * b + a];
      t[a] = -(RT[a] * RT[3] + RT[4 + a] * RT[7] + RT[8 + a] * RT[11]);
    }
  }
  oracle_inliers_bearings(b1, b2, n, R, t, threshold, mask);
  int cnt = 0;
  for (int i = 0; i < n; i++) cnt += mask[i];
  return cnt;
}
