/*
 * ransac_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never on the product path).
 *
 * CPU restatement of the geometric-verification step of OpenSfM's pair matching:
 *
 *   opensfm/matching.py:780-802  robust_match_fundamental
 *        F, mask = cv2.findFundamentalMat(p1, p2, cv2.FM_RANSAC, threshold=0.004, 0.9999)
 *        reject when F is None or F[2,2] == 0.0; return matches[mask]
 *
 * The arithmetic lives in OpenCV (opencv-python>=4.8, pyproject.toml:31; calib3d/src/fundam.cpp,
 * calib3d/src/ptsetreg.cpp), which is NOT under /root/reference and cannot be built or imported
 * here.  We restate its published algorithm (from documentation/memory):
 *   - points are converted to float32 (CV_32FC2) before estimation, maths in double;
 *   - npoints < 7 -> no model; FM_RANSAC with npoints >= 15 -> RANSACPointSetRegistrator
 *     (modelPoints = 7, maxIters = 1000), else LMedS;
 *   - cv::RNG seeded with (uint64)-1, multiply-with-carry A = 4164903690, uniform(0,n) = next % n;
 *   - getSubset: draw 7 distinct indices, reject the subset if the LAST point is (nearly)
 *     collinear with any two earlier ones in either image (haveCollinearPoints), <= 10000 tries;
 *   - runKernel = 7-point algorithm: 2-dim null space of the 7x9 epipolar system, cubic
 *     det(lambda*U + W) = 0, each real root gives a model normalised to F[8] = 1;
 *   - error = max of the two squared point-to-epipolar-line distances, cast to float32,
 *     inlier iff err <= (float)(thr*thr);
 *   - a model replaces the best iff goodCount > max(bestCount, 6); then
 *     niters = RANSACUpdateNumIters(conf, (n - good)/n, 7, niters).
 * The 7-point kernel follows cv2's ROUTE, because a model only replaces the best on strictly more inliers and so the ORDER in which
 * the <= 3 solutions are scored decides ties:
 *   - the basis (f1, f2) of the null space is the one SVDecomp(A, FULL_UV) hands run7Point: JacobiSVDImpl_ fills the two right
 *     singular vectors a 7x9 matrix does not determine with two fixed pseudo-random vectors (entries +-1/9, the sign = bit 8 of
 *     successive draws of cv::RNG(0x12345678)), projected onto the orthogonal complement of the vectors found so far and
 *     normalised -- i.e. f1 = P r1 / |P r1|, f2 = the same for r2 after removing its f1 component (P = projector onto the null
 *     space).  That basis depends only on the null space, so it is reproduced here from ANY null-space basis (Gauss-Jordan with
 *     complete pivoting, then Gram-Schmidt) up to rounding;
 *   - lambda parametrises F = lambda (f1 - f2) + f2; the roots are scored in solveCubic's output order: three distinct real roots
 *     come out as smallest, largest, middle (x_k = -2 sqrt(Q) cos((theta + 2 pi k) / 3) - a1 / 3), a simple + a double root as
 *     (simple, double), the quadratic case (leading coefficient exactly 0) larger-magnitude root first.
 * What is NOT restated from OpenCV: the root VALUES come from bisection + deflation (OpenCV: the trigonometric closed form) and the
 * projector from Gauss-Jordan (OpenCV: the seven Jacobi-rotated rows); log() in RANSACUpdateNumIters is an explicit atanh series.
 * All of it uses only + - * / sqrt so that the HIP path can reproduce every bit (both sides are compiled with
 * -ffp-contract=off).  Same models in the same order, rounding differs; a rank-deficient 7x9 system (where JacobiSVD would also
 * regenerate some of the first seven vectors) gives no model here.
 * PARITY STATUS: "parity unpinned" -- the reference holds no golden vector for this call
 *  and cv2 is absent; inlier-set identity is defined GPU == this oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MODEL_POINTS 7

/* ---- cv::RNG (multiply with carry) ---- */
typedef struct {
  uint64_t state;
} cvrng_t;
static inline unsigned cvrng_next(cvrng_t *r) {
  r->state = (uint64_t)(unsigned)r->state * 4164903690ULL + (unsigned)(r->state >> 32);
  return (unsigned)r->state;
}
static inline int cvrng_uniform(cvrng_t *r, int a, int b) {
  return a == b ? a : (int)(cvrng_next(r) % (unsigned)(b - a) + (unsigned)a);
}

/* ---- deterministic natural log (x > 0), atanh series; bit-reproducible on GPU ---- */
static double det_log(double x) {
  int e;
  double m = frexp(x, &e); /* m in [0.5, 1) */
  if (m < 0.70710678118654752440) {
    m = m * 2.0;
    e -= 1;
  }
  double t = (m - 1.0) / (m + 1.0);
  double t2 = t * t;
  double s = 1.0 / 27.0;
  s = s * t2 + 1.0 / 25.0;
  s = s * t2 + 1.0 / 23.0;
  s = s * t2 + 1.0 / 21.0;
  s = s * t2 + 1.0 / 19.0;
  s = s * t2 + 1.0 / 17.0;
  s = s * t2 + 1.0 / 15.0;
  s = s * t2 + 1.0 / 13.0;
  s = s * t2 + 1.0 / 11.0;
  s = s * t2 + 1.0 / 9.0;
  s = s * t2 + 1.0 / 7.0;
  s = s * t2 + 1.0 / 5.0;
  s = s * t2 + 1.0 / 3.0;
  s = s * t2 + 1.0;
  return (double)e * 0x1.62e42fefa39efp-1 + 2.0 * t * s;
}

/* RANSACUpdateNumIters (ptsetreg.cpp) */
static int update_num_iters(double p, double ep, int model_points, int max_iters) {
  if (p < 0.) p = 0.;
  if (p > 1.) p = 1.;
  if (ep < 0.) ep = 0.;
  if (ep > 1.) ep = 1.;
  double num = 1. - p;
  if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
  double w = 1. - ep, w2 = w * w, w4 = w2 * w2;
  double wn = w4 * w2 * w; /* model_points == 7 */
  (void)model_points;
  double denom = 1. - wn;
  if (denom < 2.2250738585072014e-308) return 0;
  num = det_log(num);
  denom = det_log(denom);
  if (denom >= 0 || -num >= max_iters * (-denom)) return max_iters;
  return (int)rint(num / denom);
}

static inline double det3(const double *a, const double *b, const double *c) {
  return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) +
         a[2] * (b[0] * c[1] - b[1] * c[0]);
}

/* monic cubic x^3 + a x^2 + b x + c: real roots, bisection + deflation. returns count. */
static int solve_cubic_monic(double a, double b, double c, double *roots) {
  double R = fabs(a);
  if (fabs(b) > R) R = fabs(b);
  if (fabs(c) > R) R = fabs(c);
  R = R + 1.0;
  if (!(R < 1e300)) return 0;
  double lo = -R, hi = R;
  for (int it = 0; it < 2200; it++) {
    double mid = 0.5 * (lo + hi);
    if (!(mid > lo && mid < hi)) break;
    double pm = ((mid + a) * mid + b) * mid + c;
    if (pm > 0)
      hi = mid;
    else
      lo = mid;
  }
  double plo = ((lo + a) * lo + b) * lo + c, phi = ((hi + a) * hi + b) * hi + c;
  double r = (fabs(plo) <= fabs(phi)) ? lo : hi;
  int n = 0;
  roots[n++] = r;
  double p = a + r;
  double q = b + p * r;
  double disc = p * p - 4.0 * q;
  if (disc > 0) {
    double sq = sqrt(disc);
    double t = (p >= 0) ? -0.5 * (p + sq) : -0.5 * (p - sq);
    roots[n++] = t;
    if (t != 0) roots[n++] = q / t;
  } else if (disc == 0) {
    roots[n++] = -0.5 * p;
  }
  return n;
}

/* sign pattern of the two vectors JacobiSVDImpl_ starts the missing singular vectors from: cv::RNG(0x12345678), bit 8 of a draw */
static void cv_svd_fill_signs(double r[2][9]) {
  cvrng_t rng = {0x12345678ULL};
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < 9; k++) r[i][k] = (cvrng_next(&rng) & 256) != 0 ? 1.0 / 9.0 : -(1.0 / 9.0);
}
void oracle_cv_svd_fill_signs(double *r18) { cv_svd_fill_signs((double(*)[9])r18); }

static double dot9(const double *a, const double *b) {
  double s = 0.0;
  for (int i = 0; i < 9; i++) s = s + a[i] * b[i];
  return s;
}

/* (v1, v2): any basis of the null space -> (f1, f2): the basis cv2's SVDecomp(FULL_UV) produces.  0 when degenerate. */
static int cv_null_basis(const double *v1, const double *v2, double *f1, double *f2) {
  double r[2][9], n1[9], n2[9];
  cv_svd_fill_signs(r);
  double s = sqrt(dot9(v1, v1));
  if (!(s > 1e-300)) return 0;
  for (int i = 0; i < 9; i++) n1[i] = v1[i] / s;
  double d = dot9(v2, n1);
  for (int i = 0; i < 9; i++) n2[i] = v2[i] - d * n1[i];
  s = sqrt(dot9(n2, n2));
  if (!(s > 1e-300)) return 0;
  for (int i = 0; i < 9; i++) n2[i] = n2[i] / s;
  double a = dot9(r[0], n1), b = dot9(r[0], n2);
  for (int i = 0; i < 9; i++) f1[i] = a * n1[i] + b * n2[i];
  s = sqrt(dot9(f1, f1));
  if (!(s > 1e-12)) return 0;
  for (int i = 0; i < 9; i++) f1[i] = f1[i] / s;
  a = dot9(r[1], n1);
  b = dot9(r[1], n2);
  for (int i = 0; i < 9; i++) f2[i] = a * n1[i] + b * n2[i];
  d = dot9(f2, f1);
  for (int i = 0; i < 9; i++) f2[i] = f2[i] - d * f1[i];
  s = sqrt(dot9(f2, f2));
  if (!(s > 1e-12)) return 0;
  for (int i = 0; i < 9; i++) f2[i] = f2[i] / s;
  return 1;
}

/* three distinct real roots in solveCubic's order: smallest, largest, middle */
static void cv_root_order3(double *r) {
  double lo = r[0], hi = r[0];
  for (int k = 1; k < 3; k++) {
    if (r[k] < lo) lo = r[k];
    if (r[k] > hi) hi = r[k];
  }
  double mid = r[0];
  for (int k = 0; k < 3; k++)
    if (r[k] != lo && r[k] != hi) mid = r[k];
  r[0] = lo;
  r[1] = hi;
  r[2] = mid;
}

/* 7-point algorithm.  m1, m2: 7 points (x,y).  F: up to 3 models x 9.  returns #models. */
static int run_7point(const double *m1, const double *m2, double *F) {
  double A[7][9];
  for (int i = 0; i < 7; i++) {
    double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
    A[i][0] = x1 * x0;
    A[i][1] = x1 * y0;
    A[i][2] = x1;
    A[i][3] = y1 * x0;
    A[i][4] = y1 * y0;
    A[i][5] = y1;
    A[i][6] = x0;
    A[i][7] = y0;
    A[i][8] = 1.0;
  }
  int colperm[9];
  for (int c = 0; c < 9; c++) colperm[c] = c;
  /* Gauss-Jordan, complete pivoting */
  for (int k = 0; k < 7; k++) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int r = k; r < 7; r++)
      for (int c = k; c < 9; c++) {
        double v = fabs(A[r][c]);
        if (v > best) {
          best = v;
          pr = r;
          pc = c;
        }
      }
    if (!(best > 1e-300)) return 0; /* rank deficient */
    if (pr != k)
      for (int c = 0; c < 9; c++) {
        double t = A[k][c];
        A[k][c] = A[pr][c];
        A[pr][c] = t;
      }
    if (pc != k) {
      for (int r = 0; r < 7; r++) {
        double t = A[r][k];
        A[r][k] = A[r][pc];
        A[r][pc] = t;
      }
      int t = colperm[k];
      colperm[k] = colperm[pc];
      colperm[pc] = t;
    }
    double inv = 1.0 / A[k][k];
    for (int c = 0; c < 9; c++) A[k][c] = A[k][c] * inv;
    for (int r = 0; r < 7; r++) {
      if (r == k) continue;
      double f = A[r][k];
      for (int c = 0; c < 9; c++) A[r][c] = A[r][c] - f * A[k][c];
    }
  }
  /* null vectors in permuted order: v1 free col 7, v2 free col 8 */
  double v1[9], v2[9];
  for (int k = 0; k < 7; k++) {
    v1[colperm[k]] = -A[k][7];
    v2[colperm[k]] = -A[k][8];
  }
  v1[colperm[7]] = 1.0;
  v1[colperm[8]] = 0.0;
  v2[colperm[7]] = 0.0;
  v2[colperm[8]] = 1.0;
  double U[9], W[9], f1[9];
  if (!cv_null_basis(v1, v2, f1, W)) return 0;
  for (int i = 0; i < 9; i++) U[i] = f1[i] - W[i];
  /* det(lambda*U + W) = a3 l^3 + a2 l^2 + a1 l + a0 */
  double a3 = det3(U, U + 3, U + 6);
  double a0 = det3(W, W + 3, W + 6);
  double a2 = det3(W, U + 3, U + 6) + det3(U, W + 3, U + 6) + det3(U, U + 3, W + 6);
  double a1 = det3(U, W + 3, W + 6) + det3(W, U + 3, W + 6) + det3(W, W + 3, U + 6);
  double roots[3];
  int nr = 0;
  if (a3 != 0) {
    nr = solve_cubic_monic(a2 / a3, a1 / a3, a0 / a3, roots);
    if (nr == 3) {
      if (roots[0] != roots[1] && roots[1] != roots[2] && roots[0] != roots[2])
        cv_root_order3(roots);
      else
        nr = 1; /* solveCubic reports three roots only when they are distinct (d > 0); a numerically repeated root is its n = 1 */
    }
  } else if (a2 != 0) { /* solveCubic's quadratic branch: q1 = (-b + d) / 2, q2 = -(b + d) / 2, the larger |q| first */
    double disc = a1 * a1 - 4.0 * a2 * a0;
    if (disc >= 0) {
      double sq = sqrt(disc);
      double q1 = (-a1 + sq) * 0.5, q2 = (a1 + sq) * -0.5;
      double q = fabs(q1) > fabs(q2) ? q1 : q2;
      roots[nr++] = q / a2;
      if (disc > 0) roots[nr++] = a0 / q;
    }
  } else if (a1 != 0) {
    roots[nr++] = -a0 / a1;
  }
  int n = 0;
  for (int k = 0; k < nr; k++) {
    double lambda = roots[k], mu = 1.0;
    double s = U[8] * lambda + W[8];
    double *Fk = F + 9 * n;
    if (fabs(s) > 2.220446049250313e-16) {
      mu = 1.0 / s;
      lambda = lambda * mu;
      Fk[8] = 1.0;
    } else {
      Fk[8] = 0.0;
    }
    int ok = 1;
    for (int i = 0; i < 8; i++) {
      Fk[i] = U[i] * lambda + W[i] * mu;
      if (!(fabs(Fk[i]) < 1e300)) ok = 0;
    }
    if (ok) n++;
  }
  return n;
}

/* FMEstimatorCallback::computeError for one correspondence */
static inline float epi_error(const double *F, double x1, double y1, double x2, double y2) {
  double a, b, c, d1, d2, s1, s2;
  a = F[0] * x1 + F[1] * y1 + F[2];
  b = F[3] * x1 + F[4] * y1 + F[5];
  c = F[6] * x1 + F[7] * y1 + F[8];
  s2 = 1. / (a * a + b * b);
  d2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6];
  b = F[1] * x2 + F[4] * y2 + F[7];
  c = F[2] * x2 + F[5] * y2 + F[8];
  s1 = 1. / (a * a + b * b);
  d1 = x1 * a + y1 * b + c;
  double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
  return (float)((e1 < e2) ? e2 : e1); /* std::max(d1*d1*s1, d2*d2*s2) == (a<b)?b:a */
}

static int have_collinear(const double *m, int count) {
  int i = count - 1;
  for (int j = 0; j < i; j++) {
    double dx1 = m[2 * j] - m[2 * i];
    double dy1 = m[2 * j + 1] - m[2 * i + 1];
    for (int k = 0; k < j; k++) {
      double dx2 = m[2 * k] - m[2 * i];
      double dy2 = m[2 * k + 1] - m[2 * i + 1];
      if (fabs(dx2 * dy1 - dy2 * dx1) <=
          1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
        return 1;
    }
  }
  return 0;
}

static int get_subset(const double *m1, const double *m2, int count, cvrng_t *rng, int max_attempts,
                      double *ms1, double *ms2, int *idx) {
  for (int iters = 0; iters < max_attempts; ++iters) {
    for (int i = 0; i < MODEL_POINTS; ++i) {
      int idx_i;
      for (;;) {
        idx_i = cvrng_uniform(rng, 0, count);
        int dup = 0;
        for (int j = 0; j < i; j++)
          if (idx[j] == idx_i) dup = 1;
        if (!dup) break;
      }
      idx[i] = idx_i;
      ms1[2 * i] = m1[2 * idx_i];
      ms1[2 * i + 1] = m1[2 * idx_i + 1];
      ms2[2 * i] = m2[2 * idx_i];
      ms2[2 * i + 1] = m2[2 * idx_i + 1];
    }
    if (!have_collinear(ms1, MODEL_POINTS) && !have_collinear(ms2, MODEL_POINTS)) return 1;
  }
  return 0;
}

/*
 * cv2.findFundamentalMat(p1, p2, FM_RANSAC, thr, conf) for n >= 15 (RANSAC branch).
 * p1, p2: n x 2 doubles.  Returns 1 and fills F (row-major 3x3), mask (n bytes), *iters_run when
 * a model was found; 0 when F would be None (mask zeroed).  n < 15 returns -1 (LMedS branch, not
 * restated in this round).
 */
/* cv2.findFundamentalMat(FM_RANSAC) with 8 <= n < 15 correspondences silently switches to the LMedS
 * registrator (fundam.cpp: `(method & ~3) == FM_RANSAC && npoints >= 15` else createLMeDSPointSetRegistrator).
 * Restated from the published algorithm (ptsetreg.cpp, LMeDSPointSetRegistrator::run): a FIXED number of
 * iterations niters = max(RANSACUpdateNumIters(confidence, 0.45, 7, maxIters), 3); per hypothesis the
 * median (element count/2 of the sorted float errors); the model with the smallest median wins (strict <);
 * then sigma = 2.5*1.4826*(1 + 5/(count-7))*sqrt(minMedian), clamped to >= 0.001, inliers = err <= sigma^2,
 * success iff at least 7 inliers.  parity unpinned vs cv2 (no golden vectors; same caveats as RANSAC). */
static int find_fundamental_lmeds(const double *p1, const double *p2, int n, double conf, int max_iters, double *F,
                                  uint8_t *mask, int *iters_run) {
  double m1[2 * 16], m2[2 * 16];
  for (int i = 0; i < 2 * n; i++) {
    m1[i] = (double)(float)p1[i];
    m2[i] = (double)(float)p2[i];
  }
  if (conf < 2.220446049250313e-16 || conf > 1 - 2.220446049250313e-16) conf = 0.99;
  cvrng_t rng = {(uint64_t)-1};
  int niters = update_num_iters(conf, 0.45, MODEL_POINTS, max_iters);
  if (niters < 3) niters = 3;
  double min_median = 1.7976931348623157e308;
  double best[9], models[27], ms1[14], ms2[14];
  int idx[MODEL_POINTS];
  int iter;
  for (iter = 0; iter < niters; iter++) {
    if (!get_subset(m1, m2, n, &rng, 10000, ms1, ms2, idx)) {
      if (iter == 0) return 0;
      break;
    }
    const int nm = run_7point(ms1, ms2, models);
    for (int k = 0; k < nm; k++) {
      float e[16];
      for (int i = 0; i < n; i++) e[i] = epi_error(models + 9 * k, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]);
      for (int i = 1; i < n; i++) { /* std::nth_element(count/2): the value at that rank */
        const float v = e[i];
        int j = i - 1;
        for (; j >= 0 && e[j] > v; j--) e[j + 1] = e[j];
        e[j + 1] = v;
      }
      const double median = (double)e[n / 2];
      if (median < min_median) {
        min_median = median;
        memcpy(best, models + 9 * k, sizeof(best));
      }
    }
  }
  if (iters_run) *iters_run = iter;
  if (!(min_median < 1.7976931348623157e308)) return 0;
  double sigma = 2.5 * 1.4826 * (1 + 5. / (n - MODEL_POINTS)) * sqrt(min_median);
  if (sigma < 0.001) sigma = 0.001;
  const float t = (float)(sigma * sigma);
  int count = 0;
  for (int i = 0; i < n; i++) {
    mask[i] = epi_error(best, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]) <= t;
    count += mask[i];
  }
  memcpy(F, best, sizeof(best));
  if (count < MODEL_POINTS) {
    memset(mask, 0, (size_t)n);
    return 0;
  }
  return 1;
}

int oracle_find_fundamental_ransac(const double *p1, const double *p2, int n, double thr,
                                   double conf, int max_iters, double *F, uint8_t *mask,
                                   int *iters_run) {
  if (iters_run) *iters_run = 0;
  memset(mask, 0, (size_t)(n > 0 ? n : 0));
  if (n >= 8 && n < 15) return find_fundamental_lmeds(p1, p2, n, conf, max_iters, F, mask, iters_run);
  if (n < 15) return -1; /* n == 7: cv2 returns all 7-point solutions stacked; the reference never gets here (matching.py:787) */
  double *m1 = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  double *m2 = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  for (int i = 0; i < 2 * n; i++) {
    m1[i] = (double)(float)p1[i];
    m2[i] = (double)(float)p2[i];
  }
  if (thr <= 0) thr = 3;
  if (conf < 2.220446049250313e-16 || conf > 1 - 2.220446049250313e-16) conf = 0.99;
  const float t = (float)(thr * thr);
  cvrng_t rng = {(uint64_t)-1};
  int niters = max_iters > 1 ? max_iters : 1;
  int max_good = 0;
  double best[9];
  double models[27], ms1[14], ms2[14];
  int idx[MODEL_POINTS];
  int iter;
  for (iter = 0; iter < niters; iter++) {
    int found = get_subset(m1, m2, n, &rng, 10000, ms1, ms2, idx);
    if (!found) {
      if (iter == 0) {
        free(m1);
        free(m2);
        return 0;
      }
      break;
    }
    int nm = run_7point(ms1, ms2, models);
    for (int k = 0; k < nm; k++) {
      const double *Fk = models + 9 * k;
      int good = 0;
      for (int i = 0; i < n; i++)
        good += epi_error(Fk, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]) <= t;
      int lim = max_good > MODEL_POINTS - 1 ? max_good : MODEL_POINTS - 1;
      if (good > lim) {
        memcpy(best, Fk, sizeof(best));
        max_good = good;
        niters = update_num_iters(conf, (double)(n - good) / n, MODEL_POINTS, niters);
      }
    }
  }
  if (iters_run) *iters_run = iter;
  int ret = 0;
  if (max_good > 0) {
    memcpy(F, best, sizeof(best));
    for (int i = 0; i < n; i++)
      mask[i] = epi_error(best, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]) <= t;
    ret = 1;
  }
  free(m1);
  free(m2);
  return ret;
}

/* exposed for unit tests */
int oracle_run_7point(const double *m1, const double *m2, double *F) { return run_7point(m1, m2, F); }
double oracle_det_log(double x) { return det_log(x); }
int oracle_update_num_iters(double p, double ep, int max_iters) {
  return update_num_iters(p, ep, 7, max_iters);
}
unsigned oracle_cvrng_sequence(uint64_t seed, int count, unsigned *out) {
  cvrng_t r = {seed};
  unsigned last = 0;
  for (int i = 0; i < count; i++) out[i] = last = cvrng_next(&r);
  return last;
}
