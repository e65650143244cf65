// ransac.hip -- batched fundamental-matrix RANSAC for gfx950 (MI355X).
//
// Replaces cv2.findFundamentalMat(p1, p2, FM_RANSAC, 0.004, 0.9999) as called by
// robust_match_fundamental (opensfm/matching.py:780-802), plus the two min-match gates of
// matching.match (matching.py:590-598, 632-634), for every pair of a batch: one workgroup per
// image pair.  Results are bit-identical to the sequential algorithm (same RNG stream, same
// hypothesis order, same adaptive stopping rule), but executed batch-parallel:
//   1. lane 0 draws the next kBatch 7-point subsets from the cv::RNG stream (sequential, cheap);
//   2. kBatch lanes solve their 7-point problems in parallel (fp64 Gauss-Jordan + cubic);
//   3. every wavefront scores whole models: 64 correspondences per step, ballot + popcount
//      ("wavefront-per-model reduction");
//   4. lane 0 replays the batch in order, applying `good > max(best, 6)` and
//      RANSACUpdateNumIters, and stops exactly where the sequential loop would.
// All fp64 arithmetic is plain + - * / sqrt in a fixed order (this file is compiled with
// -ffp-contract=off), so it reproduces the CPU statement of the same algorithm bit for bit.
#include "osfm_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kBatch = 64;  // hypotheses generated per round
constexpr int kFirstBatch = 8;  // hypotheses of the first round (ransac_core)

struct CvRng {
  unsigned long long state;
  __device__ unsigned next() {
    state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  __device__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + (unsigned)a); }
};

__device__ double det_log(double x) {
  int e;
  double m = frexp(x, &e);
  if (m < 0.70710678118654752440) {
    m = m * 2.0;
    e -= 1;
  }
  const double t = (m - 1.0) / (m + 1.0);
  const double t2 = t * t;
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

__device__ int update_num_iters(double p, double ep, int max_iters) {
  if (p < 0.) p = 0.;
  if (p > 1.) p = 1.;
  if (ep < 0.) ep = 0.;
  if (ep > 1.) ep = 1.;
  double num = 1. - p;
  if (num < 2.2250738585072014e-308) num = 2.2250738585072014e-308;
  const double w = 1. - ep, w2 = w * w, w4 = w2 * w2;
  const double wn = w4 * w2 * w;
  double denom = 1. - wn;
  if (denom < 2.2250738585072014e-308) return 0;
  num = det_log(num);
  denom = det_log(denom);
  if (denom >= 0 || -num >= max_iters * (-denom)) return max_iters;
  return (int)rint(num / denom);
}

__device__ __forceinline__ double det3(const double *a, const double *b, const double *c) {
  return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
}

__device__ int solve_cubic_monic(double a, double b, double c, double *roots) {
  double R = fabs(a);
  if (fabs(b) > R) R = fabs(b);
  if (fabs(c) > R) R = fabs(c);
  R = R + 1.0;
  if (!(R < 1e300)) return 0;
  double lo = -R, hi = R;
  for (int it = 0; it < 2200; it++) {
    const double mid = 0.5 * (lo + hi);
    if (!(mid > lo && mid < hi)) break;
    const double pm = ((mid + a) * mid + b) * mid + c;
    if (pm > 0)
      hi = mid;
    else
      lo = mid;
  }
  const double plo = ((lo + a) * lo + b) * lo + c, phi = ((hi + a) * hi + b) * hi + c;
  const double r = (fabs(plo) <= fabs(phi)) ? lo : hi;
  int n = 0;
  roots[n++] = r;
  const double p = a + r;
  const double q = b + p * r;
  const double disc = p * p - 4.0 * q;
  if (disc > 0) {
    const double sq = sqrt(disc);
    const double t = (p >= 0) ? -0.5 * (p + sq) : -0.5 * (p - sq);
    roots[n++] = t;
    if (t != 0) roots[n++] = q / t;
  } else if (disc == 0) {
    roots[n++] = -0.5 * p;
  }
  return n;
}

// cv2's basis of the null space (fundam.cpp run7Point takes the last two rows of V from SVDecomp(A, FULL_UV), which JacobiSVDImpl_
// builds from two fixed pseudo-random +-1/9 vectors -- cv::RNG(0x12345678), bit 8 of a draw -- projected onto the complement of the
// computed singular vectors): f1 = P r1 / |P r1|, f2 likewise from r2 minus its f1 component.
struct CvSvdFill {
  double r[2][9];
  constexpr CvSvdFill() : r{} {
    unsigned long long state = 0x12345678ULL;
    for (int i = 0; i < 2; i++)
      for (int k = 0; k < 9; k++) {
        state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
        r[i][k] = ((unsigned)state & 256u) != 0 ? 1.0 / 9.0 : -(1.0 / 9.0);
      }
  }
};
__constant__ CvSvdFill kCvSvdFill{};

__device__ __forceinline__ double dot9(const double *a, const double *b) {
  double s = 0.0;
  for (int i = 0; i < 9; i++) s = s + a[i] * b[i];
  return s;
}

__device__ int cv_null_basis(const double *v1, const double *v2, double *f1, double *f2) {
  double n1[9], n2[9];
  double s = sqrt(dot9(v1, v1));
  if (!(s > 1e-300)) return 0;
  for (int i = 0; i < 9; i++) n1[i] = v1[i] / s;
  double d = dot9(v2, n1);
  for (int i = 0; i < 9; i++) n2[i] = v2[i] - d * n1[i];
  s = sqrt(dot9(n2, n2));
  if (!(s > 1e-300)) return 0;
  for (int i = 0; i < 9; i++) n2[i] = n2[i] / s;
  double a = dot9(kCvSvdFill.r[0], n1), b = dot9(kCvSvdFill.r[0], n2);
  for (int i = 0; i < 9; i++) f1[i] = a * n1[i] + b * n2[i];
  s = sqrt(dot9(f1, f1));
  if (!(s > 1e-12)) return 0;
  for (int i = 0; i < 9; i++) f1[i] = f1[i] / s;
  a = dot9(kCvSvdFill.r[1], n1);
  b = dot9(kCvSvdFill.r[1], n2);
  for (int i = 0; i < 9; i++) f2[i] = a * n1[i] + b * n2[i];
  d = dot9(f2, f1);
  for (int i = 0; i < 9; i++) f2[i] = f2[i] - d * f1[i];
  s = sqrt(dot9(f2, f2));
  if (!(s > 1e-12)) return 0;
  for (int i = 0; i < 9; i++) f2[i] = f2[i] / s;
  return 1;
}

// three distinct real roots in cv::solveCubic's output order: smallest, largest, middle
__device__ void cv_root_order3(double *r) {
  double lo = r[0], hi = r[0], mid = r[0];
  for (int k = 1; k < 3; k++) {
    if (r[k] < lo) lo = r[k];
    if (r[k] > hi) hi = r[k];
  }
  for (int k = 0; k < 3; k++)
    if (r[k] != lo && r[k] != hi) mid = r[k];
  r[0] = lo;
  r[1] = hi;
  r[2] = mid;
}

// 7-point algorithm.  The 7 x 9 system is eliminated with FULL pivoting (dynamic row / column indices): as a private array it lived
// in scratch memory and every access of the elimination was a dependent round trip through the vector memory path (~0.3 ms per solve,
// the latency that bound the whole RANSAC kernel).  It now lives in LDS, lane-minor (element e of lane l at priv[e * 64 + l]: any
// per-lane dynamic index is bank-conflict free): elements 0..62 = A, 63..71 = v1, 72..80 = v2; the column permutation in ipriv.
constexpr int kPrivDoubles = 81;
#define OSFM_A(r, c) priv[((r) * 9 + (c)) * 64]
__device__ int run_7point(const double *m1, const double *m2, double *F, double *priv, int *ipriv) {
  for (int i = 0; i < 7; i++) {
    const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
    OSFM_A(i, 0) = x1 * x0;
    OSFM_A(i, 1) = x1 * y0;
    OSFM_A(i, 2) = x1;
    OSFM_A(i, 3) = y1 * x0;
    OSFM_A(i, 4) = y1 * y0;
    OSFM_A(i, 5) = y1;
    OSFM_A(i, 6) = x0;
    OSFM_A(i, 7) = y0;
    OSFM_A(i, 8) = 1.0;
  }
  for (int c = 0; c < 9; c++) ipriv[c * 64] = c;
  for (int k = 0; k < 7; k++) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int r = k; r < 7; r++)
      for (int c = k; c < 9; c++) {
        const double v = fabs(OSFM_A(r, c));
        if (v > best) {
          best = v;
          pr = r;
          pc = c;
        }
      }
    if (!(best > 1e-300)) return 0;
    if (pr != k)
      for (int c = 0; c < 9; c++) {
        const double t = OSFM_A(k, c);
        OSFM_A(k, c) = OSFM_A(pr, c);
        OSFM_A(pr, c) = t;
      }
    if (pc != k) {
      for (int r = 0; r < 7; r++) {
        const double t = OSFM_A(r, k);
        OSFM_A(r, k) = OSFM_A(r, pc);
        OSFM_A(r, pc) = t;
      }
      const int t = ipriv[k * 64];
      ipriv[k * 64] = ipriv[pc * 64];
      ipriv[pc * 64] = t;
    }
    const double inv = 1.0 / OSFM_A(k, k);
    double rowk[9];
#pragma unroll
    for (int c = 0; c < 9; c++) {
      rowk[c] = OSFM_A(k, c) * inv;
      OSFM_A(k, c) = rowk[c];
    }
    for (int r = 0; r < 7; r++) {
      if (r == k) continue;
      const double f = OSFM_A(r, k);
#pragma unroll
      for (int c = 0; c < 9; c++) OSFM_A(r, c) = OSFM_A(r, c) - f * rowk[c];
    }
  }
  double *v1p = priv + 63 * 64, *v2p = priv + 72 * 64;
  for (int k = 0; k < 7; k++) {
    const int cp = ipriv[k * 64];
    v1p[cp * 64] = -OSFM_A(k, 7);
    v2p[cp * 64] = -OSFM_A(k, 8);
  }
  {
    const int c7 = ipriv[7 * 64], c8 = ipriv[8 * 64];
    v1p[c7 * 64] = 1.0;
    v1p[c8 * 64] = 0.0;
    v2p[c7 * 64] = 0.0;
    v2p[c8 * 64] = 1.0;
  }
  double v1[9], v2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    v1[i] = v1p[i * 64];
    v2[i] = v2p[i * 64];
  }
  double U[9], W[9];
  if (!cv_null_basis(v1, v2, U, W)) return 0;
  for (int i = 0; i < 9; i++) U[i] = U[i] - W[i];
  const double a3 = det3(U, U + 3, U + 6);
  const double a0 = det3(W, W + 3, W + 6);
  const double a2 = det3(W, U + 3, U + 6) + det3(U, W + 3, U + 6) + det3(U, U + 3, W + 6);
  const double a1 = det3(U, W + 3, W + 6) + det3(W, U + 3, W + 6) + det3(W, W + 3, U + 6);
  double roots[3];
  int nr = 0;
  if (a3 != 0) {
    nr = solve_cubic_monic(a2 / a3, a1 / a3, a0 / a3, roots);
    if (nr == 3) {
      if (roots[0] != roots[1] && roots[1] != roots[2] && roots[0] != roots[2])
        cv_root_order3(roots);
      else
        nr = 1;
    }
  } else if (a2 != 0) {  // solveCubic's quadratic branch: the root of larger |q| first
    const double disc = a1 * a1 - 4.0 * a2 * a0;
    if (disc >= 0) {
      const double sq = sqrt(disc);
      const double q1 = (-a1 + sq) * 0.5, q2 = (a1 + sq) * -0.5;
      const double q = fabs(q1) > fabs(q2) ? q1 : q2;
      roots[nr++] = q / a2;
      if (disc > 0) roots[nr++] = a0 / q;
    }
  } else if (a1 != 0) {
    roots[nr++] = -a0 / a1;
  }
  int n = 0;
  for (int k = 0; k < nr; k++) {
    double lambda = roots[k], mu = 1.0;
    const double s = U[8] * lambda + W[8];
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

__device__ __forceinline__ float epi_error(const double *F, double x1, double y1, double x2, double y2) {
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
  const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
  return (float)((e1 < e2) ? e2 : e1);
}

__device__ bool have_collinear(const double *m, int count) {
  const int i = count - 1;
  for (int j = 0; j < i; j++) {
    const double dx1 = m[2 * j] - m[2 * i];
    const double dy1 = m[2 * j + 1] - m[2 * i + 1];
    for (int k = 0; k < j; k++) {
      const double dx2 = m[2 * k] - m[2 * i];
      const double dy2 = m[2 * k + 1] - m[2 * i + 1];
      if (fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
        return true;
    }
  }
  return false;
}

// LDS image of a RANSAC problem: correspondences as float4 (x1, y1, x2, y2) -- cv2 converts the
// points to CV_32F before estimating, so float storage is exact.
struct RansacShared {
  float4 *pts;  // [n] in dynamic LDS behind this struct -- or null: more correspondences than the buffer holds, read through:
  const uint32_t *gm;        //   the pair's packed match list (i | j << 16), or null: correspondence k = (gp1[k], gp2[k])
  const double *gp1, *gp2;   //   keypoints of the two images (x, y per feature)
  double models[kBatch][27];
  unsigned short subset[kBatch][8];
  unsigned char nmodels[kBatch];
  unsigned char subset_ok[kBatch];
  int good[kBatch][3];
  double best[9];
  int ctrl[8];  // 0: niters, 1: max_good, 2: done, 3: iters run, 4: found_any, 5: models scored so far
  unsigned long long rng_state;
};

// correspondence k as cv2 sees it (CV_32F): from the LDS buffer, or -- pairs with more matches than it holds, rare -- gathered from HBM
__device__ __forceinline__ float4 ransac_pt(const RansacShared &sh, int k) {
  if (sh.pts) return sh.pts[k];
  int i = k, j = k;
  if (sh.gm) {
    const uint32_t m = sh.gm[k];
    i = (int)(m & 0xFFFFu);
    j = (int)(m >> 16);
  }
  return make_float4((float)sh.gp1[2 * i], (float)sh.gp1[2 * i + 1], (float)sh.gp2[2 * j], (float)sh.gp2[2 * j + 1]);
}

// Runs the RANSAC loop over the n correspondences of sh (ransac_pt).
// On return sh.ctrl[1] = inlier count of the best model (0: none), sh.best = its F.
__device__ void ransac_core(RansacShared &sh, int n, double thr, double conf, int max_iters, int tid, double *priv, int *ipriv) {
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (thr <= 0) thr = 3;
  if (conf < 2.220446049250313e-16 || conf > 1 - 2.220446049250313e-16) conf = 0.99;
  const float t = (float)(thr * thr);
  if (tid == 0) {
    sh.ctrl[0] = max_iters > 1 ? max_iters : 1;
    sh.ctrl[1] = 0;
    sh.ctrl[2] = 0;
    sh.ctrl[3] = 0;
    sh.ctrl[5] = 0;
    sh.rng_state = ~0ull;
  }
  __syncthreads();
  // Speculation depth: the batch a round solves and scores ahead of the sequential decision.  cv2 stops after niters iterations and
  // niters collapses as soon as a good model is found (6 iterations at 97 % inliers), so the first rounds are short -- 8, 16, 32, then 64
  // hypotheses -- and only pairs that really need hundreds of iterations pay for full batches.  The subsets are drawn in stream
  // order whatever the batch boundaries are, so the result does not depend on this schedule.
  int bs = kFirstBatch;
  for (int it0 = 0;;) {
    // ---- 1. subsets (sequential RNG stream) ----
    if (tid == 0) {
      CvRng rng{sh.rng_state};
      const int niters = sh.ctrl[0];
      for (int b = 0; b < bs; ++b) sh.subset_ok[b] = 0;
      for (int b = 0; b < bs; ++b) {
        if (it0 + b >= niters) break;  // never consumed by the sequential loop
        int idx[7];
        double ms1[14], ms2[14];
        bool found = false;
        for (int attempt = 0; attempt < 10000 && !found; ++attempt) {
          for (int i = 0; i < 7; ++i) {
            int idx_i;
            for (;;) {
              idx_i = rng.uniform(0, n);
              bool dup = false;
              for (int j = 0; j < i; j++)
                if (idx[j] == idx_i) dup = true;
              if (!dup) break;
            }
            idx[i] = idx_i;
            const float4 q = ransac_pt(sh, idx_i);
            ms1[2 * i] = (double)q.x;
            ms1[2 * i + 1] = (double)q.y;
            ms2[2 * i] = (double)q.z;
            ms2[2 * i + 1] = (double)q.w;
          }
          found = !have_collinear(ms1, 7) && !have_collinear(ms2, 7);
        }
        if (found) {
          sh.subset_ok[b] = 1;
          for (int i = 0; i < 7; ++i) sh.subset[b][i] = (unsigned short)idx[i];
        } else {
          break;  // sequential loop stops here (iter == 0: no model at all)
        }
      }
      sh.rng_state = rng.state;
    }
    __syncthreads();
    // ---- 2. hypotheses ----
    if (tid < bs) {
      int nm = 0;
      if (sh.subset_ok[tid]) {
        double ms1[14], ms2[14];
        for (int i = 0; i < 7; ++i) {
          const float4 q = ransac_pt(sh, sh.subset[tid][i]);
          ms1[2 * i] = (double)q.x;
          ms1[2 * i + 1] = (double)q.y;
          ms2[2 * i] = (double)q.z;
          ms2[2 * i + 1] = (double)q.w;
        }
        nm = run_7point(ms1, ms2, sh.models[tid], priv + tid, ipriv + tid);
      }
      sh.nmodels[tid] = (unsigned char)nm;
    }
    __syncthreads();
    // ---- 3. scoring: one wavefront per model ----
    for (int mi = w; mi < bs * 3; mi += kWaves) {
      const int b = mi / 3, k = mi - 3 * b;
      if (k >= sh.nmodels[b]) continue;
      double F[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) F[i] = sh.models[b][9 * k + i];
      int good = 0;
      for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        if (i < n) {
          const float4 q = ransac_pt(sh, i);
          in = epi_error(F, (double)q.x, (double)q.y, (double)q.z, (double)q.w) <= t;
        }
        good += __popcll(__ballot(in));
      }
      if (lane == 0) sh.good[b][k] = good;
    }
    __syncthreads();
    // ---- 4. sequential replay of the batch ----
    if (tid == 0) {
      int niters = sh.ctrl[0], max_good = sh.ctrl[1];
      int iter = it0;
      bool done = false;
      {
        int scored = 0;
        for (int b = 0; b < bs; ++b) scored += sh.nmodels[b];
        sh.ctrl[5] += scored;
      }
      for (int b = 0; b < bs; ++b, ++iter) {
        if (iter >= niters) {
          done = true;
          break;
        }
        if (!sh.subset_ok[b]) {  // getSubset failed: `if (iter == 0) return false; break;`
          done = true;
          break;
        }
        const int nm = sh.nmodels[b];
        for (int k = 0; k < nm; ++k) {
          const int good = sh.good[b][k];
          const int lim = max_good > 6 ? max_good : 6;
          if (good > lim) {
            for (int i = 0; i < 9; ++i) sh.best[i] = sh.models[b][9 * k + i];
            max_good = good;
            niters = update_num_iters(conf, (double)(n - good) / n, niters);
          }
        }
      }
      if (!done && iter >= niters) done = true;
      sh.ctrl[0] = niters;
      sh.ctrl[1] = max_good;
      sh.ctrl[2] = done ? 1 : 0;
      sh.ctrl[3] = iter;
    }
    __syncthreads();
    if (sh.ctrl[2]) break;
    it0 += bs;
    bs = bs * 2 < kBatch ? bs * 2 : kBatch;
  }
}

struct RansacPairsArgs {
  const double *pts;  // store keypoints, padded tile rows
  const int64_t *tile_off;
  const int32_t *pairs;
  long n_pairs;
  int cap;
  int min_match;
  double thr, conf;
  int max_iters;
  int32_t *counts;
  uint32_t *matches;
  double *F_out;
  unsigned long long *work;  // optional: += (models scored) x (correspondences) of every pair, the work the roofline line counts
  int lds_pts;               // capacity of the LDS point buffer (correspondences)
};

__global__ void __launch_bounds__(kThreads) ransac_pairs_kernel(RansacPairsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  RansacShared &sh = *reinterpret_cast<RansacShared *>(smem);
  double *priv = reinterpret_cast<double *>(smem + sizeof(RansacShared));     // [kPrivDoubles][64]: the 7-point systems (run_7point)
  int *ipriv = reinterpret_cast<int *>(priv + kPrivDoubles * 64);             // [9][64]
  float4 *ptsbuf = reinterpret_cast<float4 *>(ipriv + 9 * 64);                // [lds_pts]
  int *misc = reinterpret_cast<int *>(ptsbuf + a.lds_pts);                    // [8]
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const long p = blockIdx.x;
  const int n = min(a.counts[p], a.cap);
  // gates: matching.py:590-598 (min match) and matching.py:787-788 (< 8)
  if (n < a.min_match || n < 8) {
    if (tid == 0) a.counts[p] = 0;
    return;
  }
  const int img1 = a.pairs[2 * p], img2 = a.pairs[2 * p + 1];
  const double *pts1 = a.pts + a.tile_off[img1] * 64;
  const double *pts2 = a.pts + a.tile_off[img2] * 64;
  const bool in_lds = n <= a.lds_pts;  // uniform per workgroup
  if (tid == 0) {
    sh.pts = in_lds ? ptsbuf : nullptr;
    sh.gm = a.matches + p * a.cap;
    sh.gp1 = pts1;
    sh.gp2 = pts2;
  }
  if (in_lds)
    for (int k = tid; k < n; k += kThreads) {
      const uint32_t m = a.matches[p * a.cap + k];
      const int i = m & 0xFFFF, j = m >> 16;
      ptsbuf[k] = make_float4((float)pts1[2 * i], (float)pts1[2 * i + 1], (float)pts2[2 * j], (float)pts2[2 * j + 1]);
    }
  __syncthreads();
  ransac_core(sh, n, a.thr, a.conf, a.max_iters, tid, priv, ipriv);
  const int max_good = sh.ctrl[1];
  if (a.work && tid == 0) atomicAdd(a.work, (unsigned long long)sh.ctrl[5] * (unsigned long long)n);
  if (a.F_out && tid < 9) a.F_out[p * 9 + tid] = max_good > 0 ? sh.best[tid] : 0.0;
  // matching.py:798-800: F is None or F[2,2] == 0 -> no matches
  if (max_good <= 0 || sh.best[8] == 0.0) {
    if (tid == 0) a.counts[p] = 0;
    return;
  }
  double F[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) F[i] = sh.best[i];
  const double thr = a.thr <= 0 ? 3 : a.thr;
  const float t = (float)(thr * thr);
  // final mask + ordered compaction, in place
  int base = 0;
  for (int k0 = 0; k0 < n; k0 += kThreads) {
    const int k = k0 + tid;
    bool in = false;
    uint32_t mk = 0;  // read before the barrier below: the in-place writes of this round only go to slots <= k
    if (k < n) {
      const float4 q = ransac_pt(sh, k);
      in = epi_error(F, (double)q.x, (double)q.y, (double)q.z, (double)q.w) <= t;
      mk = a.matches[p * a.cap + k];
    }
    const unsigned long long bal = __ballot(in);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) misc[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < kWaves; ++w2) {
      const int cnt = misc[w2];
      woff += (w2 < w) ? cnt : 0;
      total += cnt;
    }
    if (in) a.matches[p * a.cap + base + woff + prefix] = mk;
    base += total;
    __syncthreads();
  }
  // matching.py:632-634
  if (tid == 0) a.counts[p] = base >= a.min_match ? base : 0;
}

__global__ void __launch_bounds__(kThreads) ransac_single_kernel(const double *p1, const double *p2, int n, double thr,
                                                                   double conf, int max_iters, double *F_out,
                                                                   uint8_t *mask, int32_t *info, int in_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  RansacShared &sh = *reinterpret_cast<RansacShared *>(smem);
  const int tid = threadIdx.x;
  double *priv = reinterpret_cast<double *>(smem + sizeof(RansacShared));
  int *ipriv = reinterpret_cast<int *>(priv + kPrivDoubles * 64);
  float4 *ptsbuf = reinterpret_cast<float4 *>(ipriv + 9 * 64);
  if (tid == 0) {
    sh.pts = in_lds ? ptsbuf : nullptr;
    sh.gm = nullptr;
    sh.gp1 = p1;
    sh.gp2 = p2;
  }
  if (in_lds)
    for (int k = tid; k < n; k += kThreads)
      ptsbuf[k] = make_float4((float)p1[2 * k], (float)p1[2 * k + 1], (float)p2[2 * k], (float)p2[2 * k + 1]);
  __syncthreads();
  ransac_core(sh, n, thr, conf, max_iters, tid, priv, ipriv);
  const int max_good = sh.ctrl[1];
  if (tid == 0) {
    info[0] = max_good > 0 ? 1 : 0;
    info[1] = sh.ctrl[3];
    info[2] = max_good;
  }
  if (tid < 9) F_out[tid] = max_good > 0 ? sh.best[tid] : 0.0;
  double F[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) F[i] = sh.best[i];
  const double thr2 = thr <= 0 ? 3 : thr;
  const float t = (float)(thr2 * thr2);
  for (int k = tid; k < n; k += kThreads) {
    const float4 q = ransac_pt(sh, k);
    mask[k] = (max_good > 0 && epi_error(F, (double)q.x, (double)q.y, (double)q.z, (double)q.w) <= t) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// LMedS branch: cv2.findFundamentalMat(FM_RANSAC) with 8 <= n < 15 points runs the LMedS registrator
// (fundam.cpp: RANSAC only for npoints >= 15).  Published algorithm (ptsetreg.cpp): a FIXED number of
// iterations niters = max(RANSACUpdateNumIters(conf, 0.45, 7, maxIters), 3); per hypothesis the median
// = element n/2 of the sorted float errors; smallest median wins (strict <, first in stream order);
// sigma = 2.5*1.4826*(1 + 5/(n-7))*sqrt(minMedian) >= 0.001; inliers err <= sigma^2; success iff >= 7.
// Same batch-parallel scheme as ransac_core: lane 0 draws 64 subsets from the RNG stream, 64 lanes
// solve and score them (n <= 14 points: a 14-element insertion sort per hypothesis), lane 0 replays.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) lmeds_single_kernel(const double *p1, const double *p2, int n, double conf, int max_iters,
                                                           double *F_out, uint8_t *mask, int32_t *info) {
  __shared__ float4 pts[16];
  __shared__ unsigned short subset[kBatch][8];
  __shared__ unsigned char subset_ok[kBatch];
  __shared__ double models[kBatch][27];
  __shared__ float med[kBatch][3];
  __shared__ unsigned char nmodels[kBatch];
  __shared__ double priv[kPrivDoubles * 64];
  __shared__ int ipriv[9 * 64];
  __shared__ double best[9];
  __shared__ int ctrl[4];
  __shared__ unsigned long long rng_state;
  const int tid = threadIdx.x;
  if (tid < n) pts[tid] = make_float4((float)p1[2 * tid], (float)p1[2 * tid + 1], (float)p2[2 * tid], (float)p2[2 * tid + 1]);
  if (conf < 2.220446049250313e-16 || conf > 1 - 2.220446049250313e-16) conf = 0.99;
  int niters = update_num_iters(conf, 0.45, max_iters);
  if (niters < 3) niters = 3;
  double min_median = 1.7976931348623157e308;
  if (tid == 0) {
    rng_state = ~0ull;
    ctrl[0] = 0;  // done
    ctrl[1] = 0;  // iterations run
    ctrl[2] = 0;  // a model exists
  }
  __syncthreads();
  for (int it0 = 0; it0 < niters; it0 += kBatch) {
    if (tid == 0) {
      CvRng rng{rng_state};
      for (int b = 0; b < kBatch; ++b) subset_ok[b] = 0;
      for (int b = 0; b < kBatch && it0 + b < niters; ++b) {
        int idx[7];
        double ms1[14], ms2[14];
        bool found = false;
        for (int attempt = 0; attempt < 10000 && !found; ++attempt) {
          for (int i = 0; i < 7; ++i) {
            int idx_i;
            for (;;) {
              idx_i = rng.uniform(0, n);
              bool dup = false;
              for (int j = 0; j < i; j++)
                if (idx[j] == idx_i) dup = true;
              if (!dup) break;
            }
            idx[i] = idx_i;
            const float4 q = pts[idx_i];
            ms1[2 * i] = (double)q.x;
            ms1[2 * i + 1] = (double)q.y;
            ms2[2 * i] = (double)q.z;
            ms2[2 * i + 1] = (double)q.w;
          }
          found = !have_collinear(ms1, 7) && !have_collinear(ms2, 7);
        }
        if (!found) break;
        subset_ok[b] = 1;
        for (int i = 0; i < 7; ++i) subset[b][i] = (unsigned short)idx[i];
      }
      rng_state = rng.state;
    }
    __syncthreads();
    {
      int nm = 0;
      if (subset_ok[tid]) {
        double ms1[14], ms2[14];
        for (int i = 0; i < 7; ++i) {
          const float4 q = pts[subset[tid][i]];
          ms1[2 * i] = (double)q.x;
          ms1[2 * i + 1] = (double)q.y;
          ms2[2 * i] = (double)q.z;
          ms2[2 * i + 1] = (double)q.w;
        }
        nm = run_7point(ms1, ms2, models[tid], priv + tid, ipriv + tid);
        for (int k = 0; k < nm; ++k) {
          float e[16];
          for (int i = 0; i < n; ++i) {
            const float4 q = pts[i];
            e[i] = epi_error(models[tid] + 9 * k, (double)q.x, (double)q.y, (double)q.z, (double)q.w);
          }
          for (int i = 1; i < n; ++i) {
            const float v = e[i];
            int j = i - 1;
            for (; j >= 0 && e[j] > v; --j) e[j + 1] = e[j];
            e[j + 1] = v;
          }
          med[tid][k] = e[n / 2];
        }
      }
      nmodels[tid] = (unsigned char)nm;
    }
    __syncthreads();
    if (tid == 0) {
      int iter = it0;
      bool done = false;
      for (int b = 0; b < kBatch; ++b, ++iter) {
        if (iter >= niters) { done = true; break; }
        if (!subset_ok[b]) { done = true; break; }  // getSubset failed: `if (iter == 0) return false; break;`
        for (int k = 0; k < nmodels[b]; ++k) {
          const double median = (double)med[b][k];
          if (median < min_median) {
            min_median = median;
            for (int i = 0; i < 9; ++i) best[i] = models[b][9 * k + i];
            ctrl[2] = 1;
          }
        }
      }
      ctrl[0] = done ? 1 : 0;
      ctrl[1] = iter;
    }
    __syncthreads();
    if (ctrl[0]) break;
  }
  // lane 0 owns min_median; the others only need the verdict
  __shared__ float tsq;
  if (tid == 0) {
    double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * sqrt(min_median);
    if (sigma < 0.001) sigma = 0.001;
    tsq = (float)(sigma * sigma);
  }
  __syncthreads();
  const bool have = ctrl[2] != 0;
  bool in = false;
  if (have && tid < n) {
    const float4 q = pts[tid];
    in = epi_error(best, (double)q.x, (double)q.y, (double)q.z, (double)q.w) <= tsq;
  }
  const int count = __popcll(__ballot(in));
  const bool ok = have && count >= 7;
  if (tid < n) mask[tid] = (ok && in) ? 1 : 0;
  if (tid < 9) F_out[tid] = ok ? best[tid] : 0.0;
  if (tid == 0) {
    info[0] = ok ? 1 : 0;
    info[1] = ctrl[1];
    info[2] = ok ? count : 0;
  }
}

}  // namespace

static constexpr size_t kRansacFixedLds = sizeof(RansacShared) + (size_t)kPrivDoubles * 64 * sizeof(double) + 9 * 64 * sizeof(int);
// correspondences the single-problem kernel stages in LDS (it owns a CU); the batched kernel keeps at most kPairsLdsPts (two workgroups
// per CU) and takes pairs with more matches from HBM
int osfm_ransac_lds_points() { return (int)((160 * 1024 - kRansacFixedLds - 64) / 16) & ~3; }
static constexpr int kPairsLdsPts = 1024;
int osfm_ransac_pairs_lds_points() { return kPairsLdsPts; }

static int ensure_ransac_attributes(int device) {
  static OsfmPerDeviceOnce once;
  return once.run(device, []() -> int {
    OSFM_HIP(hipFuncSetAttribute((const void *)ransac_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OSFM_HIP(hipFuncSetAttribute((const void *)ransac_single_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return OSFM_OK;
  });
}

int osfm_launch_ransac_pairs(osfm_ctx *ctx, const osfm_store *store, const int32_t *d_pairs, int64_t n_pairs, int cap,
                             int min_match, double thr, double conf, int max_iters, int32_t *d_counts,
                             uint32_t *d_matches, double *d_F_or_null, hipStream_t stream, unsigned long long *d_work_or_null) {
  if (n_pairs == 0) return OSFM_OK;
  RansacPairsArgs a;
  a.pts = store->d_pts;
  a.tile_off = store->d_tile_off;
  a.pairs = d_pairs;
  a.n_pairs = n_pairs;
  a.cap = cap;
  a.min_match = min_match;
  a.thr = thr;
  a.conf = conf;
  a.max_iters = max_iters;
  a.counts = d_counts;
  a.matches = d_matches;
  a.F_out = d_F_or_null;
  a.work = d_work_or_null;
  const int capr = (cap + 3) & ~3;
  a.lds_pts = std::min(capr, kPairsLdsPts);
  const size_t lds = kRansacFixedLds + (size_t)a.lds_pts * 16 + 64;
  {
    const int rc = ensure_ransac_attributes(ctx->device);
    if (rc != OSFM_OK) return rc;
  }
  hipLaunchKernelGGL(ransac_pairs_kernel, dim3((unsigned)n_pairs), dim3(kThreads), lds, stream, a);
  OSFM_HIP(hipGetLastError());
  return OSFM_OK;
}

int osfm_launch_ransac_single(osfm_ctx *ctx, const double *d_p1, const double *d_p2, int n, double thr, double conf,
                              int max_iters, double *d_F, uint8_t *d_mask, int32_t *d_info) {
  const bool in_lds = ((n + 3) & ~3) <= osfm_ransac_lds_points();
  {
    const int rc = ensure_ransac_attributes(ctx->device);
    if (rc != OSFM_OK) return rc;
  }
  if (n < 15)  // cv2 switches to LMedS below 15 correspondences
    hipLaunchKernelGGL(lmeds_single_kernel, dim3(1), dim3(64), 0, ctx->stream, d_p1, d_p2, n, conf, max_iters, d_F, d_mask, d_info);
  else
    hipLaunchKernelGGL(ransac_single_kernel, dim3(1), dim3(kThreads), kRansacFixedLds + (in_lds ? (size_t)((n + 3) & ~3) * 16 : 0) + 64, ctx->stream,
                       d_p1, d_p2, n, thr, conf, max_iters, d_F, d_mask, d_info, in_lds ? 1 : 0);
  OSFM_HIP(hipGetLastError());
  return OSFM_OK;
}

extern "C" int osfm_ransac_fundamental(osfm_ctx *ctx, const double *p1, const double *p2, int n, double thr, double conf,
                                       int max_iters, double F[9], uint8_t *mask, int *found, int *iters_run) {
  OSFM_REQUIRE(ctx && p1 && p2 && F && mask && found, OSFM_E_INVALID, "osfm_ransac_fundamental: null argument");
  *found = 0;
  if (iters_run) *iters_run = 0;
  for (int i = 0; i < n; ++i) mask[i] = 0;
  if (n < 7) return OSFM_OK;  // cv2: npoints < 7 -> empty Mat
  OSFM_REQUIRE(n >= 8, OSFM_E_UNSUPPORTED, "n = 7: cv2 returns the stacked 7-point solutions; the reference requires >= 8 matches (matching.py:787)");
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  double *d_p1 = nullptr, *d_p2 = nullptr, *d_F = nullptr;
  uint8_t *d_mask = nullptr;
  int32_t *d_info = nullptr;
  OSFM_HIP(hipMalloc((void **)&d_p1, (size_t)n * 16));
  OSFM_HIP(hipMalloc((void **)&d_p2, (size_t)n * 16));
  OSFM_HIP(hipMalloc((void **)&d_F, 9 * 8));
  OSFM_HIP(hipMalloc((void **)&d_mask, (size_t)n));
  OSFM_HIP(hipMalloc((void **)&d_info, 16));
  int rc = OSFM_OK;
  hipError_t e;
  do {
    if ((e = hipMemcpyAsync(d_p1, p1, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(d_p2, p2, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
    rc = osfm_launch_ransac_single(ctx, d_p1, d_p2, n, thr, conf, max_iters, d_F, d_mask, d_info);
    if (rc != OSFM_OK) break;
    int32_t info[4] = {0, 0, 0, 0};
    if ((e = hipMemcpyAsync(F, d_F, 72, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(mask, d_mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(info, d_info, 12, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) break;
    *found = info[0];
    if (iters_run) *iters_run = info[1];
  } while (0);
  (void)hipFree(d_p1);
  (void)hipFree(d_p2);
  (void)hipFree(d_F);
  (void)hipFree(d_mask);
  (void)hipFree(d_info);
  if (rc == OSFM_OK && e != hipSuccess) {
    osfm_set_error("osfm_ransac_fundamental: %s", hipGetErrorString(e));
    rc = OSFM_E_HIP;
  }
  return rc;
}
