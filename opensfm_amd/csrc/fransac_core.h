// fransac_core.h -- numerics and per-pair logic of the fundamental-matrix RANSAC (cv2.findFundamentalMat(FM_RANSAC) as
// robust_match_fundamental calls it, opensfm/matching.py:780-802), shared by the kernels of ransac.hip and by the host emulation
// tests/native/fransac_host.cpp, which runs the same functions with loops in place of lanes and compares every result with the CPU
// oracle bit for bit (tests/test_fransac_host.py).  Compiled with -ffp-contract=off on both sides: + - * / sqrt in a fixed order.
//
// Round 3 organisation (ransac.hip): one WAVEFRONT per pair for the first 8 hypotheses (fransac_first_kernel) -- at the inlier ratios
// pair preselection produces cv2's adaptive iteration count collapses to ~6 after the first all-inlier sample, so four pairs in five
// end there --, and one workgroup per remaining pair for the long runs (fransac_rest_kernel, a persistent grid that pulls pairs from
// the list the first kernel leaves).  The draws are no longer a single lane's loop:
//   cv::RNG is sequential, but the chain of subset STARTS in its output stream depends only on the raw values and on n (a subset
//   consumes 7 values plus one per duplicate; a subset rejected by the collinearity test consumes its values all the same), not on the
//   points.  So lane 0 only produces the raw stream; every lane reduces values mod n and, for every stream position, finds how many
//   values a subset starting there consumes; lane 0 hops along that table; then one lane per subset gathers its 7 correspondences and
//   runs the collinearity test, and the accepted subsets are compacted in stream order.  The result is the sequence of
//   getSubset() calls of ptsetreg.cpp, whatever the batch boundaries are.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define FR_FN __host__ __device__ inline
#define FR_CONST __constant__
#else
#define FR_FN inline
#define FR_CONST static const
#endif

namespace fransac {

struct CvRng {
  unsigned long long state;
  FR_FN unsigned next() {
    state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  FR_FN int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + (unsigned)a); }
};

FR_FN double det_log(double x) {
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

FR_FN int update_num_iters(double p, double ep, int max_iters) {
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

FR_FN double det3(const double *a, const double *b, const double *c) {
  return a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
}

FR_FN int solve_cubic_monic(double a, double b, double c, double *roots) {
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
FR_CONST CvSvdFill kCvSvdFill{};

FR_FN double dot9(const double *a, const double *b) {
  double s = 0.0;
  for (int i = 0; i < 9; i++) s = s + a[i] * b[i];
  return s;
}

FR_FN int cv_null_basis(const double *v1, const double *v2, double *f1, double *f2) {
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
FR_FN void cv_root_order3(double *r) {
  double lo = r[0], hi = r[0], mid = r[0];
#pragma unroll
  for (int k = 1; k < 3; k++) {
    if (r[k] < lo) lo = r[k];
    if (r[k] > hi) hi = r[k];
  }
#pragma unroll
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
#define OSFM_A(r, c) priv[((r) * 9 + (c)) * STRIDE]
template <int STRIDE>
FR_FN int run_7point(const double *m1, const double *m2, double *F, double *priv, int *ipriv) {
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
  for (int c = 0; c < 9; c++) ipriv[c * STRIDE] = c;
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
      const int t = ipriv[k * STRIDE];
      ipriv[k * STRIDE] = ipriv[pc * STRIDE];
      ipriv[pc * STRIDE] = t;
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
  double *v1p = priv + 63 * STRIDE, *v2p = priv + 72 * STRIDE;
  for (int k = 0; k < 7; k++) {
    const int cp = ipriv[k * STRIDE];
    v1p[cp * STRIDE] = -OSFM_A(k, 7);
    v2p[cp * STRIDE] = -OSFM_A(k, 8);
  }
  {
    const int c7 = ipriv[7 * STRIDE], c8 = ipriv[8 * STRIDE];
    v1p[c7 * STRIDE] = 1.0;
    v1p[c8 * STRIDE] = 0.0;
    v2p[c7 * STRIDE] = 0.0;
    v2p[c8 * STRIDE] = 1.0;
  }
  double v1[9], v2[9];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    v1[i] = v1p[i * STRIDE];
    v2[i] = v2p[i * STRIDE];
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
#pragma unroll
  for (int k = 0; k < 3; k++) {  // unrolled: roots[] is indexed statically and stays in registers
    if (k >= nr) break;
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

FR_FN float epi_error(const double *F, double x1, double y1, double x2, double y2) {
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

// haveCollinearPoints for the LAST of 7 points (fundam.cpp); fully unrolled so that m[] is indexed statically and stays in registers
FR_FN bool have_collinear7(const double *m) {
  bool col = false;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const double dx1 = m[2 * j] - m[12];
    const double dy1 = m[2 * j + 1] - m[13];
#pragma unroll
    for (int k = 0; k < j; k++) {
      const double dx2 = m[2 * k] - m[12];
      const double dy2 = m[2 * k + 1] - m[13];
      col = col || (fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)));
    }
  }
  return col;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// the draws
// ---------------------------------------------------------------------------------------------------------------------------------
struct Pt4 {
  float x, y, z, w;  // (x1, y1, x2, y2) as cv2 sees a correspondence (CV_32F)
};

constexpr int kMaxAttempts = 10000;  // getSubset's maxAttempts (ptsetreg.cpp)

// LDS scratch of one round of draws: RAW values of the stream, at most LMAX subsets
template <int RAW, int LMAX>
struct DrawBuf {
  unsigned long long states[RAW];  // cv::RNG state after each step; the value drawn is its low word
  unsigned short u[RAW];           // value mod n: cv::RNG::uniform(0, n)
  unsigned char clen[RAW];         // values a subset STARTING at this position consumes (7 + duplicates); 0: runs off the buffer
  unsigned short chain[LMAX];      // start positions of the successive getSubset attempts
  unsigned char acc[LMAX];         // attempt accepted (no collinear triple in either image)
  unsigned char slot[LMAX];        // accepted attempt -> iteration slot of this round (0xFF: none)
  unsigned short tmp[LMAX][8];     // the 7 indices of every attempt
  int nraw, nchain;
};
// what a round of draws hands on (kept apart from DrawBuf, whose storage the 7-point systems reuse afterwards)
struct DrawOut {
  int nsub;                      // iterations (accepted subsets) this round produced
  int fail;                      // 1: the iteration after those exhausted getSubset's 10000 attempts (it returned false)
  int rej_run;                   // in / out: rejected attempts of the pending getSubset call so far
  int done;                      // set by the replay: the RANSAC loop has ended
  unsigned long long state_out;  // RNG state after the last value an attempt of this round consumed
};

// phase 1 (one lane): the raw stream
template <int RAW, int LMAX>
FR_FN void draw_gen(DrawBuf<RAW, LMAX> &B, unsigned long long state, int nraw) {
  B.nraw = nraw;
  for (int s = 0; s < nraw; ++s) {
    state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
    B.states[s] = state;
  }
}
// phase 2 (any lane, position s): uniform(0, n) = next % n
template <int RAW, int LMAX>
FR_FN void draw_mod(DrawBuf<RAW, LMAX> &B, int s, int n) {
  B.u[s] = (unsigned short)((unsigned)B.states[s] % (unsigned)n);
}
// the subset an attempt starting at position s draws: successive values, a value equal to an earlier index of the subset is drawn
// again (getSubset's inner `for (;;)`).  Returns the number of values consumed, 0 when the buffer ends first.  idx[] statically indexed.
template <int RAW, int LMAX>
FR_FN int draw_walk(const DrawBuf<RAW, LMAX> &B, int s, unsigned short *idx) {
#pragma unroll
  for (int j = 0; j < 7; ++j) idx[j] = 0xFFFF;  // n <= 16000: never a valid index
  int cnt = 0, t = s;
  while (cnt < 7 && t < B.nraw) {
    const unsigned short v = B.u[t++];
    bool dup = false;
#pragma unroll
    for (int j = 0; j < 7; ++j) dup = dup || idx[j] == v;
    if (!dup) {
#pragma unroll
      for (int j = 0; j < 7; ++j)
        if (j == cnt) idx[j] = v;
      ++cnt;
    }
  }
  const int used = t - s;
  return (cnt == 7 && used <= 255) ? used : 0;
}
// phase 3 (any lane, position s)
template <int RAW, int LMAX>
FR_FN void draw_len(DrawBuf<RAW, LMAX> &B, int s) {
  unsigned short idx[7];
  B.clen[s] = (unsigned char)draw_walk(B, s, idx);
}
// phase 4 (one lane): hop along the starts
template <int RAW, int LMAX>
FR_FN void draw_chain(DrawBuf<RAW, LMAX> &B, int lmax) {
  int pos = 0, e = 0;
  while (e < lmax && pos < B.nraw && B.clen[pos] != 0) {
    B.chain[e++] = (unsigned short)pos;
    pos += B.clen[pos];
  }
  B.nchain = e;
}
// phase 5 (lane e < nchain): the attempt's correspondences and the collinearity test; PTS: k -> Pt4
template <int RAW, int LMAX, class PTS>
FR_FN void draw_check(DrawBuf<RAW, LMAX> &B, int e, const PTS &pts) {
  unsigned short idx[7];
  draw_walk(B, B.chain[e], idx);
  double ms1[14], ms2[14];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const Pt4 q = pts(idx[i]);
    ms1[2 * i] = (double)q.x;
    ms1[2 * i + 1] = (double)q.y;
    ms2[2 * i] = (double)q.z;
    ms2[2 * i + 1] = (double)q.w;
    B.tmp[e][i] = idx[i];
  }
  B.acc[e] = (!have_collinear7(ms1) && !have_collinear7(ms2)) ? 1 : 0;
}
// phase 6 (one lane): iteration slots in stream order, the 10000-attempt rule, the state the next round continues from
template <int RAW, int LMAX>
FR_FN void draw_slots(DrawBuf<RAW, LMAX> &B, DrawOut &O, unsigned long long state_in) {
  int run = O.rej_run, nsub = 0, fail = 0, last = -1;
  for (int e = 0; e < B.nchain && !fail; ++e) {
    last = e;
    if (B.acc[e]) {
      B.slot[e] = (unsigned char)nsub++;
      run = 0;
    } else {
      B.slot[e] = 0xFF;
      if (++run >= kMaxAttempts) fail = 1;
    }
  }
  for (int e = last + 1; e < B.nchain; ++e) B.slot[e] = 0xFF;
  O.nsub = nsub;
  O.fail = fail;
  O.rej_run = run;
  O.state_out = last >= 0 ? B.states[B.chain[last] + B.clen[B.chain[last]] - 1] : state_in;
}
// phase 7 (lane e < nchain): accepted attempts into their iteration slots; subset: [LMAX][8]
template <int RAW, int LMAX>
FR_FN void draw_emit(const DrawBuf<RAW, LMAX> &B, int e, unsigned short (*subset)[8]) {
  const int sl = B.slot[e];
  if (sl != 0xFF) {
#pragma unroll
    for (int i = 0; i < 7; ++i) subset[sl][i] = B.tmp[e][i];
  }
}
// The stream position the table could not resolve (a subset that needs more values than the buffer holds: n close to 7 and a long
// run of duplicates): ONE getSubset attempt drawn the plain sequential way by one lane.  Practically never taken.
template <class PTS>
FR_FN void draw_one_sequential(DrawOut &B, unsigned long long state, int n, const PTS &pts, unsigned short (*subset)[8]) {
  CvRng rng{state};
  unsigned short idx[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) idx[j] = 0xFFFF;
  int cnt = 0;
  while (cnt < 7) {
    const unsigned short v = (unsigned short)rng.uniform(0, n);
    bool dup = false;
#pragma unroll
    for (int j = 0; j < 7; ++j) dup = dup || idx[j] == v;
    if (!dup) {
#pragma unroll
      for (int j = 0; j < 7; ++j)
        if (j == cnt) idx[j] = v;
      ++cnt;
    }
  }
  double ms1[14], ms2[14];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const Pt4 q = pts(idx[i]);
    ms1[2 * i] = (double)q.x;
    ms1[2 * i + 1] = (double)q.y;
    ms2[2 * i] = (double)q.z;
    ms2[2 * i + 1] = (double)q.w;
  }
  const bool ok = !have_collinear7(ms1) && !have_collinear7(ms2);
  B.nsub = 0;
  B.fail = 0;
  if (ok) {
#pragma unroll
    for (int i = 0; i < 7; ++i) subset[0][i] = idx[i];
    B.nsub = 1;
    B.rej_run = 0;
  } else if (++B.rej_run >= kMaxAttempts) {
    B.fail = 1;
  }
  B.state_out = rng.state;
}
// values worth generating for a round of lmax subsets: 7 each plus the expected duplicates (sum over i < 7 of n / (n - i) draws per
// subset) plus slack; a round that runs short simply produces fewer iterations
FR_FN int draw_raw_count(int lmax, int n, int raw_cap) {
  double per = 0.0;
  for (int i = 0; i < 7; ++i) per += (double)n / (double)(n - i);
  int want = (int)(per * lmax * 1.15) + 24;
  return want < raw_cap ? want : raw_cap;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// per-pair state between the first kernel and the long-run kernel, and the sequential decision rules
// ---------------------------------------------------------------------------------------------------------------------------------
struct PairState {
  unsigned long long rng;     // cv::RNG state (seed (uint64)-1)
  unsigned long long scored;  // models scored so far (work counter of the roofline line)
  double best[9];
  int niters, max_good, iters, rej_run;
};
FR_FN void state_init(PairState &s, int max_iters) {
  s.rng = ~0ull;
  s.scored = 0;
  for (int i = 0; i < 9; ++i) s.best[i] = 0.0;
  s.niters = max_iters > 1 ? max_iters : 1;
  s.max_good = 0;
  s.iters = 0;
  s.rej_run = 0;
}
// RANSACPointSetRegistrator::run, the part after runKernel / findInliers, for the nsub iterations a round prepared:
// models: [slot][27], nmodels[slot], good: [slot][3] inlier counts.  `fail`: getSubset returned false for the iteration after them.
// Returns true when the loop has ended (iteration count reached, or getSubset failed: `if (iter == 0) return false; break;`).
FR_FN bool replay_round(PairState &s, int n, double conf, int nsub, int fail, const double (*models)[27], const unsigned char *nmodels,
                        const int (*good)[3]) {
  int iter = s.iters;
  bool done = false;
  for (int b = 0; b < nsub; ++b, ++iter) {
    if (iter >= s.niters) {
      done = true;
      break;
    }
    const int nm = nmodels[b];
    s.scored += (unsigned long long)nm;
    for (int k = 0; k < nm; ++k) {
      const int g = good[b][k];
      const int lim = s.max_good > 6 ? s.max_good : 6;
      if (g > lim) {
        for (int i = 0; i < 9; ++i) s.best[i] = models[b][9 * k + i];
        s.max_good = g;
        s.niters = update_num_iters(conf, (double)(n - g) / n, s.niters);
      }
    }
  }
  if (!done && iter >= s.niters) done = true;
  if (!done && fail) done = true;  // iter < niters and getSubset found nothing
  s.iters = iter;
  return done;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// One round for one pair in three stages: draw up to lmax subsets; solve them; score the models and apply the sequential decisions.
// EX supplies the parallelism: one(f) runs f on one lane, par(n, f) runs f(i) for i < n across the lanes, both followed by a barrier;
// score(...) counts the inliers of every model.  On the GPU EX is a wavefront or a 256-thread workgroup; the host emulation uses
// loops.  st, D, O, subset, models, nmodels, good, priv, ipriv live in memory all lanes share (LDS).
// D may alias priv: the draws are finished (and handed on through O and subset) before the first 7-point system is written.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int RAW, int LMAX, class EX, class PTS>
FR_FN void round_draw(EX &ex, PairState &st, DrawBuf<RAW, LMAX> &D, DrawOut &O, unsigned short (*subset)[8], const PTS &pts, int n, int lmax,
                      int raw_cap = RAW) {
  ex.one([&]() {
    O.rej_run = st.rej_run;
    draw_gen(D, st.rng, draw_raw_count(lmax, n, raw_cap < RAW ? raw_cap : RAW));
  });
  ex.par(D.nraw, [&](int s) { draw_mod(D, s, n); });
  ex.par(D.nraw, [&](int s) { draw_len(D, s); });
  ex.one([&]() { draw_chain(D, lmax); });
  if (D.nchain == 0) {
    ex.one([&]() { draw_one_sequential(O, st.rng, n, pts, subset); });
  } else {
    ex.par(D.nchain, [&](int e) { draw_check(D, e, pts); });
    ex.one([&]() { draw_slots(D, O, st.rng); });
    ex.par(D.nchain, [&](int e) { draw_emit(D, e, subset); });
  }
  ex.one([&]() {
    st.rng = O.state_out;
    st.rej_run = O.rej_run;
  });
}
// the 7 correspondences of a subset as the 7-point solver takes them
template <class PTS>
FR_FN void subset_points(const unsigned short *idx, const PTS &pts, double *ms1, double *ms2) {
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const Pt4 q = pts(idx[i]);
    ms1[2 * i] = (double)q.x;
    ms1[2 * i + 1] = (double)q.y;
    ms2[2 * i] = (double)q.z;
    ms2[2 * i + 1] = (double)q.w;
  }
}
template <int STRIDE, class EX, class PTS>
FR_FN void round_solve(EX &ex, int nsub, const unsigned short (*subset)[8], const PTS &pts, double (*models)[27], unsigned char *nmodels,
                       double *priv, int *ipriv) {
  ex.par(nsub, [&](int b) {
    double ms1[14], ms2[14];
    subset_points(subset[b], pts, ms1, ms2);
    nmodels[b] = (unsigned char)run_7point<STRIDE>(ms1, ms2, models[b], priv + b, ipriv + b);
  });
}
template <int RAW, int LMAX, int STRIDE, class EX, class PTS>
FR_FN bool fransac_round(EX &ex, PairState &st, DrawBuf<RAW, LMAX> &D, DrawOut &O, unsigned short (*subset)[8], double (*models)[27],
                         unsigned char *nmodels, int (*good)[3], double *priv, int *ipriv, const PTS &pts, int n, float t, double conf,
                         int lmax, int raw_cap = RAW) {
  round_draw(ex, st, D, O, subset, pts, n, lmax, raw_cap);
  round_solve<STRIDE>(ex, O.nsub, subset, pts, models, nmodels, priv, ipriv);
  ex.score(O.nsub, models, nmodels, pts, n, t, good);
  ex.one([&]() { O.done = replay_round(st, n, conf, O.nsub, O.fail, models, nmodels, good) ? 1 : 0; });
  return O.done != 0;
}

// The decisions with LAZY scoring, for a single wavefront whose lanes all carry the same state: a model is only scored when the
// sequential loop would reach it -- niters collapses after the first good model (to ~6 at 97 % inliers), and the hypotheses drawn and
// solved beyond it are never looked at.  CNT: model (9 doubles) -> number of inliers, the same value on every lane.
// models: [nsub][27], nmodels: [nsub].  Same return value as replay_round.
template <class CNT>
FR_FN bool decide_lazy(PairState &s, int n, double conf, int nsub, int fail, const double *models, const unsigned char *nmodels, const CNT &count) {
  int iter = s.iters;
  bool done = false;
  for (int b = 0; b < nsub; ++b, ++iter) {
    if (iter >= s.niters) {
      done = true;
      break;
    }
    const int nm = nmodels[b];
    s.scored += (unsigned long long)nm;
    for (int k = 0; k < nm; ++k) {
      const double *Fm = models + 27 * b + 9 * k;
      const int g = count(Fm);
      const int lim = s.max_good > 6 ? s.max_good : 6;
      if (g > lim) {
#pragma unroll
        for (int i = 0; i < 9; ++i) s.best[i] = Fm[i];
        s.max_good = g;
        s.niters = update_num_iters(conf, (double)(n - g) / n, s.niters);
      }
    }
  }
  if (!done && iter >= s.niters) done = true;
  if (!done && fail) done = true;
  s.iters = iter;
  return done;
}

}  // namespace fransac
