/*
 * match_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never on the product path).
 *
 * CPU restatement of the descriptor-matching half of OpenSfM's pair-matching hot path:
 *
 *   opensfm/matching.py:723-756  match_brute_force           (cv2 BruteForce L2 knnMatch k=2 + Lowe ratio)
 *   opensfm/matching.py:759-777  match_brute_force_symmetric (both directions + set intersection)
 *   opensfm/matching.py:683-720  match_flann[_symmetric]     (ratio on SQUARED distances, fp32 compare)
 *
 * The arithmetic of knnMatch lives in OpenCV (opencv-python>=4.8, pyproject.toml:31), which is
 * NOT vendored under /root/reference.  We restate its published behaviour (modules/core/src/
 * batch_distance.cpp + modules/features2d/src/matchers.cpp, from documentation/memory):
 *   - L2 distance between float32 rows = sqrtf( float sum_k (a_k-b_k)^2 )
 *   - k=2 nearest by insertion with strict '<' against the current K-th, '>' when shifting:
 *     among equal float distances the LOWEST train index stays in front
 *   - ratio test is done by the reference in Python doubles on the float32 distances:
 *         double(m.distance) < ratio * double(n.distance)            (matching.py:752)
 * PARITY STATUS: "parity unpinned" against cv2 itself (no golden vectors exist in the reference
 * for this path).  For integer-valued descriptors in [0,255] (the HAHOG/SIFT uint8
 * round trip, features.py:526-534,259-262) every partial sum is an integer < 2^24, hence exactly
 * representable in fp32 under ANY summation order -- so on that domain this restatement is
 * independent of OpenCV's SIMD accumulation order and the result is well defined.
 *
 * Build: see oracle/Makefile (gcc -O3 -march=native -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* float sum of squared differences; 4 accumulators of 8 lanes, the structure of OpenCV's
 * normL2Sqr_ on an AVX2 build (4 x v_float32x8, then lane reduction, then scalar tail). */
typedef float v8f __attribute__((vector_size(32), aligned(4)));

static inline float l2sqr_f32(const float *a, const float *b, int n) {
  v8f d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
  int j = 0;
  for (; j + 32 <= n; j += 32) {
    v8f t0 = *(const v8f *)(a + j) - *(const v8f *)(b + j);
    v8f t1 = *(const v8f *)(a + j + 8) - *(const v8f *)(b + j + 8);
    v8f t2 = *(const v8f *)(a + j + 16) - *(const v8f *)(b + j + 16);
    v8f t3 = *(const v8f *)(a + j + 24) - *(const v8f *)(b + j + 24);
    d0 += t0 * t0;
    d1 += t1 * t1;
    d2 += t2 * t2;
    d3 += t3 * t3;
  }
  v8f s = (d0 + d1) + (d2 + d3);
  float d = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  for (; j < n; j++) {
    float t = a[j] - b[j];
    d += t * t;
  }
  return d;
}

/* One direction: for every row i of f1 (n1 x dim) find the two nearest rows of f2 (n2 x dim).
 * out_idx[i] = best train index (or -1 when n2 < 2: knnMatch returns <2 neighbours and
 * matching.py:750 drops the row), out_d1[i], out_d2[i] = the float32 L2 distances.
 * Restates BFMatcher::knnMatchImpl -> batchDistance(K=2). */
void oracle_knn2_l2(const float *f1, int n1, const float *f2, int n2, int dim, int *out_idx,
                    float *out_d1, float *out_d2, float *out_s1, float *out_s2) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < n1; i++) {
    float bd0 = INFINITY, bd1 = INFINITY, bs0 = INFINITY, bs1 = INFINITY;
    int bi0 = -1, bi1 = -1;
    const float *a = f1 + (size_t)i * dim;
    for (int j = 0; j < n2; j++) {
      float sq = l2sqr_f32(a, f2 + (size_t)j * dim, dim);
      float d = sqrtf(sq);
      /* batchDistance top-K insertion: if (d < dist[K-1]) shift while dist[k] > d */
      if (d < bd1) {
        if (bd0 > d) {
          bd1 = bd0;
          bs1 = bs0;
          bi1 = bi0;
          bd0 = d;
          bs0 = sq;
          bi0 = j;
        } else {
          bd1 = d;
          bs1 = sq;
          bi1 = j;
        }
      }
    }
    (void)bi1;
    if (n2 < 2) bi0 = -1;
    out_idx[i] = bi0;
    out_d1[i] = bd0;
    out_d2[i] = bd1;
    if (out_s1) out_s1[i] = bs0;
    if (out_s2) out_s2[i] = bs1;
  }
}

/* matching.py:723-756: returns good[i] = j or -1.  squared_mode restates match_flann's test
 * (matching.py:695-696): numpy float32 array * python float stays float32:
 *     d0 < float32(ratio**2) * d1     on SQUARED distances. */
/* The exact limit of index.knnSearch(f2, 2) (matching.py:694): FLANN computes, compares and returns SQUARED L2 distances,
 * no square root anywhere; two nearest by the same insertion rule (lowest index first among equals). */
static void knn2_squared(const float *f1, int n1, const float *f2, int n2, int dim, int *out_idx, float *out_s1, float *out_s2) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < n1; i++) {
    float bs0 = INFINITY, bs1 = INFINITY;
    int bi0 = -1;
    const float *a = f1 + (size_t)i * dim;
    for (int j = 0; j < n2; j++) {
      float sq = l2sqr_f32(a, f2 + (size_t)j * dim, dim);
      if (sq < bs1) {
        if (bs0 > sq) {
          bs1 = bs0;
          bs0 = sq;
          bi0 = j;
        } else {
          bs1 = sq;
        }
      }
    }
    out_idx[i] = n2 < 2 ? -1 : bi0;
    out_s1[i] = bs0;
    out_s2[i] = bs1;
  }
}

void oracle_match_brute_force(const float *f1, int n1, const float *f2, int n2, int dim,
                              double ratio, int squared_mode, int *good) {
  int *idx = (int *)malloc(sizeof(int) * (size_t)(n1 > 0 ? n1 : 1));
  float *d1 = (float *)malloc(sizeof(float) * (size_t)(n1 > 0 ? n1 : 1));
  float *d2 = (float *)malloc(sizeof(float) * (size_t)(n1 > 0 ? n1 : 1));
  float *s1 = (float *)malloc(sizeof(float) * (size_t)(n1 > 0 ? n1 : 1));
  float *s2 = (float *)malloc(sizeof(float) * (size_t)(n1 > 0 ? n1 : 1));
  if (squared_mode)
    knn2_squared(f1, n1, f2, n2, dim, idx, s1, s2);
  else
    oracle_knn2_l2(f1, n1, f2, n2, dim, idx, d1, d2, s1, s2);
  for (int i = 0; i < n1; i++) {
    int ok = 0;
    if (idx[i] >= 0) {
      if (squared_mode) {
        float r2 = (float)(ratio * ratio);
        ok = s1[i] < r2 * s2[i];
      } else {
        ok = (double)d1[i] < ratio * (double)d2[i];
      }
    }
    good[i] = ok ? idx[i] : -1;
  }
  free(idx);
  free(d1);
  free(d2);
  free(s1);
  free(s2);
}

/* matching.py:683-697 match_flann(index of image 1, f2): queries are the rows of f2, the index holds f1; returns (index row,
 * query row) in QUERY order, as `list(zip(results[good, 0], good.nonzero()[0]))` does.  Returns the number of pairs. */
int oracle_match_flann(const float *f1, int n1, const float *f2, int n2, int dim, double ratio, int *out_pairs, int cap) {
  int *g = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
  oracle_match_brute_force(f2, n2, f1, n1, dim, ratio, 1, g);
  int n = 0;
  for (int j = 0; j < n2; j++)
    if (g[j] >= 0) {
      if (n < cap) {
        out_pairs[2 * n] = g[j];
        out_pairs[2 * n + 1] = j;
      }
      n++;
    }
  free(g);
  return n;
}

/* matching.py:759-777 (and, with squared_mode, match_flann_symmetric :700-720).  Output pairs (i, j) sorted by (i, j) (the
 * reference returns an unordered python set; we canonicalise).  Returns the number of pairs written (<= cap). */
int oracle_match_brute_force_symmetric(const float *fi, int ni, const float *fj, int nj, int dim,
                                       double ratio, int squared_mode, int *out_pairs, int cap) {
  int *gij = (int *)malloc(sizeof(int) * (size_t)(ni > 0 ? ni : 1));
  int *gji = (int *)malloc(sizeof(int) * (size_t)(nj > 0 ? nj : 1));
  oracle_match_brute_force(fi, ni, fj, nj, dim, ratio, squared_mode, gij);
  oracle_match_brute_force(fj, nj, fi, ni, dim, ratio, squared_mode, gji);
  int n = 0;
  for (int i = 0; i < ni; i++) {
    int j = gij[i];
    if (j >= 0 && gji[j] == i) {
      if (n < cap) {
        out_pairs[2 * n] = i;
        out_pairs[2 * n + 1] = j;
      }
      n++;
    }
  }
  free(gij);
  free(gji);
  return n;
}

/* matching.py:737-740: uint8 descriptors -> cv2.DescriptorMatcher_create("BruteForce-Hamming"): batchDistance with NORM_HAMMING gives
 * int distances (popcount of the xor over the row's bytes), the same K = 2 insertion as above on ints, DMatch.distance = float(int);
 * Lowe's test in Python doubles (matching.py:752).  good[i] = j or -1.  parity unpinned vs cv2 (third party, restated). */
void oracle_match_hamming(const uint8_t *f1, int n1, const uint8_t *f2, int n2, int width, double ratio, int *good) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < n1; i++) {
    int bd0 = 0x7fffffff, bd1 = 0x7fffffff, bi0 = -1;
    const uint8_t *a = f1 + (size_t)i * width;
    for (int j = 0; j < n2; j++) {
      const uint8_t *b = f2 + (size_t)j * width;
      int d = 0;
      for (int k = 0; k < width; k++) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
      if (d < bd1) {
        if (bd0 > d) {
          bd1 = bd0;
          bd0 = d;
          bi0 = j;
        } else {
          bd1 = d;
        }
      }
    }
    good[i] = (n2 >= 2 && (double)(float)bd0 < ratio * (double)(float)bd1) ? bi0 : -1;
  }
}

int oracle_match_hamming_symmetric(const uint8_t *fi, int ni, const uint8_t *fj, int nj, int width, double ratio, int *out_pairs, int cap) {
  int *gij = (int *)malloc(sizeof(int) * (size_t)(ni > 0 ? ni : 1));
  int *gji = (int *)malloc(sizeof(int) * (size_t)(nj > 0 ? nj : 1));
  oracle_match_hamming(fi, ni, fj, nj, width, ratio, gij);
  oracle_match_hamming(fj, nj, fi, ni, width, ratio, gji);
  int n = 0;
  for (int i = 0; i < ni; i++) {
    int j = gij[i];
    if (j >= 0 && gji[j] == i) {
      if (n < cap) {
        out_pairs[2 * n] = i;
        out_pairs[2 * n + 1] = j;
      }
      n++;
    }
  }
  free(gij);
  free(gji);
  return n;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- the CPU path someone would actually write (bench.py's second cpu_baseline figure, not a parity reference) -----------------------
 * For INTEGER-VALUED descriptors in [0, 255] (HAHOG / SIFT uchar levels held in float32) the squared distance in GEMM form,
 * |a|^2 + |b|^2 - 2 a.b, is exact in float32 -- every product <= 65 025, every partial sum < 2^24 -- and equal, bit for bit, to the
 * sum of squared differences of the direct form above.  One 2000 x 2000 product then serves BOTH directions of the symmetric matcher:
 * a register-blocked micro-kernel (4 rows x 64 columns of accumulators in sixteen 512-bit registers, B transposed once per pair so
 * that the columns vectorise), the same top-2 insertion as oracle_knn2_l2 on the finished tile for the row's and for the column's
 * running best (i and j visited in increasing order: the lowest index wins among equals, as cv2's batchDistance).  Returns what
 * oracle_match_brute_force_symmetric(..., squared_mode = 0) returns; tests/test_oracle_matching.py asserts that. */
typedef float v16f __attribute__((vector_size(64), aligned(4)));
#pragma GCC push_options
#pragma GCC optimize("fp-contract=fast") /* (exact either way: the products and sums are integers below 2^24) */
static void gemm_tile_4x64(const float *a0, const float *a1, const float *a2, const float *a3, const float *bt, int ldb, int dim, v16f acc[4][4]) {
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) acc[r][c] = (v16f){0};
  for (int k = 0; k < dim; k++) {
    const float *brow = bt + (size_t)k * ldb;
    v16f b[4];
    for (int c = 0; c < 4; c++) __builtin_memcpy(&b[c], brow + 16 * c, sizeof(v16f));
    const float av[4] = {a0[k], a1[k], a2[k], a3[k]};
    for (int r = 0; r < 4; r++) {
      v16f a = (v16f){0} + av[r];
      for (int c = 0; c < 4; c++) acc[r][c] += a * b[c];
    }
  }
}
#pragma GCC pop_options
/* the insertion of oracle_knn2_l2 on (distance, squared distance): bs1 = the squared distance behind bd1, the pre-test's threshold */
static inline void top2_insert(float d, float sq, int j, float *bd0, float *bd1, float *bs0, float *bs1, int *bi0) {
  if (d < *bd1) {
    if (*bd0 > d) {
      *bd1 = *bd0;
      *bs1 = *bs0;
      *bd0 = d;
      *bs0 = sq;
      *bi0 = j;
    } else {
      *bd1 = d;
      *bs1 = sq;
    }
  }
}
int oracle_match_brute_force_symmetric_gemm(const float *fi, int ni, const float *fj, int nj, int dim, double ratio, int *out_pairs, int cap) {
  if (ni < 2 || nj < 2) return 0;
  const int njp = (nj + 63) / 64 * 64, nip = (ni + 3) / 4 * 4;
  float *bt = (float *)calloc((size_t)dim * njp, sizeof(float)), *nb = (float *)calloc((size_t)njp, sizeof(float));
  float *ap = (float *)calloc((size_t)nip * dim, sizeof(float)), *na = (float *)calloc((size_t)nip, sizeof(float));
  for (int j = 0; j < nj; j++) {
    float s = 0.f;
    for (int k = 0; k < dim; k++) {
      const float v = fj[(size_t)j * dim + k];
      bt[(size_t)k * njp + j] = v;
      s += v * v;
    }
    nb[j] = s;
  }
  for (int i = 0; i < ni; i++) {
    float s = 0.f;
    for (int k = 0; k < dim; k++) {
      const float v = fi[(size_t)i * dim + k];
      ap[(size_t)i * dim + k] = v;
      s += v * v;
    }
    na[i] = s;
  }
  /* per row / column: best and second distance, their squares, the best index */
  float *rd0 = (float *)malloc(sizeof(float) * nip * 4), *cd0 = (float *)malloc(sizeof(float) * njp * 4);
  float *rd1 = rd0 + nip, *rs0 = rd0 + 2 * nip, *rs1 = rd0 + 3 * nip, *cd1 = cd0 + njp, *cs0 = cd0 + 2 * njp, *cs1 = cd0 + 3 * njp;
  int *ri = (int *)malloc(sizeof(int) * nip), *ci = (int *)malloc(sizeof(int) * njp);
  for (int i = 0; i < nip; i++) { rd0[i] = rd1[i] = rs0[i] = rs1[i] = INFINITY; ri[i] = -1; }
  for (int j = 0; j < njp; j++) { cd0[j] = cd1[j] = cs0[j] = cs1[j] = INFINITY; ci[j] = -1; }
  for (int i0 = 0; i0 < ni; i0 += 4)
    for (int j0 = 0; j0 < nj; j0 += 64) {
      v16f acc[4][4];
      gemm_tile_4x64(ap + (size_t)i0 * dim, ap + (size_t)(i0 + 1) * dim, ap + (size_t)(i0 + 2) * dim, ap + (size_t)(i0 + 3) * dim, bt + j0, njp, dim, acc);
      const int jn = nj - j0 < 64 ? nj - j0 : 64;
      for (int r = 0; r < 4 && i0 + r < ni; r++) {
        const int i = i0 + r;
        float sqt[64];
        const float *row = (const float *)&acc[r][0];
        for (int c = 0; c < 64; c++) sqt[c] = (na[i] + nb[j0 + c]) - 2.0f * row[c]; /* vectorised */
        for (int c = 0; c < jn; c++) {
          const int j = j0 + c;
          const float sq = sqt[c];
          /* sqrtf is monotone: a squared distance above the one behind the running second best cannot enter (d >= bd1) */
          if (sq <= rs1[i] || sq <= cs1[j]) {
            const float d = sqrtf(sq);
            top2_insert(d, sq, j, &rd0[i], &rd1[i], &rs0[i], &rs1[i], &ri[i]);
            top2_insert(d, sq, i, &cd0[j], &cd1[j], &cs0[j], &cs1[j], &ci[j]);
          }
        }
      }
    }
  int n = 0;
  for (int i = 0; i < ni; i++) {
    const int j = ri[i];
    if (j < 0 || !((double)rd0[i] < ratio * (double)rd1[i])) continue;
    if (ci[j] != i || !((double)cd0[j] < ratio * (double)cd1[j])) continue;
    if (n < cap) {
      out_pairs[2 * n] = i;
      out_pairs[2 * n + 1] = j;
    }
    n++;
  }
  free(bt); free(nb); free(ap); free(na); free(rd0); free(cd0); free(ri); free(ci);
  return n;
}
/* the descriptor stage of `n_pairs` pairs of a packed store in GEMM form, OpenMP over pairs: counts only (a timing leg) */
void oracle_match_pairs_gemm(const float *desc, const int64_t *offsets, int dim, const int *pairs, int n_pairs, double ratio, int *counts, int *out_pairs,
                             int cap_per_pair) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int p = 0; p < n_pairs; p++) {
    const int a = pairs[2 * p], b = pairs[2 * p + 1];
    counts[p] = oracle_match_brute_force_symmetric_gemm(desc + offsets[a] * dim, (int)(offsets[a + 1] - offsets[a]), desc + offsets[b] * dim,
                                                        (int)(offsets[b + 1] - offsets[b]), dim, ratio, out_pairs + (size_t)2 * cap_per_pair * p, cap_per_pair);
  }
}
