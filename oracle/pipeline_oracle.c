/*
 * pipeline_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped).
 *
 * Restates the per-pair driver opensfm/matching.py:563-634 `match()` for the configuration the
 * north star names (matcher_type BRUTEFORCE, symmetric_matching True, pinhole cameras with
 * k1 = k2 = 0 -> robust_match_fundamental, matching.py:906-929), and the fan-out over pairs
 * opensfm/matching.py:63-98 `match_images_with_pairs` (here: OpenMP over pairs instead of the
 * reference's joblib thread pool, context.py:47-67).
 *
 *   M   = match_brute_force_symmetric(D1, D2)              matching.py:759-777
 *   if |M| < robust_matching_min_match -> []               matching.py:590-598
 *   F, inl = robust_match_fundamental(P1, P2, M)           matching.py:780-802
 *   if F is None or F[2,2] == 0 -> []                      matching.py:798-800
 *   if |inl| < robust_matching_min_match -> []             matching.py:632-634
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_match_brute_force_symmetric(const float *fi, int ni, const float *fj, int nj, int dim,
                                       double ratio, int squared_mode, int *out_pairs, int cap);
int oracle_find_fundamental_ransac(const double *p1, const double *p2, int n, double thr,
                                   double conf, int max_iters, double *F, uint8_t *mask,
                                   int *iters_run);

/* stage: 0 = descriptor stage only (symmetric matches), 1 = + min-match gate + F-RANSAC + gate.
 * out_pairs: cap x 2 ints.  returns number of matches (0 for "failed" pair). */
int oracle_match_pair(const float *d1, const double *p1, int n1, const float *d2, const double *p2,
                      int n2, int dim, double ratio, int min_match, double thr, double conf,
                      int stage, int *out_pairs, int cap) {
  if (n1 < 2 || n2 < 2) return 0; /* matching.py:363-374 */
  int mcap = n1 < n2 ? n1 : n2;
  int *m = (int *)malloc(sizeof(int) * 2 * (size_t)(mcap > 0 ? mcap : 1));
  int nm = oracle_match_brute_force_symmetric(d1, n1, d2, n2, dim, ratio, 0, m, mcap);
  int nout = 0;
  if (stage == 0) {
    nout = nm;
    for (int k = 0; k < nm && k < cap; k++) {
      out_pairs[2 * k] = m[2 * k];
      out_pairs[2 * k + 1] = m[2 * k + 1];
    }
  } else if (nm >= min_match && nm >= 8) {
    double *x1 = (double *)malloc(sizeof(double) * 2 * (size_t)nm);
    double *x2 = (double *)malloc(sizeof(double) * 2 * (size_t)nm);
    uint8_t *mask = (uint8_t *)malloc((size_t)nm);
    for (int k = 0; k < nm; k++) {
      x1[2 * k] = p1[2 * m[2 * k]];
      x1[2 * k + 1] = p1[2 * m[2 * k] + 1];
      x2[2 * k] = p2[2 * m[2 * k + 1]];
      x2[2 * k + 1] = p2[2 * m[2 * k + 1] + 1];
    }
    double F[9];
    int it;
    int ok = oracle_find_fundamental_ransac(x1, x2, nm, thr, conf, 1000, F, mask, &it);
    if (ok == 1 && F[8] != 0.0) {
      int cnt = 0;
      for (int k = 0; k < nm; k++) cnt += mask[k] != 0;
      if (cnt >= min_match) {
        for (int k = 0; k < nm; k++)
          if (mask[k]) {
            if (nout < cap) {
              out_pairs[2 * nout] = m[2 * k];
              out_pairs[2 * nout + 1] = m[2 * k + 1];
            }
            nout++;
          }
      }
    }
    free(x1);
    free(x2);
    free(mask);
  }
  free(m);
  return nout;
}

/* match_images_with_pairs over a packed store: desc (sum n x dim floats), pts (sum n x 2 doubles),
 * offsets[n_images+1].  pairs: n_pairs x 2 image indices.  counts[n_pairs] out; matches written to
 * out_pairs + 2*cap_per_pair*pair.  OpenMP over pairs. */
void oracle_match_pairs(const float *desc, const double *pts, const int64_t *offsets, int dim,
                        const int *pairs, int n_pairs, double ratio, int min_match, double thr,
                        double conf, int stage, int *counts, int *out_pairs, int cap_per_pair) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int p = 0; p < n_pairs; p++) {
    int a = pairs[2 * p], b = pairs[2 * p + 1];
    int n1 = (int)(offsets[a + 1] - offsets[a]), n2 = (int)(offsets[b + 1] - offsets[b]);
    counts[p] = oracle_match_pair(desc + offsets[a] * dim, pts + offsets[a] * 2, n1,
                                  desc + offsets[b] * dim, pts + offsets[b] * 2, n2, dim, ratio,
                                  min_match, thr, conf, stage,
                                  out_pairs + (size_t)2 * cap_per_pair * p, cap_per_pair);
  }
}
