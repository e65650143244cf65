/* guided_oracle.c -- CPU restatement of guided (epipolar-masked) descriptor matching.
 *
 * TEST INFRASTRUCTURE ONLY, GROUNDWORK for the second half of row M-a9 / SURVEY.md 8f-3 (no product counterpart yet:
 * opensfm_amd.matching raises NotImplementedError when poses are given).
 * reference: matching._match_descriptors_guided_impl (opensfm/matching.py:260-337) =
 *   compute_inliers_bearing_epipolar (matching.py:847-868): bearings cast to float32, then
 *     geometry::EpipolarAngleTwoBearingsMany (opensfm/src/geometry/src/triangulation.cc:195-219) in doubles:
 *       t^ = t / |t|, b2w_j = R b2_j, e1_i = (t^ x b1_i)^, e2_j = (t^ x b2w_j)^   (Eigen normalized(): unchanged when the norm is 0)
 *       angle_ij = pi/2 - acos((|e1_i . b2w_j| + |b1_i . e2_j|) / 2),  mask_ij = angle_ij < guided_matching_threshold
 *   match_brute_force_symmetric(d1, d2, config, mask) (matching.py:723-777): cv2 knnMatch(k = 2, masks = [mask]) only ranks the
 *     train descriptors the mask allows; queries with fewer than two allowed neighbours are dropped (`len(match) == 2`);
 *     the reverse direction uses the transposed mask.
 * Dot products are written left to right (Eigen's order / contraction in the reference build is unknown): parity with a
 * reference binary is unpinned in the last bit of angle_ij, i.e. for candidate pairs within ~1e-16 rad of the threshold.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static void normalized3(double *v) {
  const double z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (z > 0) {
    const double n = sqrt(z);
    v[0] /= n; v[1] /= n; v[2] /= n;
  }
}
static void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* b1: n1 x 3, b2: n2 x 3 (float32, as the reference casts them); R (row-major) = pose.get_R_cam_to_world(), t = pose.get_origin()
 * of the second camera relative to the first; angle (n1 x n2, may be NULL), mask (n1 x n2). */
void oracle_epipolar_mask(const float *b1, int n1, const float *b2, int n2, const double *R, const double *t, double threshold,
                          double *angle, uint8_t *mask) {
  double tn[3] = {t[0], t[1], t[2]};
  normalized3(tn);
  double *w = (double *)malloc(sizeof(double) * 3 * (size_t)(n2 > 0 ? n2 : 1));
  double *e2 = (double *)malloc(sizeof(double) * 3 * (size_t)(n2 > 0 ? n2 : 1));
  for (int j = 0; j < n2; j++) {
    const double y[3] = {(double)b2[3 * j], (double)b2[3 * j + 1], (double)b2[3 * j + 2]};
    for (int a = 0; a < 3; a++) w[3 * j + a] = y[0] * R[3 * a] + y[1] * R[3 * a + 1] + y[2] * R[3 * a + 2]; /* row of b2 * R^T */
    cross3(tn, w + 3 * j, e2 + 3 * j);
    normalized3(e2 + 3 * j);
  }
  for (int i = 0; i < n1; i++) {
    const double x[3] = {(double)b1[3 * i], (double)b1[3 * i + 1], (double)b1[3 * i + 2]};
    double e1[3];
    cross3(tn, x, e1);
    normalized3(e1);
    for (int j = 0; j < n2; j++) {
      const double a = fabs(e1[0] * w[3 * j] + e1[1] * w[3 * j + 1] + e1[2] * w[3 * j + 2]);
      const double b = fabs(x[0] * e2[3 * j] + x[1] * e2[3 * j + 1] + x[2] * e2[3 * j + 2]);
      const double ang = M_PI / 2.0 - acos((a + b) / 2.0);
      if (angle) angle[(size_t)i * n2 + j] = ang;
      mask[(size_t)i * n2 + j] = ang < threshold;
    }
  }
  free(w);
  free(e2);
}

/* cv2's normL2Sqr_ on an AVX2 build, as oracle/match_oracle.c (l2sqr_f32) restates it: four 8-lane accumulators over blocks of 32
 * dimensions, (d0 + d1) + (d2 + d3), the lanes pairwise, then the scalar tail.  Exact -- hence order-free -- for integer-valued
 * descriptors; for float descriptors (root-SIFT) this order is what "the float distance" means in this repository. */
static float l2sqr(const float *a, const float *b, int n) {
  float acc[4][8] = {{0}};
  int j = 0;
  for (; j + 32 <= n; j += 32)
    for (int v = 0; v < 4; v++)
      for (int l = 0; l < 8; l++) {
        const float t = a[j + 8 * v + l] - b[j + 8 * v + l];
        acc[v][l] += t * t;
      }
  float sv[8];
  for (int l = 0; l < 8; l++) sv[l] = (acc[0][l] + acc[1][l]) + (acc[2][l] + acc[3][l]);
  float d = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
  for (; j < n; j++) {
    const float t = a[j] - b[j];
    d += t * t;
  }
  return d;
}
/* one direction; mask element of (query i, train j) = mask[i * si + j * sj]; good[i] = j or -1 */
static void match_masked(const float *f1, int n1, const float *f2, int n2, int dim, double ratio, const uint8_t *mask, size_t si, size_t sj,
                         int *good) {
  for (int i = 0; i < n1; i++) {
    float bd0 = INFINITY, bd1 = INFINITY;
    int bi0 = -1, bi1 = -1;
    for (int j = 0; j < n2; j++) {
      if (!mask[(size_t)i * si + (size_t)j * sj]) continue;
      const float d = sqrtf(l2sqr(f1 + (size_t)i * dim, f2 + (size_t)j * dim, dim));
      if (d < bd1) { /* same top-2 insertion as oracle_knn2_l2 (match_oracle.c) */
        if (bd0 > d) { bd1 = bd0; bi1 = bi0; bd0 = d; bi0 = j; }
        else { bd1 = d; bi1 = j; }
      }
    }
    good[i] = (bi1 >= 0 && (double)bd0 < ratio * (double)bd1) ? bi0 : -1;
  }
}
void oracle_match_brute_force_masked(const float *f1, int n1, const float *f2, int n2, int dim, double ratio, const uint8_t *mask, int *good) {
  match_masked(f1, n1, f2, n2, dim, ratio, mask, (size_t)n2, 1, good);
}
/* pairs (i, j) sorted by (i, j); returns their number (<= cap written) */
int oracle_match_brute_force_symmetric_masked(const float *fi, int ni, const float *fj, int nj, int dim, double ratio, const uint8_t *mask,
                                              int *out_pairs, int cap) {
  int *gij = (int *)malloc(sizeof(int) * (size_t)(ni > 0 ? ni : 1)), *gji = (int *)malloc(sizeof(int) * (size_t)(nj > 0 ? nj : 1));
  match_masked(fi, ni, fj, nj, dim, ratio, mask, (size_t)nj, 1, gij);
  match_masked(fj, nj, fi, ni, dim, ratio, mask, 1, (size_t)nj, gji); /* maskij.T */
  int n = 0;
  for (int i = 0; i < ni; i++) {
    const int j = gij[i];
    if (j >= 0 && gji[j] == i) {
      if (n < cap) { out_pairs[2 * n] = i; out_pairs[2 * n + 1] = j; }
      n++;
    }
  }
  free(gij);
  free(gji);
  return n;
}

/* cv2 knnMatch(k = 2, masks) as numbers (for tests/test_reference_flow.py, which hands them to the reference's own Python):
 * idx[2 i], idx[2 i + 1] = the two nearest allowed train indices of query i (-1 = fewer candidates), dist likewise; mask may be NULL */
void oracle_knn2_masked(const float *f1, int n1, const float *f2, int n2, int dim, const uint8_t *mask, int *idx, float *dist) {
  for (int i = 0; i < n1; i++) {
    float bd0 = INFINITY, bd1 = INFINITY;
    int bi0 = -1, bi1 = -1;
    for (int j = 0; j < n2; j++) {
      if (mask && !mask[(size_t)i * n2 + j]) continue;
      const float d = sqrtf(l2sqr(f1 + (size_t)i * dim, f2 + (size_t)j * dim, dim));
      if (d < bd1) {
        if (bd0 > d) { bd1 = bd0; bi1 = bi0; bd0 = d; bi0 = j; }
        else { bd1 = d; bi1 = j; }
      }
    }
    idx[2 * i] = bi0; idx[2 * i + 1] = bi1;
    dist[2 * i] = bd0; dist[2 * i + 1] = bd1;
  }
}
