/*
 * words_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, never shipped, never on the product path).
 *
 * CPU restatement of the WORDS matcher and of the VLAD descriptor:
 *
 *   opensfm/src/features/src/matching.cc:16-22    DistanceL2        float sum of (a-b)^2 in index order, then sqrt
 *   opensfm/src/features/src/matching.cc:24-72    MatchUsingWords   multimap word -> features of image 2 (equal keys keep their
 *        insertion order = ascending feature index); for every feature of image 1 its words in order, every feature under a word is
 *        checked ("<" against the best, else "<" against the second), the word loop stops AFTER the word that brought the count to
 *        max_checks; Lowe: best < ratio * second in float
 *   opensfm/matching.py:637-680                   match_words / match_words_symmetric (words2[:, 0] only; set intersection)
 *   opensfm/src/features/src/matching.cc:93-124   compute_vlad_descriptor  nearest centre by squared distance (first minimum), the
 *        residuals are accumulated per centre in feature order
 *   opensfm/src/features/src/matching.cc:126-152  compute_vlad_distances   L2 norm of the difference of two descriptors
 * The reference file needs OpenCV / pybind11 / Eigen to compile, none of which is here: not buildable into oracle/_ref.
 * PARITY STATUS: pinned by the reference's own test of this call (opensfm/test/test_matching.py:50-68, restated in
 * tests/test_oracle_words.py: 1000 noisy copies must match i -> i); the float rounding of the compiled reference (FMA contraction,
 * Eigen's vectorised squaredNorm) is not reproducible here -- the oracle fixes it as mul, then add, in index order
 * (-ffp-contract=off), which the HIP kernels follow bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float distance_l2(const float *pa, const float *pb, int n) {
  float distance = 0;
  for (int i = 0; i < n; ++i) {
    float d = pa[i] - pb[i];
    float sq = d * d;
    distance = distance + sq;
  }
  return sqrtf(distance);
}

typedef struct {
  int32_t word, idx;
} word_entry;

static int cmp_entry(const void *a, const void *b) {
  const word_entry *x = (const word_entry *)a, *y = (const word_entry *)b;
  if (x->word != y->word) return x->word < y->word ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* f1: n1 x dim, w1: n1 x nw (row-major), f2: n2 x dim, w2: n2 primary words.  matches: up to n1 (i, j) pairs.  Returns the count. */
int oracle_match_words(const float *f1, const int32_t *w1, int n1, int nw, const float *f2, const int32_t *w2, int n2, int dim,
                       float lowes_ratio, int max_checks, int32_t *matches) {
  word_entry *index2 = (word_entry *)malloc(sizeof(word_entry) * (size_t)(n2 > 0 ? n2 : 1));
  for (int i = 0; i < n2; ++i) {
    index2[i].word = w2[i];
    index2[i].idx = i;
  }
  qsort(index2, (size_t)n2, sizeof(word_entry), cmp_entry);
  int count = 0;
  for (int i = 0; i < n1; ++i) {
    int best_match = -1, checks = 0;
    float best = INFINITY, second = INFINITY;
    for (int j = 0; j < nw; ++j) {
      const int word = w1[(size_t)i * nw + j];
      int lo = 0, hi = n2; /* first entry with .word >= word */
      while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (index2[mid].word < word)
          lo = mid + 1;
        else
          hi = mid;
      }
      for (int e = lo; e < n2 && index2[e].word == word; ++e) {
        const int match = index2[e].idx;
        const float distance = distance_l2(f1 + (size_t)i * dim, f2 + (size_t)match * dim, dim);
        if (distance < best) {
          second = best;
          best = distance;
          best_match = match;
        } else if (distance < second) {
          second = distance;
        }
        checks++;
      }
      if (checks >= max_checks) break;
    }
    if (best < lowes_ratio * second) {
      matches[2 * count] = i;
      matches[2 * count + 1] = best_match;
      count++;
    }
  }
  free(index2);
  return count;
}

/* features: n x dim, centers: nc x dim -> out: nc * dim */
void oracle_vlad_descriptor(const float *features, int n, const float *centers, int nc, int dim, float *out) {
  for (int k = 0; k < nc * dim; ++k) out[k] = 0.0f;
  for (int i = 0; i < n; ++i) {
    const float *f = features + (size_t)i * dim;
    float best_distance = 3.402823466e+38F;
    int best_center = -1;
    for (int j = 0; j < nc; ++j) {
      const float *c = centers + (size_t)j * dim;
      float s = 0;
      for (int k = 0; k < dim; ++k) {
        float d = f[k] - c[k];
        float sq = d * d;
        s = s + sq;
      }
      if (s < best_distance) {
        best_distance = s;
        best_center = j;
      }
    }
    if (best_center < 0) continue; /* NaN features: the reference would index segment(-dim) -- undefined; skipped here */
    const float *c = centers + (size_t)best_center * dim;
    float *o = out + (size_t)best_center * dim;
    for (int k = 0; k < dim; ++k) {
      float d = f[k] - c[k];
      o[k] = o[k] + d;
    }
  }
}

/* distances of descriptor `ref` (len) to m others (m x len) */
void oracle_vlad_distances(const float *ref, const float *others, int m, int len, double *out) {
  for (int j = 0; j < m; ++j) {
    float s = 0;
    for (int k = 0; k < len; ++k) {
      float d = ref[k] - others[(size_t)j * len + k];
      float sq = d * d;
      s = s + sq;
    }
    out[j] = (double)sqrtf(s);
  }
}

/* ---------------------------------------------------------------------------------------------------------------------------------
 * BoW affinity: pairs_selection.bow_distances (opensfm/pairs_selection.py:690-708) = np.fabs(h - h2).sum() over float64 histograms
 * (bow.py:34-36).  The value depends on numpy's summation order, restated here (numpy/_core/src/umath/loops_utils.h.src,
 * DOUBLE_pairwise_sum, and the 8192-element buffering of the reduction):
 *   sum = fold over chunks of 8192 elements of   res = res + pairwise(chunk),   res starting at 0;
 *   pairwise(a, n): n < 8: sequential from 0;  n <= 128: eight accumulators r[j] = a[j], r[j] += a[i + j] for i = 8, 16, ... while
 *   i < n - n % 8, then ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the n % 8 tail sequentially;  else split at
 *   n2 = n / 2 rounded down to a multiple of 8: pairwise(a, n2) + pairwise(a + n2, n - n2).
 * Pinned against numpy itself (tests/test_oracle_words.py runs the reference's bow_distances from its file).
 * --------------------------------------------------------------------------------------------------------------------------------- */
static double np_pairwise_absdiff(const double *h, const double *g, long n) {
  if (n < 8) {
    double res = 0.0;
    for (long i = 0; i < n; i++) res += fabs(h[i] - g[i]);
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = fabs(h[j] - g[j]);
    long i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; j++) r[j] += fabs(h[i + j] - g[i + j]);
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += fabs(h[i] - g[i]);
    return res;
  }
  long n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_absdiff(h, g, n2) + np_pairwise_absdiff(h + n2, g + n2, n - n2);
}

void oracle_bow_distances(const double *ref, const double *others, int m, int len, double *out) {
  for (int j = 0; j < m; ++j) {
    const double *g = others + (size_t)j * len;
    double res = 0.0;
    for (long c = 0; c < len; c += 8192) res = res + np_pairwise_absdiff(ref + c, g + c, (len - c) < 8192 ? (len - c) : 8192);
    out[j] = res;
  }
}
