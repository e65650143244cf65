// words.hip -- the bag-of-words side of pair matching on gfx950: the WORDS matcher, VLAD descriptors / distances and
// the GPS neighbour search of pair preselection.
//
// reference:
//   opensfm/src/features/src/matching.cc:16-88     DistanceL2 / MatchUsingWords / match_using_words   (osfm_match_words_pairs)
//   opensfm/matching.py:637-680                    match_words / match_words_symmetric
//   opensfm/src/features/src/matching.cc:93-152    compute_vlad_descriptor / compute_vlad_distances    (osfm_vlad_*)
//   opensfm/pairs_selection.py:154-212             match_candidates_by_distance: cKDTree.query(k, distance_upper_bound) (osfm_knn_points)
//
// All of it is gather / stream work on fp32 rows -- HBM / L2 bound, nothing for the matrix cores: one thread per query feature walks
// its candidates in the reference's order (the rule "first strictly smaller wins" and the float summation order are part of the
// result), the candidate rows come through L2 (an image's descriptors are 1 MB and every query of a pair reads from the same two).
// Float arithmetic follows the oracle (oracle/words_oracle.c): mul, then add, in index order; the library is built with
// -ffp-contract=off and HIP's correctly rounded sqrtf.
#include <math.h>

#include <algorithm>
#include <cmath>
#include <memory>
#include <numeric>
#include <vector>

#include "osfm_internal.h"

struct osfm_words_store {
  osfm_ctx *ctx = nullptr;
  int n_images = 0, dim = 0, nw = 0;
  std::vector<int32_t> counts;
  std::vector<int64_t> row_off;
  float *d_desc = nullptr;      // total x dim
  int32_t *d_words = nullptr;   // total x nw
  int32_t *d_skey = nullptr;    // total: primary words of an image's features, ascending
  int32_t *d_sidx = nullptr;    // total: feature index (within the image) of each sorted entry; equal words keep ascending index
  int64_t *d_row_off = nullptr; // n_images + 1
  int32_t *d_counts = nullptr;  // n_images
  int64_t total = 0;
  int max_count = 0;
};

namespace {

constexpr int kWT = 128;  // threads per block of the words kernel: one query each, its descriptor in registers

// sqrtf( sum_k (a_k - b_k)^2 ), float, index order (features/src/matching.cc:16-22)
template <int DIM>
__device__ __forceinline__ float distance_l2(const float (&q)[DIM], const float *__restrict__ pb) {
  float distance = 0.f;
#pragma unroll
  for (int k4 = 0; k4 < DIM; k4 += 4) {
    const float4 c = *(const float4 *)(pb + k4);
    const float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = q[k4 + e] - cc[e];
      const float sq = d * d;
      distance = distance + sq;
    }
  }
  return sqrtf(distance);
}

struct WordsArgs {
  const float *desc;
  const int32_t *words, *skey, *sidx;
  const int64_t *row_off;
  const int32_t *counts;
  const int32_t *pairs;
  int nw, max_count;
  float ratio;
  int max_checks;
  int32_t *best;  // [n_pairs][2][max_count]: direction 0 = features of image 1 -> image 2, direction 1 the other way
};

// grid: (query blocks, 2 directions, pairs)
template <int DIM>
__global__ void __launch_bounds__(kWT) words_match_kernel(WordsArgs a) {
  const long p = blockIdx.z;
  const int dir = blockIdx.y;
  const int imq = a.pairs[2 * p + dir], imt = a.pairs[2 * p + 1 - dir];
  const int nq = a.counts[imq], nt = a.counts[imt];
  const int i = blockIdx.x * kWT + threadIdx.x;
  if (i >= nq) return;
  const long rq = a.row_off[imq], rt = a.row_off[imt];
  float q[DIM];
  {
    const float *pq = a.desc + (rq + i) * DIM;
#pragma unroll
    for (int k4 = 0; k4 < DIM; k4 += 4) {
      const float4 v = *(const float4 *)(pq + k4);
      q[k4] = v.x;
      q[k4 + 1] = v.y;
      q[k4 + 2] = v.z;
      q[k4 + 3] = v.w;
    }
  }
  const int32_t *key = a.skey + rt, *idx = a.sidx + rt;
  int best_match = -1, checks = 0;
  float best = INFINITY, second = INFINITY;
  for (int j = 0; j < a.nw; ++j) {
    const int word = a.words[(rq + i) * a.nw + j];
    int lo = 0, hi = nt;  // first entry whose word is >= word
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (key[mid] < word)
        lo = mid + 1;
      else
        hi = mid;
    }
    for (int e = lo; e < nt && key[e] == word; ++e) {
      const int match = idx[e];
      const float distance = distance_l2<DIM>(q, a.desc + (rt + match) * DIM);
      if (distance < best) {
        second = best;
        best = distance;
        best_match = match;
      } else if (distance < second) {
        second = distance;
      }
      checks++;
    }
    if (checks >= a.max_checks) break;
  }
  a.best[(p * 2 + dir) * a.max_count + i] = (best < a.ratio * second) ? best_match : -1;
}

// one block per pair: ordered compaction of (i, j) over the features i of the pair's first image
__global__ void __launch_bounds__(256) words_emit_kernel(const int32_t *pairs, const int32_t *counts, const int32_t *best, int max_count,
                                                         int symmetric, int32_t *out_counts, int32_t *out_matches) {
  __shared__ int wsum[4];
  const long p = blockIdx.x;
  const int n1 = counts[pairs[2 * p]];
  const int32_t *b12 = best + (p * 2) * max_count, *b21 = best + (p * 2 + 1) * max_count;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int base = 0;
  for (int i0 = 0; i0 < n1; i0 += 256) {
    const int i = i0 + threadIdx.x;
    int j = -1;
    if (i < n1) j = b12[i];
    const bool m = j >= 0 && (!symmetric || b21[j] == i);
    const unsigned long long bal = __ballot(m);
    if (lane == 0) wsum[w] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      off += k < w ? wsum[k] : 0;
      total += wsum[k];
    }
    if (m) {
      const long o = (p * max_count + base + off + __popcll(bal & ((1ull << lane) - 1ull))) * 2;
      out_matches[o] = i;
      out_matches[o + 1] = j;
    }
    base += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) out_counts[p] = base;
}

// ---- VLAD ----
// nearest centre of every feature (first minimum of the squared distance), one thread per feature, centres staged in LDS
__global__ void __launch_bounds__(256) vlad_assign_kernel(const float *__restrict__ features, int n, const float *__restrict__ centers, int nc,
                                                          int dim, int32_t *__restrict__ assign) {
  extern __shared__ float cs[];
  for (int k = threadIdx.x; k < nc * dim; k += blockDim.x) cs[k] = centers[k];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *f = features + (size_t)i * dim;
  float best_distance = 3.402823466e+38F;
  int best_center = -1;
  for (int j = 0; j < nc; ++j) {
    float s = 0.f;
    for (int k = 0; k < dim; ++k) {
      const float d = f[k] - cs[j * dim + k];
      const float sq = d * d;
      s = s + sq;
    }
    if (s < best_distance) {
      best_distance = s;
      best_center = j;
    }
  }
  assign[i] = best_center;
}

// out[c][k] = sum over the features assigned to c, in feature order, of (f[k] - centre[c][k]): one thread per (c, k), so the float
// sums run in the reference's order
__global__ void __launch_bounds__(256) vlad_accumulate_kernel(const float *__restrict__ features, int n, const float *__restrict__ centers, int nc,
                                                              int dim, const int32_t *__restrict__ assign, float *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nc * dim) return;
  const int c = t / dim, k = t - c * dim;
  const float ck = centers[t];
  float s = 0.f;
  for (int i = 0; i < n; ++i)
    if (assign[i] == c) {
      const float d = features[(size_t)i * dim + k] - ck;
      s = s + d;
    }
  out[t] = s;
}

// one wavefront... no: the float sum has to run in index order, so one thread per candidate descriptor
__global__ void __launch_bounds__(64) vlad_distance_kernel(const float *__restrict__ ref, const float *__restrict__ others, int m, int len,
                                                           double *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const float *o = others + (size_t)j * len;
  float s = 0.f;
  for (int k = 0; k < len; ++k) {
    const float d = ref[k] - o[k];
    const float sq = d * d;
    s = s + sq;
  }
  out[j] = (double)sqrtf(s);
}

// ---- BoW affinity: np.fabs(h - h2).sum() over float64 histograms (pairs_selection.py:690-708, bow.py:34-36) ----
// The value is defined by numpy's summation order (DOUBLE_pairwise_sum + the 8192-element buffering of reductions, restated in
// oracle/words_oracle.c): a binary tree over leaves of <= 128 elements, each leaf summed by eight interleaved accumulators.  The host
// flattens the tree once per histogram length into a leaf table and a postfix program; on the device one wavefront takes a candidate:
// eight lanes per leaf (one accumulator each) sum eight leaves at a time, then lane 0 folds the leaf sums in the program's order.
struct BowPlan {
  const int *leaf_off, *leaf_len;  // n_leaves
  const int *prog;                 // postfix: >= 0 push leaf, -1 add the two top entries, -2 fold into the running sum (end of a chunk)
  int n_leaves, n_prog;
};
__global__ void __launch_bounds__(64) bow_distance_kernel(const double *__restrict__ ref, const double *__restrict__ others, int m, int len,
                                                          BowPlan plan, double *__restrict__ out) {
  extern __shared__ double leaf_sum[];  // n_leaves
  const int c = blockIdx.x, lane = threadIdx.x;
  const double *g = others + (size_t)c * len;
  const int q = lane >> 3, j = lane & 7;
  for (int l0 = 0; l0 < plan.n_leaves; l0 += 8) {
    const int l = l0 + q;
    double r = 0.0;
    int off = 0, n = 0;
    if (l < plan.n_leaves) {
      off = plan.leaf_off[l];
      n = plan.leaf_len[l];
    }
    const bool wide = n >= 8;
    if (wide) {
      r = fabs(ref[off + j] - g[off + j]);
      for (int i = 8; i < n - (n % 8); i += 8) r += fabs(ref[off + i + j] - g[off + i + j]);
    }
    // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) inside each group of eight lanes
    const double s1 = r + __shfl_xor(r, 1);
    const double s2 = s1 + __shfl_xor(s1, 2);
    double res = s2 + __shfl_xor(s2, 4);
    if (j == 0 && l < plan.n_leaves) {
      if (wide) {
        for (int i = n - (n % 8); i < n; i++) res += fabs(ref[off + i] - g[off + i]);
      } else {  // n < 8: sequential from zero
        res = 0.0;
        for (int i = 0; i < n; i++) res += fabs(ref[off + i] - g[off + i]);
      }
      leaf_sum[l] = res;
    }
  }
  __syncthreads();
  if (lane == 0) {
    double stack[40];
    int sp = 0;
    double total = 0.0;
    for (int k = 0; k < plan.n_prog; k++) {
      const int op = plan.prog[k];
      if (op >= 0) {
        stack[sp++] = leaf_sum[op];
      } else if (op == -1) {
        const double b = stack[--sp], a = stack[--sp];
        stack[sp++] = a + b;
      } else {
        total = total + stack[--sp];
      }
    }
    out[c] = total;
  }
}

// ---- k nearest points within a radius: one thread per query, the candidates stream through LDS; insertion into a sorted list of k
//      (distance, index) kept in global scratch rows (k can be the whole set) ----
__global__ void __launch_bounds__(256) knn_points_kernel(const double *__restrict__ cand, int nc, const double *__restrict__ query, int nq, int k,
                                                         double max_d2, double *__restrict__ out_d2, int32_t *__restrict__ out_idx) {
  __shared__ double tile[256 * 3];
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  double qx = 0, qy = 0, qz = 0;
  if (qi < nq) {
    qx = query[3 * (size_t)qi];
    qy = query[3 * (size_t)qi + 1];
    qz = query[3 * (size_t)qi + 2];
  }
  double *rd = out_d2 + (size_t)qi * k;
  int32_t *ri = out_idx + (size_t)qi * k;
  int cnt = 0;
  for (int c0 = 0; c0 < nc; c0 += 256) {
    __syncthreads();
    const int cl = min(256, nc - c0);
    for (int t = threadIdx.x; t < cl * 3; t += 256) tile[t] = cand[(size_t)c0 * 3 + t];
    __syncthreads();
    if (qi >= nq) continue;
    for (int c = 0; c < cl; ++c) {
      const double dx = tile[3 * c] - qx, dy = tile[3 * c + 1] - qy, dz = tile[3 * c + 2] - qz;
      const double d2 = dx * dx + dy * dy + dz * dz;
      if (!(d2 <= max_d2)) continue;
      if (cnt == k && !(d2 < rd[k - 1])) continue;  // equal distances: the earlier index stays
      int pos = cnt < k ? cnt : k - 1;
      while (pos > 0 && rd[pos - 1] > d2) {
        rd[pos] = rd[pos - 1];
        ri[pos] = ri[pos - 1];
        --pos;
      }
      rd[pos] = d2;
      ri[pos] = c0 + c;
      if (cnt < k) ++cnt;
    }
  }
  if (qi < nq)
    for (int t = cnt; t < k; ++t) {
      rd[t] = INFINITY;
      ri[t] = -1;
    }
}

// every candidate within the radius, as a bit mask per query (row of ceil(nc / 32) words): one thread per (query, word)
__global__ void __launch_bounds__(256) radius_points_kernel(const double *__restrict__ cand, int nc, const double *__restrict__ query, int nq,
                                                            double max_d2, uint32_t *__restrict__ mask) {
  const int words = (nc + 31) / 32;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)nq * words) return;
  const int qi = (int)(t / words), wd = (int)(t - (long)qi * words);
  const double qx = query[3 * (size_t)qi], qy = query[3 * (size_t)qi + 1], qz = query[3 * (size_t)qi + 2];
  uint32_t m = 0;
  for (int b = 0; b < 32; ++b) {
    const int c = wd * 32 + b;
    if (c < nc) {
      const double dx = cand[3 * (size_t)c] - qx, dy = cand[3 * (size_t)c + 1] - qy, dz = cand[3 * (size_t)c + 2] - qz;
      const double d2 = dx * dx + dy * dy + dz * dz;
      m |= (d2 <= max_d2) ? (1u << b) : 0u;
    }
  }
  mask[t] = m;
}

template <class T>
int dev_alloc(T **p, size_t n) {
  *p = nullptr;
  return hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)) == hipSuccess ? OSFM_OK : OSFM_E_NOMEM;
}

}  // namespace

extern "C" int osfm_words_store_create(osfm_ctx *ctx, int n_images, const int32_t *counts, int dim, int words_per_feature, const float *desc,
                                       const int32_t *words, osfm_words_store **out) {
  OSFM_REQUIRE(ctx && counts && out && (desc || n_images == 0) && (words || n_images == 0), OSFM_E_INVALID, "osfm_words_store_create: null argument");
  OSFM_REQUIRE(n_images >= 0 && words_per_feature >= 1, OSFM_E_INVALID, "osfm_words_store_create: bad sizes");
  OSFM_REQUIRE(dim == 128, OSFM_E_UNSUPPORTED, "osfm_words_store_create: %d-dimensional descriptors (128 supported)", dim);
  *out = nullptr;
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<osfm_words_store, void (*)(osfm_words_store *)> s(new (std::nothrow) osfm_words_store(), osfm_words_store_destroy);
  OSFM_REQUIRE(s != nullptr, OSFM_E_NOMEM, "out of host memory");
  s->ctx = ctx;
  s->n_images = n_images;
  s->dim = dim;
  s->nw = words_per_feature;
  s->counts.assign(counts, counts + n_images);
  s->row_off.assign(n_images + 1, 0);
  for (int i = 0; i < n_images; ++i) {
    OSFM_REQUIRE(counts[i] >= 0, OSFM_E_INVALID, "image %d has %d features", i, counts[i]);
    s->row_off[i + 1] = s->row_off[i] + counts[i];
    s->max_count = std::max(s->max_count, counts[i]);
  }
  s->total = s->row_off[n_images];
  // the multimap of MatchUsingWords, as a sorted array per image: by primary word, equal words in insertion (= index) order
  std::vector<int32_t> skey((size_t)s->total), sidx((size_t)s->total);
  for (int im = 0; im < n_images; ++im) {
    const int64_t r0 = s->row_off[im];
    const int n = counts[im];
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](int32_t x, int32_t y) { return words[(r0 + x) * words_per_feature] < words[(r0 + y) * words_per_feature]; });
    for (int e = 0; e < n; ++e) {
      sidx[r0 + e] = order[e];
      skey[r0 + e] = words[(r0 + order[e]) * words_per_feature];
    }
  }
  bool ok = dev_alloc(&s->d_desc, (size_t)s->total * dim) == OSFM_OK && dev_alloc(&s->d_words, (size_t)s->total * words_per_feature) == OSFM_OK &&
            dev_alloc(&s->d_skey, (size_t)s->total) == OSFM_OK && dev_alloc(&s->d_sidx, (size_t)s->total) == OSFM_OK &&
            dev_alloc(&s->d_row_off, (size_t)n_images + 1) == OSFM_OK && dev_alloc(&s->d_counts, (size_t)n_images) == OSFM_OK;
  OSFM_REQUIRE(ok, OSFM_E_NOMEM, "osfm_words_store_create: out of device memory (%lld features)", (long long)s->total);
  if (s->total > 0) {
    OSFM_HIP(hipMemcpy(s->d_desc, desc, (size_t)s->total * dim * sizeof(float), hipMemcpyHostToDevice));
    OSFM_HIP(hipMemcpy(s->d_words, words, (size_t)s->total * words_per_feature * sizeof(int32_t), hipMemcpyHostToDevice));
    OSFM_HIP(hipMemcpy(s->d_skey, skey.data(), skey.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    OSFM_HIP(hipMemcpy(s->d_sidx, sidx.data(), sidx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  OSFM_HIP(hipMemcpy(s->d_row_off, s->row_off.data(), s->row_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  if (n_images > 0) OSFM_HIP(hipMemcpy(s->d_counts, s->counts.data(), s->counts.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  *out = s.release();
  return OSFM_OK;
}

extern "C" void osfm_words_store_destroy(osfm_words_store *s) {
  if (!s) return;
  if (s->ctx) (void)hipSetDevice(s->ctx->device);
  (void)hipFree(s->d_desc);
  (void)hipFree(s->d_words);
  (void)hipFree(s->d_skey);
  (void)hipFree(s->d_sidx);
  (void)hipFree(s->d_row_off);
  (void)hipFree(s->d_counts);
  delete s;
}

extern "C" int osfm_match_words_pairs(osfm_ctx *ctx, const osfm_words_store *store, const int32_t *pairs, int64_t n_pairs, float lowes_ratio,
                                      int max_checks, int symmetric, int32_t *counts, int32_t *matches, double *kernel_ms) {
  OSFM_REQUIRE(ctx && store && (pairs || n_pairs == 0) && (counts || n_pairs == 0) && (matches || n_pairs == 0), OSFM_E_INVALID,
               "osfm_match_words_pairs: null argument");
  OSFM_REQUIRE(n_pairs >= 0 && n_pairs < (1 << 16) * 64ll, OSFM_E_INVALID, "osfm_match_words_pairs: %lld pairs", (long long)n_pairs);
  if (kernel_ms) *kernel_ms = 0.0;
  if (n_pairs == 0) return OSFM_OK;
  for (int64_t p = 0; p < 2 * n_pairs; ++p)
    OSFM_REQUIRE(pairs[p] >= 0 && pairs[p] < store->n_images, OSFM_E_INVALID, "pair %lld references image %d", (long long)(p / 2), pairs[p]);
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int mc = std::max(store->max_count, 1);
  const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(n_pairs, (1ll << 28) / mc));  // <= 1 GiB of per-direction results
  int32_t *d_pairs = nullptr, *d_best = nullptr, *d_counts = nullptr, *d_matches = nullptr;
  struct Free {
    int32_t **p[4];
    ~Free() {
      for (auto q : p) (void)hipFree(*q);
    }
  } guard{{&d_pairs, &d_best, &d_counts, &d_matches}};
  OSFM_REQUIRE(dev_alloc(&d_pairs, (size_t)batch * 2) == OSFM_OK && dev_alloc(&d_best, (size_t)batch * 2 * mc) == OSFM_OK &&
                   dev_alloc(&d_counts, (size_t)batch) == OSFM_OK && dev_alloc(&d_matches, (size_t)batch * mc * 2) == OSFM_OK,
               OSFM_E_NOMEM, "osfm_match_words_pairs: out of device memory");
  hipEvent_t e0, e1;
  OSFM_HIP(hipEventCreate(&e0));
  OSFM_HIP(hipEventCreate(&e1));
  double ms_total = 0.0;
  int rc = OSFM_OK;
  for (int64_t p0 = 0; p0 < n_pairs && rc == OSFM_OK; p0 += batch) {
    const int64_t np = std::min(batch, n_pairs - p0);
    WordsArgs a;
    a.desc = store->d_desc;
    a.words = store->d_words;
    a.skey = store->d_skey;
    a.sidx = store->d_sidx;
    a.row_off = store->d_row_off;
    a.counts = store->d_counts;
    a.pairs = d_pairs;
    a.nw = store->nw;
    a.max_count = mc;
    a.ratio = lowes_ratio;
    a.max_checks = max_checks;
    a.best = d_best;
    hipError_t e = hipMemcpyAsync(d_pairs, pairs + 2 * p0, (size_t)np * 2 * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    if (e == hipSuccess) {
      for (int64_t z0 = 0; z0 < np; z0 += 65535) {  // gridDim.z limit
        const int nz = (int)std::min<int64_t>(65535, np - z0);
        WordsArgs b = a;
        b.pairs = d_pairs + 2 * z0;
        b.best = d_best + z0 * 2 * mc;
        hipLaunchKernelGGL(words_match_kernel<128>, dim3((mc + kWT - 1) / kWT, symmetric ? 2 : 1, nz), dim3(kWT), 0, st, b);
      }
      hipLaunchKernelGGL(words_emit_kernel, dim3((unsigned)np), dim3(256), 0, st, d_pairs, store->d_counts, d_best, mc, symmetric, d_counts, d_matches);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipMemcpyAsync(counts + p0, d_counts, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    for (int64_t p = 0; p < np && e == hipSuccess; ++p)  // only the rows that hold matches travel back
      if (counts[p0 + p] > 0)
        e = hipMemcpyAsync(matches + (p0 + p) * (int64_t)mc * 2, d_matches + p * (int64_t)mc * 2, (size_t)counts[p0 + p] * 2 * sizeof(int32_t),
                           hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      osfm_set_error("osfm_match_words_pairs: %s", hipGetErrorString(e));
      rc = OSFM_E_HIP;
    } else {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      ms_total += ms;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (kernel_ms) *kernel_ms = ms_total;
  return rc;
}

extern "C" int osfm_words_store_max_count(const osfm_words_store *s) { return s ? std::max(s->max_count, 1) : 0; }

extern "C" int osfm_vlad_descriptor(osfm_ctx *ctx, const float *features, int n, const float *centers, int n_centers, int dim, float *out) {
  OSFM_REQUIRE(ctx && centers && out && (features || n == 0), OSFM_E_INVALID, "osfm_vlad_descriptor: null argument");
  OSFM_REQUIRE(n >= 0, OSFM_E_INVALID, "osfm_vlad_descriptor: n < 0");
  // compute_vlad_descriptor throws on an empty vocabulary (matching.cc:98-100)
  OSFM_REQUIRE(n_centers > 0 && dim > 0, OSFM_E_INVALID, "Zero VLAD centers or zero length VLAD words.");
  OSFM_REQUIRE((size_t)n_centers * dim * sizeof(float) <= 64 * 1024, OSFM_E_UNSUPPORTED, "osfm_vlad_descriptor: vocabulary of %d x %d floats", n_centers, dim);
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  float *d_f = nullptr, *d_c = nullptr, *d_o = nullptr;
  int32_t *d_a = nullptr;
  struct Free {
    void **p[4];
    ~Free() {
      for (auto q : p) (void)hipFree(*q);
    }
  } guard{{(void **)&d_f, (void **)&d_c, (void **)&d_o, (void **)&d_a}};
  OSFM_REQUIRE(dev_alloc(&d_f, (size_t)n * dim) == OSFM_OK && dev_alloc(&d_c, (size_t)n_centers * dim) == OSFM_OK &&
                   dev_alloc(&d_o, (size_t)n_centers * dim) == OSFM_OK && dev_alloc(&d_a, (size_t)n) == OSFM_OK,
               OSFM_E_NOMEM, "osfm_vlad_descriptor: out of device memory");
  if (n > 0) OSFM_HIP(hipMemcpyAsync(d_f, features, (size_t)n * dim * sizeof(float), hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(d_c, centers, (size_t)n_centers * dim * sizeof(float), hipMemcpyHostToDevice, st));
  if (n > 0)
    hipLaunchKernelGGL(vlad_assign_kernel, dim3((n + 255) / 256), dim3(256), (size_t)n_centers * dim * sizeof(float), st, d_f, n, d_c, n_centers, dim, d_a);
  hipLaunchKernelGGL(vlad_accumulate_kernel, dim3((n_centers * dim + 255) / 256), dim3(256), 0, st, d_f, n, d_c, n_centers, dim, d_a, d_o);
  OSFM_HIP(hipGetLastError());
  OSFM_HIP(hipMemcpyAsync(out, d_o, (size_t)n_centers * dim * sizeof(float), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  return OSFM_OK;
}

extern "C" int osfm_vlad_distances(osfm_ctx *ctx, const float *reference, const float *others, int m, int len, double *out) {
  OSFM_REQUIRE(ctx && reference && (others || m == 0) && (out || m == 0), OSFM_E_INVALID, "osfm_vlad_distances: null argument");
  OSFM_REQUIRE(m >= 0 && len > 0, OSFM_E_INVALID, "osfm_vlad_distances: bad sizes");
  if (m == 0) return OSFM_OK;
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  float *d_r = nullptr, *d_o = nullptr;
  double *d_d = nullptr;
  struct Free {
    void **p[3];
    ~Free() {
      for (auto q : p) (void)hipFree(*q);
    }
  } guard{{(void **)&d_r, (void **)&d_o, (void **)&d_d}};
  OSFM_REQUIRE(dev_alloc(&d_r, (size_t)len) == OSFM_OK && dev_alloc(&d_o, (size_t)m * len) == OSFM_OK && dev_alloc(&d_d, (size_t)m) == OSFM_OK,
               OSFM_E_NOMEM, "osfm_vlad_distances: out of device memory");
  OSFM_HIP(hipMemcpyAsync(d_r, reference, (size_t)len * sizeof(float), hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(d_o, others, (size_t)m * len * sizeof(float), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(vlad_distance_kernel, dim3((m + 63) / 64), dim3(64), 0, st, d_r, d_o, m, len, d_d);
  OSFM_HIP(hipGetLastError());
  OSFM_HIP(hipMemcpyAsync(out, d_d, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  return OSFM_OK;
}

static void bow_plan_rec(int off, int n, std::vector<int> &lo, std::vector<int> &ll, std::vector<int> &prog) {
  if (n <= 128) {
    prog.push_back((int)lo.size());
    lo.push_back(off);
    ll.push_back(n);
    return;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  bow_plan_rec(off, n2, lo, ll, prog);
  bow_plan_rec(off + n2, n - n2, lo, ll, prog);
  prog.push_back(-1);
}

extern "C" int osfm_bow_distances(osfm_ctx *ctx, const double *reference, const double *others, int m, int len, double *out) {
  OSFM_REQUIRE(ctx && reference && (others || m == 0) && (out || m == 0), OSFM_E_INVALID, "osfm_bow_distances: null argument");
  OSFM_REQUIRE(m >= 0 && len > 0, OSFM_E_INVALID, "osfm_bow_distances: bad sizes");
  if (m == 0) return OSFM_OK;
  std::vector<int> lo, ll, prog;
  for (int c = 0; c < len; c += 8192) {  // numpy reduces through an 8192-element buffer: one pairwise tree per chunk, folded in order
    bow_plan_rec(c, std::min(8192, len - c), lo, ll, prog);
    prog.push_back(-2);
  }
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double *d_r = nullptr, *d_o = nullptr, *d_d = nullptr;
  int *d_plan = nullptr;
  struct Free {
    void **p[4];
    ~Free() {
      for (auto q : p) (void)hipFree(*q);
    }
  } guard{{(void **)&d_r, (void **)&d_o, (void **)&d_d, (void **)&d_plan}};
  const size_t nl = lo.size(), np_ = prog.size();
  OSFM_REQUIRE(dev_alloc(&d_r, (size_t)len) == OSFM_OK && dev_alloc(&d_o, (size_t)m * len) == OSFM_OK && dev_alloc(&d_d, (size_t)m) == OSFM_OK &&
                   dev_alloc(&d_plan, 2 * nl + np_) == OSFM_OK,
               OSFM_E_NOMEM, "osfm_bow_distances: out of device memory");
  OSFM_HIP(hipMemcpyAsync(d_r, reference, (size_t)len * sizeof(double), hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(d_o, others, (size_t)m * len * sizeof(double), hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(d_plan, lo.data(), nl * sizeof(int), hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(d_plan + nl, ll.data(), nl * sizeof(int), hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(d_plan + 2 * nl, prog.data(), np_ * sizeof(int), hipMemcpyHostToDevice, st));
  const BowPlan plan{d_plan, d_plan + nl, d_plan + 2 * nl, (int)nl, (int)np_};
  OSFM_REQUIRE(nl * sizeof(double) <= 60 * 1024, OSFM_E_UNSUPPORTED, "osfm_bow_distances: histogram of %d bins", len);
  hipLaunchKernelGGL(bow_distance_kernel, dim3(m), dim3(64), nl * sizeof(double), st, d_r, d_o, m, len, plan, d_d);
  OSFM_HIP(hipGetLastError());
  OSFM_HIP(hipMemcpyAsync(out, d_d, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  return OSFM_OK;
}

extern "C" int osfm_radius_points(osfm_ctx *ctx, const double *candidates, int n_candidates, const double *queries, int n_queries,
                                  double max_distance, uint32_t *out_mask) {
  OSFM_REQUIRE(ctx && (candidates || n_candidates == 0) && (queries || n_queries == 0) && (out_mask || n_queries == 0 || n_candidates == 0),
               OSFM_E_INVALID, "osfm_radius_points: null argument");
  OSFM_REQUIRE(n_candidates >= 0 && n_queries >= 0, OSFM_E_INVALID, "osfm_radius_points: bad sizes");
  if (n_queries == 0 || n_candidates == 0) return OSFM_OK;
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double *d_c = nullptr, *d_q = nullptr;
  uint32_t *d_m = nullptr;
  struct Free {
    void **p[3];
    ~Free() {
      for (auto q : p) (void)hipFree(*q);
    }
  } guard{{(void **)&d_c, (void **)&d_q, (void **)&d_m}};
  const size_t words = (size_t)(n_candidates + 31) / 32, total = words * n_queries;
  OSFM_REQUIRE(dev_alloc(&d_c, (size_t)n_candidates * 3) == OSFM_OK && dev_alloc(&d_q, (size_t)n_queries * 3) == OSFM_OK && dev_alloc(&d_m, total) == OSFM_OK,
               OSFM_E_NOMEM, "osfm_radius_points: out of device memory");
  OSFM_HIP(hipMemcpyAsync(d_c, candidates, (size_t)n_candidates * 3 * sizeof(double), hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(d_q, queries, (size_t)n_queries * 3 * sizeof(double), hipMemcpyHostToDevice, st));
  const double md2 = std::isfinite(max_distance) ? max_distance * max_distance : INFINITY;
  hipLaunchKernelGGL(radius_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_c, n_candidates, d_q, n_queries, md2, d_m);
  OSFM_HIP(hipGetLastError());
  OSFM_HIP(hipMemcpyAsync(out_mask, d_m, total * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  return OSFM_OK;
}

extern "C" int osfm_knn_points(osfm_ctx *ctx, const double *candidates, int n_candidates, const double *queries, int n_queries, int k,
                               double max_distance, double *out_distance, int32_t *out_index) {
  OSFM_REQUIRE(ctx && (candidates || n_candidates == 0) && (queries || n_queries == 0) && (out_distance || n_queries == 0) &&
                   (out_index || n_queries == 0),
               OSFM_E_INVALID, "osfm_knn_points: null argument");
  OSFM_REQUIRE(n_candidates >= 0 && n_queries >= 0 && k >= 1, OSFM_E_INVALID, "osfm_knn_points: bad sizes");
  if (n_queries == 0) return OSFM_OK;
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double *d_c = nullptr, *d_q = nullptr, *d_d = nullptr;
  int32_t *d_i = nullptr;
  struct Free {
    void **p[4];
    ~Free() {
      for (auto q : p) (void)hipFree(*q);
    }
  } guard{{(void **)&d_c, (void **)&d_q, (void **)&d_d, (void **)&d_i}};
  const int nqp = (n_queries + 255) / 256 * 256;
  OSFM_REQUIRE(dev_alloc(&d_c, (size_t)n_candidates * 3) == OSFM_OK && dev_alloc(&d_q, (size_t)n_queries * 3) == OSFM_OK &&
                   dev_alloc(&d_d, (size_t)nqp * k) == OSFM_OK && dev_alloc(&d_i, (size_t)nqp * k) == OSFM_OK,
               OSFM_E_NOMEM, "osfm_knn_points: out of device memory");
  if (n_candidates > 0) OSFM_HIP(hipMemcpyAsync(d_c, candidates, (size_t)n_candidates * 3 * sizeof(double), hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(d_q, queries, (size_t)n_queries * 3 * sizeof(double), hipMemcpyHostToDevice, st));
  const double md2 = std::isfinite(max_distance) ? max_distance * max_distance : INFINITY;
  hipLaunchKernelGGL(knn_points_kernel, dim3(nqp / 256), dim3(256), 0, st, d_c, n_candidates, d_q, n_queries, k, md2, d_d, d_i);
  OSFM_HIP(hipGetLastError());
  OSFM_HIP(hipMemcpyAsync(out_distance, d_d, (size_t)n_queries * k * sizeof(double), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipMemcpyAsync(out_index, d_i, (size_t)n_queries * k * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  for (size_t t = 0; t < (size_t)n_queries * k; ++t) out_distance[t] = std::sqrt(out_distance[t]);  // inf stays inf
  return OSFM_OK;
}
