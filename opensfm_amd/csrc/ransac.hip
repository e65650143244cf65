// ransac.hip -- batched fundamental-matrix RANSAC for gfx950 (MI355X).
//
// Replaces cv2.findFundamentalMat(p1, p2, FM_RANSAC, 0.004, 0.9999) as called by robust_match_fundamental
// (opensfm/matching.py:780-802), plus the two min-match gates of matching.match (matching.py:590-598, 632-634), for every pair of a
// batch.  Results are bit-identical to the sequential algorithm (same RNG stream, same hypothesis order, same adaptive stopping rule);
// the numerics and the per-pair logic are in fransac_core.h, shared with the host emulation that pins them against the CPU oracle.
//
// Organisation (round 3): the first 16 hypotheses of EVERY pair in three launches, pooled across pairs where lanes would idle; at the
// inlier ratios pair preselection produces cv2's adaptive iteration count collapses to ~6 after the first all-inlier sample and 99 %
// of the pairs end inside these 16 (median 6 iterations).
//   fransac_draw_kernel    one wavefront per pair: the gates, then 16 subsets from the cv::RNG stream by the table scheme of
//                          fransac_core.h (lane 0 only steps the generator; dedup, subset boundaries and collinearity tests run
//                          across the lanes).  2.6 KB of LDS: the whole chunk is resident at once.
//   fransac_solve_kernel   one LANE per 7-point problem, 64 problems of 4 pairs side by side in a wave (the 7 x 9 systems lane-minor
//                          in LDS): the solver's ~7 k instructions are issued once per 64 problems instead of once per 8.
//   fransac_decide_kernel  one wavefront per pair: correspondences into LDS, then the sequential decisions with LAZY scoring -- a
//                          model is scored (64 correspondences per step, ballot + popcount) only when the loop reaches it -- and,
//                          where the loop has ended, the final mask and the in-place ordered compaction.  The rest go on a list.
//   fransac_rest_kernel    a persistent grid (two workgroups per CU) pulls pairs from that list: 32, 64, 64, ... hypotheses per
//                          round, 64 lanes solve, four waves score whole models each.
// Round 2 ran one 256-thread workgroup per pair with lane 0 drawing every subset in a loop (indices and points in scratch memory):
// 0.63 ms per pair, 14.4 ms for the 11.6 k overlapping pairs of the neighbour list.
#include "osfm_internal.h"

#include "fransac_core.h"

namespace {
using namespace fransac;

constexpr int kThreads = 256;  // long-run kernel
constexpr int kWaves = 4;
constexpr int kBatch = 64;     // hypotheses per round, long runs
constexpr int kRawLong = 512;  // values of the RNG stream per round, long runs
constexpr int kFirst = 16;     // hypotheses of the first round (draw / solve / decide kernels)
constexpr int kRawFirst = 192;
constexpr int kFirstPts = 768;  // correspondences the decide kernel stages in LDS; pairs with more read them from HBM
constexpr int kPairsLdsPts = 1024;  // correspondences the long-run kernel stages in LDS; pairs with more read them from HBM

// correspondence k as cv2 sees it (CV_32F): from LDS, or -- more matches than the buffer holds, rare -- gathered from HBM
struct PtsAny {
  const Pt4 *lds;
  const uint32_t *gm;       // the pair's packed match list (i | j << 16), or null: correspondence k = (gp1[k], gp2[k])
  const double *gp1, *gp2;  // keypoints of the two images (x, y per feature)
  __device__ __forceinline__ Pt4 operator()(int k) const {
    if (lds) return lds[k];
    int i = k, j = k;
    if (gm) {
      const uint32_t m = gm[k];
      i = (int)(m & 0xFFFFu);
      j = (int)(m >> 16);
    }
    return Pt4{(float)gp1[2 * i], (float)gp1[2 * i + 1], (float)gp2[2 * j], (float)gp2[2 * j + 1]};
  }
};

template <class PTS>
__device__ __forceinline__ int count_inliers(const double *Fm, const PTS &pts, int n, float t, int lane) {
  double F[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) F[i] = Fm[i];
  int good = 0;
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    bool in = false;
    if (i < n) {
      const Pt4 q = pts(i);
      in = epi_error(F, (double)q.x, (double)q.y, (double)q.z, (double)q.w) <= t;
    }
    good += __popcll(__ballot(in));
  }
  return good;
}

// one wavefront = one pair (64-thread workgroup)
struct WaveEx {
  int lane;
  template <class F>
  __device__ __forceinline__ void one(F f) {
    if (lane == 0) f();
    __syncthreads();
  }
  template <class F>
  __device__ __forceinline__ void par(int n, F f) {
    for (int i = lane; i < n; i += 64) f(i);
    __syncthreads();
  }
  template <class PTS>
  __device__ __forceinline__ void score(int nsub, const double (*models)[27], const unsigned char *nmodels, const PTS &pts, int n, float t,
                                        int (*good)[3]) {
    for (int b = 0; b < nsub; ++b)
      for (int k = 0; k < nmodels[b]; ++k) {
        const int g = count_inliers(models[b] + 9 * k, pts, n, t, lane);
        if (lane == 0) good[b][k] = g;
      }
    __syncthreads();
  }
};
// 256 threads = one pair: every wavefront scores whole models
struct BlockEx {
  int tid;
  template <class F>
  __device__ __forceinline__ void one(F f) {
    if (tid == 0) f();
    __syncthreads();
  }
  template <class F>
  __device__ __forceinline__ void par(int n, F f) {
    for (int i = tid; i < n; i += kThreads) f(i);
    __syncthreads();
  }
  template <class PTS>
  __device__ __forceinline__ void score(int nsub, const double (*models)[27], const unsigned char *nmodels, const PTS &pts, int n, float t,
                                        int (*good)[3]) {
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int mi = w; mi < nsub * 3; mi += kWaves) {
      const int b = mi / 3, k = mi - 3 * b;
      if (k >= nmodels[b]) continue;
      const int g = count_inliers(models[b] + 9 * k, pts, n, t, lane);
      if (lane == 0) good[b][k] = g;
    }
    __syncthreads();
  }
};

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
  // hand-over between the kernels
  PairState *states;         // [n_pairs]
  int32_t *list;             // [n_pairs] pairs that continue in the long-run kernel
  int32_t *ctl;              // [0] entries of the list, [1] the long-run kernel's pull cursor
  int32_t *hdr;              // [n_pairs][2]: subsets the draw kernel produced (-1: the pair failed a gate), getSubset-failed flag
  unsigned short *subsets;   // [n_pairs][kFirst][8]
  double *models;            // [n_pairs][kFirst][27]
  unsigned char *nmodels;    // [n_pairs][kFirst]
};

// After the RANSAC loop: the verdict (matching.py:798-800), the final mask with the best F and the ordered in-place compaction of the
// inliers, the second gate (matching.py:632-634).  NW waves of the workgroup take part; misc: NW ints of LDS.
template <int NW, class PTS>
__device__ __forceinline__ void finish_pair(const RansacPairsArgs &a, long p, int n, const PairState &st, const PTS &pts, int tid, int *misc) {
  const int lane = tid & 63, w = tid >> 6;
  if (a.work && tid == 0) atomicAdd(a.work, st.scored * (unsigned long long)n);
  if (a.F_out && tid == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) a.F_out[p * 9 + i] = st.max_good > 0 ? st.best[i] : 0.0;  // static indices: st may live in registers
  }
  if (st.max_good <= 0 || st.best[8] == 0.0) {  // F is None or F[2,2] == 0 -> no matches
    if (tid == 0) a.counts[p] = 0;
    return;
  }
  double F[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) F[i] = st.best[i];
  const double thr = a.thr <= 0 ? 3 : a.thr;
  const float t = (float)(thr * thr);
  int base = 0;
  for (int k0 = 0; k0 < n; k0 += NW * 64) {
    const int k = k0 + tid;
    bool in = false;
    uint32_t mk = 0;  // read before the barrier below: the in-place writes of this round only go to slots <= k
    if (k < n) {
      const Pt4 q = pts(k);
      in = epi_error(F, (double)q.x, (double)q.y, (double)q.z, (double)q.w) <= t;
      mk = a.matches[p * a.cap + k];
    }
    const unsigned long long bal = __ballot(in);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) misc[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) {
      const int cnt = misc[w2];
      woff += (w2 < w) ? cnt : 0;
      total += cnt;
    }
    if (in) a.matches[p * a.cap + base + woff + prefix] = mk;
    base += total;
    __syncthreads();
  }
  if (tid == 0) a.counts[p] = base >= a.min_match ? base : 0;
}

struct DrawShared {
  PairState st;
  DrawOut O;
  unsigned short subset[kFirst][8];
  DrawBuf<kRawFirst, kFirst> D;
};

__global__ void __launch_bounds__(64) fransac_draw_kernel(RansacPairsArgs a) {
  __shared__ DrawShared sh;
  const int lane = threadIdx.x;
  const long p = blockIdx.x;
  const int n = min(a.counts[p], a.cap);
  // gates: matching.py:590-598 (min match) and matching.py:787-788 (< 8)
  if (n < a.min_match || n < 8) {
    if (lane == 0) {
      a.counts[p] = 0;
      a.hdr[2 * p] = -1;
    }
    return;
  }
  if (lane == 0) state_init(sh.st, a.max_iters);
  __syncthreads();
  const int img1 = a.pairs[2 * p], img2 = a.pairs[2 * p + 1];
  const PtsAny pts{nullptr, a.matches + p * a.cap, a.pts + a.tile_off[img1] * 64, a.pts + a.tile_off[img2] * 64};
  WaveEx ex{lane};
  round_draw(ex, sh.st, sh.D, sh.O, sh.subset, pts, n, kFirst);
  const int nsub = sh.O.nsub;
  for (int k = lane; k < nsub * 8; k += 64) a.subsets[p * (kFirst * 8) + k] = sh.subset[k >> 3][k & 7];
  if (lane == 0) {
    a.hdr[2 * p] = nsub;
    a.hdr[2 * p + 1] = sh.O.fail;
    a.states[p] = sh.st;
  }
}

// lane g solves subset (g % kFirst) of pair (g / kFirst)
__global__ void __launch_bounds__(64) fransac_solve_kernel(RansacPairsArgs a) {
  __shared__ double priv[kPrivDoubles * 64];
  __shared__ int ipriv[9 * 64];
  const int lane = threadIdx.x;
  const long g = (long)blockIdx.x * 64 + lane;
  const long p = g / kFirst;
  const int b = (int)(g - p * kFirst);
  const bool active = p < a.n_pairs && b < a.hdr[2 * p];
  if (__ballot(active) == 0ull) return;
  if (!active) return;  // no barrier below: every lane works on its own columns of priv / ipriv
  const int img1 = a.pairs[2 * p], img2 = a.pairs[2 * p + 1];
  const PtsAny pts{nullptr, a.matches + p * a.cap, a.pts + a.tile_off[img1] * 64, a.pts + a.tile_off[img2] * 64};
  unsigned short idx[8];
  {
    const uint4 v = *reinterpret_cast<const uint4 *>(a.subsets + (p * kFirst + b) * 8);
    idx[0] = v.x & 0xFFFF, idx[1] = v.x >> 16, idx[2] = v.y & 0xFFFF, idx[3] = v.y >> 16;
    idx[4] = v.z & 0xFFFF, idx[5] = v.z >> 16, idx[6] = v.w & 0xFFFF, idx[7] = 0;
  }
  double ms1[14], ms2[14];
  subset_points(idx, pts, ms1, ms2);
  a.nmodels[p * kFirst + b] = (unsigned char)run_7point<64>(ms1, ms2, a.models + (p * kFirst + b) * 27, priv + lane, ipriv + lane);
}

struct DecideShared {
  int misc[4];
  Pt4 pts[kFirstPts];
};

__global__ void __launch_bounds__(64) fransac_decide_kernel(RansacPairsArgs a) {
  __shared__ DecideShared sh;
  const int lane = threadIdx.x;
  const long p = blockIdx.x;
  const int nsub = a.hdr[2 * p];
  if (nsub < 0) return;  // failed a gate (the draw kernel cleared its count)
  const int fail = a.hdr[2 * p + 1];
  const int n = min(a.counts[p], a.cap);
  const int img1 = a.pairs[2 * p], img2 = a.pairs[2 * p + 1];
  const double *pts1 = a.pts + a.tile_off[img1] * 64;
  const double *pts2 = a.pts + a.tile_off[img2] * 64;
  const bool in_lds = n <= kFirstPts;
  if (in_lds)
    for (int k = lane; k < n; k += 64) {
      const uint32_t m = a.matches[p * a.cap + k];
      const int i = m & 0xFFFF, j = m >> 16;
      sh.pts[k] = Pt4{(float)pts1[2 * i], (float)pts1[2 * i + 1], (float)pts2[2 * j], (float)pts2[2 * j + 1]};
    }
  __syncthreads();
  double thr = a.thr, conf = a.conf;
  if (thr <= 0) thr = 3;
  if (conf < 2.220446049250313e-16 || conf > 1 - 2.220446049250313e-16) conf = 0.99;
  const float t = (float)(thr * thr);
  const PtsAny pts{in_lds ? sh.pts : nullptr, a.matches + p * a.cap, pts1, pts2};
  PairState st = a.states[p];  // the same value on every lane
  const bool done = decide_lazy(st, n, conf, nsub, fail, a.models + p * (kFirst * 27), a.nmodels + p * kFirst,
                                [&](const double *Fm) { return count_inliers(Fm, pts, n, t, lane); });
  if (!done) {
    if (lane == 0) {
      a.states[p] = st;
      a.list[atomicAdd(a.ctl, 1)] = (int32_t)p;
    }
    return;
  }
  finish_pair<1>(a, p, n, st, pts, lane, sh.misc);
}

// LDS of the long-run kernels: fixed part, then the staged correspondences
struct LongShared {
  PairState st;
  DrawOut O;
  unsigned short subset[kBatch][8];
  double models[kBatch][27];
  int good[kBatch][3];
  unsigned char nmodels[kBatch];
  int misc[8];
  double priv[kPrivDoubles * 64];  // [kPrivDoubles][64]: the 7-point systems (run_7point), and before that the table of the draws
  int ipriv[9 * 64];
};
static_assert(sizeof(DrawBuf<kRawLong, kBatch>) <= sizeof(double) * kPrivDoubles * 64, "the draw table must fit the 7-point scratch");

// The RANSAC loop of one pair from the state in sh.st to its end, 256 threads.  first_lmax: hypotheses of the first round here.
template <class PTS>
__device__ void long_run(LongShared &sh, const PTS &pts, int n, double thr, double conf, int tid, int first_lmax) {
  if (thr <= 0) thr = 3;
  if (conf < 2.220446049250313e-16 || conf > 1 - 2.220446049250313e-16) conf = 0.99;
  const float t = (float)(thr * thr);
  BlockEx ex{tid};
  auto &D = *reinterpret_cast<DrawBuf<kRawLong, kBatch> *>(sh.priv);
  // Speculation depth: the batch a round solves and scores ahead of the sequential decision.  cv2 stops after niters iterations and
  // niters collapses as soon as a good model is found, so the rounds grow -- 16, 32, then 64 hypotheses -- and only pairs that really
  // need hundreds of iterations pay for full batches.  The subsets come in stream order whatever the batch boundaries are, so the
  // result does not depend on this schedule.
  int lmax = first_lmax;
  for (;;) {
    if (fransac_round<kRawLong, kBatch, 64>(ex, sh.st, D, sh.O, sh.subset, sh.models, sh.nmodels, sh.good, sh.priv, sh.ipriv, pts, n, t, conf, lmax))
      break;
    lmax = lmax * 2 < kBatch ? lmax * 2 : kBatch;
  }
}

__global__ void __launch_bounds__(kThreads) fransac_rest_kernel(RansacPairsArgs a, int lds_pts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LongShared &sh = *reinterpret_cast<LongShared *>(smem);
  Pt4 *ptsbuf = reinterpret_cast<Pt4 *>(smem + sizeof(LongShared));
  const int tid = threadIdx.x;
  for (;;) {
    __syncthreads();  // the previous pair is finished with sh
    if (tid == 0) sh.misc[7] = atomicAdd(a.ctl + 1, 1);
    __syncthreads();
    const int q = sh.misc[7];
    if (q >= a.ctl[0]) return;  // the list is complete: the first kernel ran before this one on the stream
    const long p = a.list[q];
    const int n = min(a.counts[p], a.cap);
    if (tid == 0) sh.st = a.states[p];
    const int img1 = a.pairs[2 * p], img2 = a.pairs[2 * p + 1];
    const double *pts1 = a.pts + a.tile_off[img1] * 64;
    const double *pts2 = a.pts + a.tile_off[img2] * 64;
    const bool in_lds = n <= lds_pts;  // uniform
    if (in_lds)
      for (int k = tid; k < n; k += kThreads) {
        const uint32_t m = a.matches[p * a.cap + k];
        const int i = m & 0xFFFF, j = m >> 16;
        ptsbuf[k] = Pt4{(float)pts1[2 * i], (float)pts1[2 * i + 1], (float)pts2[2 * j], (float)pts2[2 * j + 1]};
      }
    __syncthreads();
    const PtsAny pts{in_lds ? ptsbuf : nullptr, a.matches + p * a.cap, pts1, pts2};
    long_run(sh, pts, n, a.thr, a.conf, tid, 32);
    finish_pair<kWaves>(a, p, n, sh.st, pts, tid, sh.misc);
  }
}

__global__ void __launch_bounds__(kThreads) ransac_single_kernel(const double *p1, const double *p2, int n, double thr,
                                                                   double conf, int max_iters, double *F_out,
                                                                   uint8_t *mask, int32_t *info, int in_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LongShared &sh = *reinterpret_cast<LongShared *>(smem);
  Pt4 *ptsbuf = reinterpret_cast<Pt4 *>(smem + sizeof(LongShared));
  const int tid = threadIdx.x;
  if (tid == 0) state_init(sh.st, max_iters);
  if (in_lds)
    for (int k = tid; k < n; k += kThreads) ptsbuf[k] = Pt4{(float)p1[2 * k], (float)p1[2 * k + 1], (float)p2[2 * k], (float)p2[2 * k + 1]};
  __syncthreads();
  const PtsAny pts{in_lds ? ptsbuf : nullptr, nullptr, p1, p2};
  long_run(sh, pts, n, thr, conf, tid, 8);
  const int max_good = sh.st.max_good;
  if (tid == 0) {
    info[0] = max_good > 0 ? 1 : 0;
    info[1] = sh.st.iters;
    info[2] = max_good;
  }
  if (tid < 9) F_out[tid] = max_good > 0 ? sh.st.best[tid] : 0.0;
  double F[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) F[i] = sh.st.best[i];
  const double thr2 = thr <= 0 ? 3 : thr;
  const float t = (float)(thr2 * thr2);
  for (int k = tid; k < n; k += kThreads) {
    const Pt4 q = pts(k);
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
          found = !have_collinear7(ms1) && !have_collinear7(ms2);
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
        nm = run_7point<64>(ms1, ms2, models[tid], priv + tid, ipriv + tid);
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

static constexpr size_t kRansacFixedLds = sizeof(LongShared);
// correspondences the single-problem kernel stages in LDS (it owns a CU); the batched long-run kernel keeps at most kPairsLdsPts (two
// workgroups per CU) and takes pairs with more matches from HBM
int osfm_ransac_lds_points() { return (int)((160 * 1024 - kRansacFixedLds - 64) / 16) & ~3; }
int osfm_ransac_pairs_lds_points() { return kPairsLdsPts; }

static int ensure_ransac_attributes(int device) {
  static OsfmPerDeviceOnce once;
  return once.run(device, []() -> int {
    OSFM_HIP(hipFuncSetAttribute((const void *)fransac_rest_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OSFM_HIP(hipFuncSetAttribute((const void *)ransac_single_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return OSFM_OK;
  });
}

int osfm_launch_ransac_pairs(osfm_ctx *ctx, const osfm_store *store, const int32_t *d_pairs, int64_t n_pairs, int cap,
                             int min_match, double thr, double conf, int max_iters, int32_t *d_counts,
                             uint32_t *d_matches, double *d_F_or_null, hipStream_t stream, unsigned long long *d_work_or_null) {
  if (n_pairs == 0) return OSFM_OK;
  // hand-over buffers between the kernels, kept by the context (grown on demand; one robust stage runs at a time per context)
  const size_t per_pair = sizeof(PairState) + sizeof(int32_t) + 2 * sizeof(int32_t) + (size_t)kFirst * (8 * sizeof(unsigned short) + 27 * sizeof(double) + 1);
  const size_t need = (size_t)n_pairs * per_pair + 256;
  if (need > ctx->fr_scratch_bytes) {
    if (ctx->d_fr_scratch) (void)hipFree(ctx->d_fr_scratch);
    ctx->d_fr_scratch = nullptr;
    ctx->fr_scratch_bytes = 0;
    OSFM_REQUIRE(osfm_malloc_retry(ctx, &ctx->d_fr_scratch, need + need / 8) == hipSuccess, OSFM_E_NOMEM, "hipMalloc failed for the RANSAC hand-over buffers");
    ctx->fr_scratch_bytes = need + need / 8;
  }
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
  a.ctl = (int32_t *)ctx->d_fr_scratch;
  a.models = (double *)((char *)ctx->d_fr_scratch + 64);  // 8-byte members first, 16-byte aligned subsets, then the rest
  a.states = (PairState *)(a.models + (size_t)n_pairs * kFirst * 27);
  a.subsets = (unsigned short *)(((uintptr_t)(a.states + n_pairs) + 15) & ~(uintptr_t)15);
  a.hdr = (int32_t *)(a.subsets + (size_t)n_pairs * kFirst * 8);
  a.list = a.hdr + 2 * n_pairs;
  a.nmodels = (unsigned char *)(a.list + n_pairs);
  const int capr = (cap + 3) & ~3;
  const int lds_pts = std::min(capr, kPairsLdsPts);
  const size_t lds = kRansacFixedLds + (size_t)lds_pts * 16 + 64;
  {
    const int rc = ensure_ransac_attributes(ctx->device);
    if (rc != OSFM_OK) return rc;
  }
  OSFM_HIP(hipMemsetAsync(a.ctl, 0, 64, stream));
  hipLaunchKernelGGL(fransac_draw_kernel, dim3((unsigned)n_pairs), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(fransac_solve_kernel, dim3((unsigned)((n_pairs * kFirst + 63) / 64)), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(fransac_decide_kernel, dim3((unsigned)n_pairs), dim3(64), 0, stream, a);
  OSFM_HIP(hipGetLastError());
  const unsigned grid = (unsigned)std::min<int64_t>(n_pairs, 2 * (int64_t)std::max(1, ctx->num_cus));
  hipLaunchKernelGGL(fransac_rest_kernel, dim3(grid), dim3(kThreads), lds, stream, a, lds_pts);
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
