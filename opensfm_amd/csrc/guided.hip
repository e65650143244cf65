// guided.hip -- masked / guided brute-force matching on gfx950 (second half of row M-a9).
//
// reference: match_brute_force[_symmetric](f1, f2, config, maskij) (opensfm/matching.py:723-777) and the epipolar mask of
// _match_descriptors_guided_impl (matching.py:260-337, 847-868; geometry/src/triangulation.cc:195-219).
// First-correct version: one wavefront per query descriptor (guided_wave.h); the mask is either an explicit n1 x n2 byte
// array or evaluated on the fly from per-feature epipolar vectors (never materialised).
// STATUS: numerics pinned bit for bit by the host emulation (tests/test_guided_host.py); first MI355X run at the end of round 1
// (profiles/r01_guided_bringup.txt): identical to the oracle; 2000 x 2000 guided pair in 0.7 ms including the copies.
#include <math.h>

#include <vector>

#include "guided_wave.h"
#include "osfm_internal.h"

using namespace osfm_gm;

namespace {

struct GpuWave {
  int lane;
  template <class F>
  __device__ void single(F f) {
    __syncthreads();
    if (lane == 0) f();
    __syncthreads();
  }
  template <class F>
  __device__ void parallel_for(int n, F f) {
    __syncthreads();
    for (int i = lane; i < n; i += kLanes) f(i);
    __syncthreads();
  }
};

__global__ __launch_bounds__(256) void guided_prep_kernel(const float *__restrict__ b1, int n1, const float *__restrict__ b2, int n2,
                                                          const double *__restrict__ Rt /* R (9), t^ (3) */, double *__restrict__ first6,
                                                          double *__restrict__ second6) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n1) epipolar_precompute(0, b1 + 3 * (size_t)i, Rt, Rt + 9, first6 + 6 * (size_t)i);
  if (i < n2) epipolar_precompute(1, b2 + 3 * (size_t)i, Rt, Rt + 9, second6 + 6 * (size_t)i);
}

// block q < n1: query = feature q of image 1 against image 2; block q >= n1: feature q - n1 of image 2 against image 1
__global__ __launch_bounds__(kLanes) void guided_match_kernel(const uint8_t *__restrict__ d1, const int *__restrict__ norm1, int n1,
                                                              const uint8_t *__restrict__ d2, const int *__restrict__ norm2, int n2,
                                                              Allowed allowed, double ratio, int *__restrict__ good12,
                                                              int *__restrict__ good21) {
  __shared__ GuidedShared sh;
  __shared__ __attribute__((aligned(16))) uint8_t q[kDim];
  GpuWave w{(int)threadIdx.x};
  const int b = (int)blockIdx.x;
  const bool fwd = b < n1;
  const int qi = fwd ? b : b - n1;
  const uint8_t *src = (fwd ? d1 : d2) + (size_t)qi * kDim;
  for (int k = w.lane; k < kDim; k += kLanes) q[k] = src[k];
  __syncthreads();
  const int r = match_query_wave(w, sh, q, fwd ? norm1[qi] : norm2[qi], qi, fwd ? d2 : d1, fwd ? norm2 : norm1, fwd ? n2 : n1, allowed,
                                 fwd ? 1 : 0, ratio);
  if (w.lane == 0) (fwd ? good12 : good21)[qi] = r;
}

// ordered compaction of the (mutual) matches: pairs sorted by (i, j); one wavefront
__global__ __launch_bounds__(kLanes) void guided_mutual_kernel(const int *__restrict__ good12, const int *__restrict__ good21, int n1,
                                                               int symmetric, int cap, int *__restrict__ out_pairs, int *__restrict__ out_n) {
  const int lane = (int)threadIdx.x;
  int c = 0;
  for (int base = 0; base < n1; base += kLanes) {
    const int i = base + lane;
    int j = -1;
    if (i < n1) {
      j = good12[i];
      if (j >= 0 && symmetric && good21[j] != i) j = -1;
    }
    const unsigned long long m = __ballot(j >= 0);
    const int pos = c + __popcll(m & ((1ull << lane) - 1ull));
    if (j >= 0 && pos < cap) {
      out_pairs[2 * pos] = i;
      out_pairs[2 * pos + 1] = j;
    }
    c += __popcll(m);
  }
  if (lane == 0) *out_n = c;
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
  template <class T>
  T *as() const {
    return (T *)p;
  }
};

// float32 integer-valued descriptors in [0, 255] -> u8 + squared norms (host plumbing; the GPU path rejects anything else, as
// osfm_store_upload_f32 does)
bool to_u8(const float *f, int n, std::vector<uint8_t> &u, std::vector<int> &norm) {
  u.resize((size_t)n * kDim);
  norm.resize((size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    int s = 0;
    for (int k = 0; k < kDim; k++) {
      const float v = f[(size_t)i * kDim + k];
      if (!(v >= 0.0f && v <= 255.0f) || v != (float)(int)v) return false;
      u[(size_t)i * kDim + k] = (uint8_t)(int)v;
      s += (int)v * (int)v;
    }
    norm[i] = s;
  }
  return true;
}

}  // namespace

extern "C" int osfm_match_guided(osfm_ctx *ctx, const float *f1, int n1, const float *f2, int n2, int dim, const uint8_t *mask,
                                 const float *b1, const float *b2, const double *R, const double *t, double threshold, double ratio,
                                 int symmetric, int32_t *out_pairs, int cap, int *out_n) {
  OSFM_REQUIRE(ctx && out_n && (out_pairs || cap == 0), OSFM_E_INVALID, "osfm_match_guided: null argument");
  OSFM_REQUIRE(dim == kDim, OSFM_E_UNSUPPORTED, "descriptor dim %d (only 128 is implemented)", dim);
  OSFM_REQUIRE(n1 >= 0 && n2 >= 0 && (f1 || n1 == 0) && (f2 || n2 == 0) && cap >= 0, OSFM_E_INVALID, "osfm_match_guided: bad descriptor arrays");
  OSFM_REQUIRE(mask || (b1 && b2 && R && t) || n1 == 0 || n2 == 0, OSFM_E_INVALID,
               "osfm_match_guided: give either maskij or (bearings, R, t, threshold)");
  *out_n = 0;
  if (n1 < 1 || n2 < 1) return OSFM_OK;
  std::vector<uint8_t> u1, u2;
  std::vector<int> nr1, nr2;
  OSFM_REQUIRE(to_u8(f1, n1, u1, nr1) && to_u8(f2, n2, u2, nr2), OSFM_E_UNSUPPORTED,
               "osfm_match_guided: descriptors must be integer-valued in [0, 255] (cv2's float result depends on its SIMD build)");
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  DevBuf d_u1, d_u2, d_n1, d_n2, d_mask, d_b1, d_b2, d_rt, d_f6, d_s6, d_g12, d_g21, d_out, d_cnt;
  OSFM_HIP(d_u1.alloc(u1.size()));
  OSFM_HIP(d_u2.alloc(u2.size()));
  OSFM_HIP(d_n1.alloc((size_t)n1 * 4));
  OSFM_HIP(d_n2.alloc((size_t)n2 * 4));
  OSFM_HIP(d_g12.alloc((size_t)n1 * 4));
  OSFM_HIP(d_g21.alloc((size_t)n2 * 4));
  OSFM_HIP(d_out.alloc((size_t)cap * 8));
  OSFM_HIP(d_cnt.alloc(4));
  OSFM_HIP(hipMemcpyAsync(d_u1.p, u1.data(), u1.size(), hipMemcpyHostToDevice, ctx->stream));
  OSFM_HIP(hipMemcpyAsync(d_u2.p, u2.data(), u2.size(), hipMemcpyHostToDevice, ctx->stream));
  OSFM_HIP(hipMemcpyAsync(d_n1.p, nr1.data(), (size_t)n1 * 4, hipMemcpyHostToDevice, ctx->stream));
  OSFM_HIP(hipMemcpyAsync(d_n2.p, nr2.data(), (size_t)n2 * 4, hipMemcpyHostToDevice, ctx->stream));
  Allowed allowed{nullptr, nullptr, nullptr, threshold, n2};
  double rt[12];
  if (mask) {
    OSFM_HIP(d_mask.alloc((size_t)n1 * n2));
    OSFM_HIP(hipMemcpyAsync(d_mask.p, mask, (size_t)n1 * n2, hipMemcpyHostToDevice, ctx->stream));
    allowed.mask = d_mask.as<uint8_t>();
  } else {
    for (int i = 0; i < 9; i++) rt[i] = R[i];
    double tn[3] = {t[0], t[1], t[2]};
    normalized3(tn);
    for (int i = 0; i < 3; i++) rt[9 + i] = tn[i];
    OSFM_HIP(d_b1.alloc((size_t)n1 * 12));
    OSFM_HIP(d_b2.alloc((size_t)n2 * 12));
    OSFM_HIP(d_rt.alloc(sizeof(rt)));
    OSFM_HIP(d_f6.alloc((size_t)n1 * 48));
    OSFM_HIP(d_s6.alloc((size_t)n2 * 48));
    OSFM_HIP(hipMemcpyAsync(d_b1.p, b1, (size_t)n1 * 12, hipMemcpyHostToDevice, ctx->stream));
    OSFM_HIP(hipMemcpyAsync(d_b2.p, b2, (size_t)n2 * 12, hipMemcpyHostToDevice, ctx->stream));
    OSFM_HIP(hipMemcpyAsync(d_rt.p, rt, sizeof(rt), hipMemcpyHostToDevice, ctx->stream));
    const int nmax = n1 > n2 ? n1 : n2;
    hipLaunchKernelGGL(guided_prep_kernel, dim3((nmax + 255) / 256), dim3(256), 0, ctx->stream, d_b1.as<float>(), n1, d_b2.as<float>(), n2,
                       d_rt.as<double>(), d_f6.as<double>(), d_s6.as<double>());
    OSFM_HIP(hipGetLastError());
    allowed.first6 = d_f6.as<double>();
    allowed.second6 = d_s6.as<double>();
  }
  hipLaunchKernelGGL(guided_match_kernel, dim3(n1 + n2), dim3(kLanes), 0, ctx->stream, d_u1.as<uint8_t>(), d_n1.as<int>(), n1,
                     d_u2.as<uint8_t>(), d_n2.as<int>(), n2, allowed, ratio, d_g12.as<int>(), d_g21.as<int>());
  OSFM_HIP(hipGetLastError());
  hipLaunchKernelGGL(guided_mutual_kernel, dim3(1), dim3(kLanes), 0, ctx->stream, d_g12.as<int>(), d_g21.as<int>(), n1, symmetric ? 1 : 0, cap,
                     d_out.as<int>(), d_cnt.as<int>());
  OSFM_HIP(hipGetLastError());
  int cnt = 0;
  OSFM_HIP(hipMemcpyAsync(&cnt, d_cnt.p, 4, hipMemcpyDeviceToHost, ctx->stream));
  OSFM_HIP(hipStreamSynchronize(ctx->stream));
  *out_n = cnt;
  const int nw = cnt < cap ? cnt : cap;
  if (nw > 0) OSFM_HIP(hipMemcpy(out_pairs, d_out.p, (size_t)nw * 8, hipMemcpyDeviceToHost));
  return OSFM_OK;
}

// =================================================================================================================================
// Batched guided matching over the resident store (osfm_match_pairs_guided): every pair of a chunk in three launches.
//
// The epipolar mask leaves a query a band of candidates (a percent or two of the other image at the default 0.006 rad), so the work
// is the PREDICATE on all n1 x n2 combinations (a dozen fp64 operations each) and a handful of descriptor distances per query -- not a
// dense distance matrix; the matrix cores have nothing to do here.
//   guided_pairs_prep_kernel   per pair and feature: [x, e1] / [w, e2] (epipolar_precompute, as the leaf), 48 B per feature
//   guided_pairs_match_kernel  one wavefront per 64 queries of one direction, a lane owns a query.  Phase 1: the targets' vectors are
//                              wave-uniform (scalar loads), the lane keeps one bit per target in LDS -- the mask is never in HBM.  Phase 2:
//                              the lane walks the set bits of its row in ascending order: exact integer distance from the store's int8
//                              tiles (the query's descriptor in registers), sqrtf, cv2's K = 2 insertion in cv2's own order, Lowe's test.
//   guided_pairs_emit_kernel   mutual check + ordered compaction into the chunk's match buffer (the robust stage follows as usual).
// The predicate is the reference's  pi/2 - acos((a + b) / 2) < threshold  (matching.py:847-868, triangulation.cc:195-219) rewritten as
// (a + b) / 2 < c*, with c* found on the host by bisection over the doubles with the same libm acos the oracle calls: the two tests
// agree wherever acos is monotone, and (a + b) / 2 is computed by the same operations in the same order.
// =================================================================================================================================
namespace {

typedef int v4i_g __attribute__((ext_vector_type(4)));

// a'.b' of two stored descriptors (int8 tile layout of the store, osfm_internal.h)
__device__ __forceinline__ int tile_dot(const int8_t *tilesA, int rowA, const int8_t *tilesB, int rowB) {
  const int8_t *pa = tilesA + (long)(rowA >> 5) * OSFM_TILE_BYTES + (rowA & 31) * 16;
  const int8_t *pb = tilesB + (long)(rowB >> 5) * OSFM_TILE_BYTES + (rowB & 31) * 16;
  v4i_g av[8], bv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    av[q] = *(const v4i_g *)(pa + q * 512);
    bv[q] = *(const v4i_g *)(pb + q * 512);
  }
  int s0 = 0, s1 = 0;
#pragma unroll
  for (int q = 0; q < 8; q += 2)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s0 = __builtin_amdgcn_sdot4(av[q][e], bv[q][e], s0, false);
      s1 = __builtin_amdgcn_sdot4(av[q + 1][e], bv[q + 1][e], s1, false);
    }
  return s0 + s1;
}

struct GuidedPairsArgs {
  const int8_t *tiles;
  const int32_t *norms;
  const int64_t *tile_off;
  const int32_t *counts;
  const float *bearings;  // store rows (tile * 32 + row) x 3, float32 as the reference's bearings
  const float *descf;     // float store: rows (tile * 32 + row) x 128 (null for an integer store)
  const float *seg;       // segmentation column (129th descriptor dimension) per row, or null
  const int32_t *pairs;
  const double *poses;  // n_pairs x 12: R (row-major) and t of the relative pose
  long n_pairs;
  int capr;             // rows per image side in the scratch arrays
  double cstar, ratio;
  int symmetric, cap;
  double *six;          // [n_pairs][2][capr][6]
  int32_t *good;        // [n_pairs][2][capr]
  int32_t *out_counts;
  uint32_t *out_matches;
  int32_t *out_flags;
  int wpad;             // LDS words per query (odd)
};

__global__ void __launch_bounds__(256) guided_pairs_prep_kernel(GuidedPairsArgs a) {
  const long p = blockIdx.z;
  const int side = blockIdx.y;
  const int img = a.pairs[2 * p + side];
  const int n = a.counts[img];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double *Rt = a.poses + 12 * p;
  double tn[3] = {Rt[9], Rt[10], Rt[11]};
  normalized3(tn);
  epipolar_precompute(side, a.bearings + (a.tile_off[img] * 32 + i) * 3, Rt, tn, a.six + ((p * 2 + side) * a.capr + i) * 6);
}

__device__ __forceinline__ Top2 top2_shfl_xor(const Top2 &t, int m) {
  Top2 r;
  r.d0 = __shfl_xor(t.d0, m);
  r.d1 = __shfl_xor(t.d1, m);
  r.j0 = __shfl_xor(t.j0, m);
  r.n = __shfl_xor(t.n, m);
  return r;
}

constexpr int kGuidedWin = 512;                   // targets per window
constexpr int kGuidedWpad = kGuidedWin / 64 + 1;  // 64-bit words per query row in LDS (odd: conflict-free per-lane rows)
template <int DIR, bool FLT>  // DIR 0: the queries are the features of the pair's first image, 1: of its second image (the transposed mask)
                              // FLT: float store -- candidate distances in the oracle's float order (l2sqr_rows_f32) instead of exact int8
__device__ __forceinline__ void guided_pairs_match_body(const GuidedPairsArgs &a, unsigned long long *bits) {
  const long p = blockIdx.z;
  const int lane = threadIdx.x;
  const int imgQ = a.pairs[2 * p + DIR], imgT = a.pairs[2 * p + 1 - DIR];
  const int nQ = a.counts[imgQ], nT = a.counts[imgT];
  const int q0 = blockIdx.x * 64;
  if (q0 >= nQ) return;
  const int q = q0 + lane;
  const bool valid = q < nQ;
  double Q[6];
  {
    const double *src = a.six + ((p * 2 + DIR) * a.capr + (valid ? q : q0)) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) Q[k] = src[k];
  }
  const double *T6 = a.six + ((p * 2 + 1 - DIR) * (long)a.capr) * 6;
  const double cstar = a.cstar;
  const int8_t *tilesQ = a.tiles + a.tile_off[imgQ] * OSFM_TILE_BYTES;
  const int8_t *tilesT = a.tiles + a.tile_off[imgT] * OSFM_TILE_BYTES;
  const int32_t *normT = a.norms + a.tile_off[imgT] * 32;
  const int qs = valid ? q : q0;
  const float *descQ = FLT ? a.descf + a.tile_off[imgQ] * (long)(32 * 128) : nullptr;
  const float *descT = FLT ? a.descf + a.tile_off[imgT] * (long)(32 * 128) : nullptr;
  const int nqn = (a.norms + a.tile_off[imgQ] * 32)[qs];
  const float *segT = a.seg ? a.seg + a.tile_off[imgT] * 32 : nullptr;
  const float segq = a.seg ? (a.seg + a.tile_off[imgQ] * 32)[qs] : 0.f;
  v4i_g av[8];  // the query's descriptor stays in registers
  {
    const int8_t *pa = tilesQ + (long)(qs >> 5) * OSFM_TILE_BYTES + (qs & 31) * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) av[k] = *(const v4i_g *)(pa + k * 512);
  }
  unsigned long long *myrow_w = bits + (long)lane * kGuidedWpad;
  const volatile unsigned long long *myrow = myrow_w;  // written and read by this lane only, in program order
  Top2 t = top2_empty();
  // The targets go by in windows of kGuidedWin: a window's mask bits (64 B per query, 4.6 KB of LDS per wavefront instead of one bit per
  // target of the whole image) keep the occupancy at the register limit, which is what hides the scalar-load latency of phase 1.
  for (int t0 = 0; t0 < nT; t0 += kGuidedWin) {
    const int nW = min(kGuidedWin, nT - t0);
    // ---- phase 1: this lane's mask bits for the window; the targets' vectors come through the scalar cache, kUnroll per round trip;
    //      rows beyond nT exist in the scratch (capr is a multiple of 64) and their bits are cleared afterwards ----
    constexpr int kUnroll = 4;
    unsigned lo_half = 0;
    for (int jh = 0; jh < kGuidedWin / 32; ++jh) {
      unsigned half = 0;
      const int jn = min(32, nW - jh * 32);
#pragma unroll 1
      for (int b0 = 0; b0 < 32; b0 += kUnroll) {
        if (b0 >= jn) break;
        double sv[kUnroll][6];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
          for (int k = 0; k < 6; ++k) sv[u][k] = T6[(long)(t0 + jh * 32 + b0 + u) * 6 + k];  // wave-uniform: scalar loads
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
          const double *s = sv[u];
          double ea, eb;
          if (DIR == 0) {  // first6 = Q, second6 = s
            ea = fabs(Q[3] * s[0] + Q[4] * s[1] + Q[5] * s[2]);
            eb = fabs(Q[0] * s[3] + Q[1] * s[4] + Q[2] * s[5]);
          } else {  // first6 = s, second6 = Q
            ea = fabs(s[3] * Q[0] + s[4] * Q[1] + s[5] * Q[2]);
            eb = fabs(s[0] * Q[3] + s[1] * Q[4] + s[2] * Q[5]);
          }
          const double c = (ea + eb) / 2.0;
          half |= (c < cstar) ? (1u << (b0 + u)) : 0u;
        }
      }
      if (jn < 32) half &= jn > 0 ? ((1u << jn) - 1u) : 0u;
      if (!valid) half = 0u;
      if (jh & 1)
        myrow_w[jh >> 1] = (unsigned long long)lo_half | ((unsigned long long)half << 32);  // one type for the row: no aliasing games
      else
        lo_half = half;
    }
    // ---- phase 2: the lane walks the set bits of ITS row in ascending target order -- cv2's own insertion order, so no merge:
    //      exact integer distance from the store's int8 tiles, sqrtf, K = 2 insertion ----
    int w = 0;
    unsigned long long word = myrow[0];
    for (;;) {
      while (word == 0ull && w + 1 < kGuidedWin / 64) word = myrow[++w];
      const bool has = word != 0ull;
      if (!__any(has)) break;
      if (has) {
        const int b = __builtin_ctzll(word);
        word &= word - 1;
        const int j = t0 + w * 64 + b;
        if (FLT) {
          top2_insert(t, sqrtf(l2sqr_rows_f32(descQ + (long)qs * 128, descT + (long)j * 128)), j);
          continue;
        }
        const int8_t *pb = tilesT + (long)(j >> 5) * OSFM_TILE_BYTES + (j & 31) * 16;
        v4i_g bv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) bv[k] = *(const v4i_g *)(pb + k * 512);
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < 8; k += 2)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 = __builtin_amdgcn_sdot4(av[k][e], bv[k][e], s0, false);
            s1 = __builtin_amdgcn_sdot4(av[k + 1][e], bv[k + 1][e], s1, false);
          }
        const int d2 = nqn + normT[j] - 2 * (s0 + s1);
        float d2f = (float)d2;
        if (segT) {  // matching_use_segmentation: the 129th dimension, added as cv2's scalar tail does (match.hip exact_direction)
          const float ts = segq - segT[j];
          const float tt = ts * ts;
          d2f = d2f + tt;
        }
        top2_insert(t, sqrtf(d2f), j);
      }
    }
  }
  if (valid) a.good[(p * 2 + DIR) * a.capr + q] = (t.n >= 2 && (double)t.d0 < a.ratio * (double)t.d1) ? t.j0 : -1;
}

template <bool FLT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) guided_pairs_match_kernel(GuidedPairsArgs a) {
  __shared__ unsigned long long bits[64 * kGuidedWpad];  // [64 queries][kGuidedWpad words]: the mask bits of the current target window
  if (blockIdx.y == 0)
    guided_pairs_match_body<0, FLT>(a, bits);
  else if (a.symmetric)
    guided_pairs_match_body<1, FLT>(a, bits);
}

// pairs (i, j) sorted by i: good12[i] == j and (symmetric) good21[j] == i; packed like the fused matcher's output
__global__ void __launch_bounds__(256) guided_pairs_emit_kernel(GuidedPairsArgs a) {
  __shared__ int misc[4];
  const long p = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n1 = a.counts[a.pairs[2 * p]], n2 = a.counts[a.pairs[2 * p + 1]];
  if (tid == 0 && a.out_flags) a.out_flags[p] = 0;
  if (n1 < 2 || n2 < 2) {  // matching.py:291-300
    if (tid == 0) a.out_counts[p] = 0;
    return;
  }
  const int32_t *g12 = a.good + (p * 2) * a.capr, *g21 = a.good + (p * 2 + 1) * a.capr;
  int base = 0;
  for (int i0 = 0; i0 < n1; i0 += 256) {
    const int i = i0 + tid;
    int j = -1;
    if (i < n1) {
      j = g12[i];
      if (j >= 0 && a.symmetric && g21[j] != i) j = -1;
    }
    const unsigned long long bal = __ballot(j >= 0);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) misc[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const int cnt = misc[w2];
      woff += (w2 < w) ? cnt : 0;
      total += cnt;
    }
    if (j >= 0) {
      const int k = base + woff + prefix;
      if (k < a.cap) a.out_matches[p * a.cap + k] = (uint32_t)i | ((uint32_t)j << 16);
    }
    base += total;
    __syncthreads();
  }
  if (tid == 0) a.out_counts[p] = base;
}

}  // namespace

// c* = the smallest double whose epipolar angle pi/2 - acos(c) is NOT below the threshold (bisection over the ordered doubles in [0, 1+])
double osfm_guided_cos_threshold(double threshold) {
  auto below = [&](double c) { return M_PI / 2.0 - acos(c) < threshold; };  // NaN (c > 1) compares false, as in the reference
  if (!below(0.0)) return 0.0;   // nothing is allowed (threshold <= 0)
  if (below(1.0)) return nextafter(1.0, 2.0);  // everything with a defined angle is allowed
  double lo = 0.0, hi = 1.0;  // below(lo), !below(hi)
  for (;;) {
    const double mid = lo + (hi - lo) / 2.0;
    if (!(mid > lo && mid < hi)) break;
    if (below(mid)) lo = mid; else hi = mid;
  }
  return hi;
}

int osfm_guided_scratch_bytes(int cap, int64_t n_pairs, size_t *six_bytes, size_t *good_bytes) {
  const int capr = (cap + 63) & ~63;
  *six_bytes = (size_t)n_pairs * 2 * capr * 6 * sizeof(double);
  *good_bytes = (size_t)n_pairs * 2 * capr * sizeof(int32_t);
  return capr;
}

int osfm_launch_guided_pairs(osfm_ctx *ctx, const osfm_store *store, const OsfmGuidedStage &gs, const int32_t *d_pairs, const double *d_poses,
                             int64_t n_pairs, double ratio, int symmetric, int cap, int32_t *d_counts, uint32_t *d_matches, int32_t *d_flags,
                             double *d_six, int32_t *d_good, hipStream_t stream) {
  if (n_pairs == 0) return OSFM_OK;
  OSFM_REQUIRE(n_pairs <= 65535, OSFM_E_INVALID, "guided chunk of %lld pairs", (long long)n_pairs);
  OSFM_REQUIRE(!store->is_binary, OSFM_E_UNSUPPORTED, "guided matching on a binary (Hamming) store is not on the GPU path");
  GuidedPairsArgs a;
  a.tiles = store->d_tiles;
  a.norms = store->d_norms;
  a.tile_off = store->d_tile_off;
  a.counts = store->d_counts;
  a.bearings = gs.d_bearings;
  a.descf = store->is_float ? store->d_descf : nullptr;
  a.seg = store->d_seg;
  a.pairs = d_pairs;
  a.poses = d_poses;
  a.n_pairs = n_pairs;
  a.capr = (cap + 63) & ~63;
  a.cstar = gs.cstar;
  a.ratio = ratio;
  a.symmetric = symmetric;
  a.cap = cap;
  a.six = d_six;
  a.good = d_good;
  a.out_counts = d_counts;
  a.out_matches = d_matches;
  a.out_flags = d_flags;
  a.wpad = kGuidedWpad;
  const unsigned blocks = (unsigned)((store->max_count + 255) / 256), qblocks = (unsigned)((store->max_count + 63) / 64);
  hipLaunchKernelGGL(guided_pairs_prep_kernel, dim3(blocks ? blocks : 1, 2, (unsigned)n_pairs), dim3(256), 0, stream, a);
  if (store->is_float)  // root-SIFT and the like: the candidates' distances are float sums in the oracle's order, from the float rows
    hipLaunchKernelGGL(guided_pairs_match_kernel<true>, dim3(qblocks ? qblocks : 1, 2, (unsigned)n_pairs), dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL(guided_pairs_match_kernel<false>, dim3(qblocks ? qblocks : 1, 2, (unsigned)n_pairs), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(guided_pairs_emit_kernel, dim3((unsigned)n_pairs), dim3(256), 0, stream, a);
  OSFM_HIP(hipGetLastError());
  return OSFM_OK;
}
