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
