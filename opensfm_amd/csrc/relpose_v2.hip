// relpose_v2.hip -- the cooperative organisation of calibrated robust matching (relpose_coop.h) as a kernel; OPT-IN through
// OSFM_RELPOSE_V2=1 (osfm_relpose_pairs, relpose.hip).  Same entry point, same results bit for bit (tests/test_relpose_core_host.py
// pins both organisations against the oracle); written at the end of round 1 from the ISA analysis of the first kernel (DESIGN.md
// 7.2) and NOT yet run on an MI355X -- it exists so that round 2 can start with a measurement instead of a rewrite.
#include "osfm_internal.h"
#include "relpose_coop.h"

using namespace osfm_rp;

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
    for (int i = lane; i < n; i += kWave) f(i);
    __syncthreads();
  }
  template <class P>
  __device__ int count_if(int n, P p) {
    int c = 0;
    for (int base = 0; base < n; base += kWave) {
      const int i = base + lane;
      const bool b = i < n && p(i);
      c += __popcll(__ballot(b));
    }
    return c;
  }
  template <class P>
  __device__ int compact(int n, P p, int *out) {
    int c = 0;
    for (int base = 0; base < n; base += kWave) {
      const int i = base + lane;
      const bool b = i < n && p(i);
      const unsigned long long m = __ballot(b);
      if (b) out[c + __popcll(m & ((1ull << lane) - 1ull))] = i;
      c += __popcll(m);
    }
    __syncthreads();
    return c;
  }
};

struct PairOut2 {  // mirrors osfm_relpose_result
  double model[12], lo_model[12], R[9], t[3];
  int32_t score, iterations, n_inliers, pad;
};
static_assert(sizeof(PairOut2) == sizeof(osfm_relpose_result), "PairOut2 must mirror osfm_relpose_result");

__global__ __launch_bounds__(kWave) void relpose_pairs_kernel_v2(const double *__restrict__ b1, const double *__restrict__ b2,
                                                                 const int64_t *__restrict__ offsets, int n_pairs, RansacParams prm,
                                                                 int refine_iterations, int mode, const double *__restrict__ stop_bound,
                                                                 int *inl_ws, int *sub_ws, uint8_t *mask, PairOut2 *out) {
  __shared__ WaveShared sh;
  __shared__ CoopShared co;
  const int p = (int)blockIdx.x;
  if (p >= n_pairs) return;
  GpuWave w{(int)threadIdx.x};
  const int64_t o = offsets[p];
  const int n = (int)(offsets[p + 1] - o);
  PairWork P{b1 + 3 * o, b2 + 3 * o, n, nullptr, inl_ws + o, sub_ws + o, stop_bound + o + p};
  for (int i = w.lane; i < n; i += kWave) mask[o + i] = 0;
  PairOut2 r;
  for (int i = 0; i < 9; i++) r.R[i] = 0.0;
  for (int i = 0; i < 3; i++) r.t[i] = 0.0;
  r.pad = 0;
  const int *list;
  int count;
  if (mode == 0) {
    RansacResult rr;
    ransac_relative_pose_seq(w, sh, co, P, prm, rr);
    for (int i = 0; i < 12; i++) {
      r.model[i] = rr.model[i];
      r.lo_model[i] = rr.lo_model[i];
    }
    r.score = rr.best_score;
    r.iterations = rr.iterations_run;
    list = P.inliers;
    count = rr.best_score;
  } else {
    MatchResult mr;
    robust_match_calibrated_seq(w, sh, co, P, prm, refine_iterations, mr);
    for (int i = 0; i < 12; i++) {
      r.model[i] = mr.ransac.model[i];
      r.lo_model[i] = mr.ransac.lo_model[i];
    }
    for (int i = 0; i < 9; i++) r.R[i] = mr.R[i];
    for (int i = 0; i < 3; i++) r.t[i] = mr.t[i];
    r.score = mr.ransac.best_score;
    r.iterations = mr.ransac.iterations_run;
    list = P.subset;
    count = mr.n_inliers;
  }
  r.n_inliers = count;
  __syncthreads();
  for (int i = w.lane; i < count; i += kWave) mask[o + list[i]] = 1;
  if (w.lane == 0) out[p] = r;
}

}  // namespace

// launched by osfm_relpose_pairs (relpose.hip) in place of relpose_pairs_kernel when OSFM_RELPOSE_V2=1; all pointers are device memory
int osfm_launch_relpose_v2(osfm_ctx *ctx, const double *d_b1, const double *d_b2, const int64_t *d_off, int n_pairs, const double thr_angle,
                           const double thr_score, int iterations, double probability, int use_lo, int lo_iterations, int refine_iterations,
                           int mode, const double *d_stop, int *d_inl, int *d_sub, uint8_t *d_mask, void *d_out) {
  const RansacParams rp{thr_angle, thr_score, iterations, probability, use_lo, lo_iterations, kWave};
  hipLaunchKernelGGL(relpose_pairs_kernel_v2, dim3(n_pairs), dim3(kWave), 0, ctx->stream, d_b1, d_b2, d_off, n_pairs, rp, refine_iterations,
                     mode, d_stop, d_inl, d_sub, d_mask, (PairOut2 *)d_out);
  OSFM_HIP(hipGetLastError());
  return OSFM_OK;
}
