// relpose.hip -- calibrated (essential-matrix) robust matching on gfx950.
//
// reference: matching.robust_match_calibrated (opensfm/matching.py:871-903), pyrobust.ransac_relative_pose
// (opensfm/src/robust/src/instanciations.cc:33-48, robust_estimator.h:37-119), Camera::BearingsMany (opensfm/src/geometry/camera.cc).
// The numerics live in relpose_core.h (per lane), the organisation of the LO-RANSAC in relpose_rounds.h: all pairs of a batch go
// through rounds of kernels -- walk (one wavefront per pair: scoring + the reference's decision rules + the draws of the samples
// it will need next), solve5a / eig5 / vec5 / solveN (five-point stage A: one LANE per problem; its eigenvalues: sixteen lanes per problem; one lane per
// (problem, eigenvalue) for the solutions; one lane per N-point problem -- all from work lists; matrices in LDS or
// registers), pose (one lane per essential matrix) -- and one launch of the refinement stage at the end.  This file adds the GPU wave policy, the kernels, the round loop and the C ABI.
//
// History (profiles/r02_relpose_*.json): the first version solved 64 speculative five-point problems per wavefront with the
// matrices in scratch memory (4.96 k pairs/s at 300 correspondences); a wavefront-cooperative solver on LDS-resident matrices
// reached 17.6 k pairs/s and stayed latency-bound (hundreds of dependent 10-item steps per solve).  Both were removed in round 2.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <random>
#include <unordered_map>
#include <vector>

#include "osfm_internal.h"
#include "relpose_rounds.h"

using namespace osfm_rp;

namespace {

struct GpuWave {  // one wavefront = one workgroup
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
  __device__ int compact(int n, P p, int *out) {  // ascending indices, as a sequential scan would write them
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
  template <class P>
  __device__ int compact_changed(int n, P p, int *out, int *changed) {  // compact + "did any entry change"
    int c = 0;
    bool diff = false;
    for (int base = 0; base < n; base += kWave) {
      const int i = base + lane;
      const bool b = i < n && p(i);
      const unsigned long long m = __ballot(b);
      if (b) {
        const int k = c + __popcll(m & ((1ull << lane) - 1ull));
        diff |= out[k] != i;
        out[k] = i;
      }
      c += __popcll(m);
    }
    if (__ballot(diff)) *changed = 1;
    __syncthreads();
    return c;
  }
  __device__ int atomic_add(int *p, int v) { return atomicAdd(p, v); }
  // a window of the generator stream next to the wavefront: the draws of lane 0 then cost an LDS read each, not a trip to L2
  __device__ RngView stage_rng(const RngTable &T, uint32_t *buf, int pos, bool want) {
    int n = 0;
    if (want) {
      n = T.size - pos < kRngCache ? T.size - pos : kRngCache;
      if (n < 0) n = 0;
      for (int i = lane; i < n; i += kWave) buf[i] = T.tab[pos + i];
    }
    __syncthreads();
    return RngView{T, buf, pos, n};
  }
};

struct PairOut {  // mirrors osfm_relpose_result
  double model[12], lo_model[12], R[9], t[3];
  int32_t score, iterations, n_inliers, pad;
};
static_assert(sizeof(PairOut) == sizeof(osfm_relpose_result), "PairOut must mirror osfm_relpose_result");

__global__ void rp_init_kernel(Rounds R, long first, long total) {
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < R.n_pairs) pair_init(R, (int)k);
  if (k < total) pair_normalise(R, first + k);
}

__global__ __launch_bounds__(kWave) void rp_walk_kernel(Rounds R) {
  const int p = (int)blockIdx.x;
  if (p >= R.n_pairs) return;
  __shared__ WalkShared sh;
  GpuWave w{(int)threadIdx.x};
  pair_walk(w, sh, R, p);
}

// LDS of the solver kernels: element-major, lane-minor (LaneArr stride 64): element e of lane l at [e * 64 + l]
constexpr int kStageADoubles = 36 + 200;  // basis | M (the epipolar system first, the action matrix rows last)
constexpr int kStageAInts = 9;            // colperm
constexpr size_t kStageALds = (size_t)kWave * (kStageADoubles * sizeof(double) + kStageAInts * sizeof(int));
constexpr size_t kStageBLds = (size_t)kWave * 100 * sizeof(double);  // one 10 x 10 matrix per lane: three wavefronts per CU
static_assert(3 * kStageBLds <= 160 * 1024, "stage B is sized for three wavefronts per CU");

__global__ __launch_bounds__(kWave) void rp_solve5a_kernel(Rounds R, int count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int k = blockIdx.x * kWave + threadIdx.x;
  if (k >= count) return;
  __builtin_amdgcn_s_setprio(3);  // the five-point chain is the round's critical path: issue ahead of the N-point waves on the same SIMD
  typedef LaneArr<double, kWave> D;
  typedef LaneArr<int, kWave> I;
  const D base{(double *)smem + threadIdx.x};
  const I ibase{(int *)(smem + (size_t)kWave * kStageADoubles * sizeof(double)) + threadIdx.x};
  solve5_stage_a(R, k, base, base + 36, ibase);
}

// Stage B1: the eigenvalues of the 10 x 10 action matrices, kEigGroup lanes per problem (real_eigenvalues10_group), four problems per
// wavefront, sixteen per workgroup; a problem's matrix is 100 doubles of LDS its lanes share
constexpr int kEigBlock = 256;
constexpr int kEigPerBlock = kEigBlock / kEigGroup;
__global__ __launch_bounds__(kEigBlock) void rp_eig5_kernel(Rounds R, int count) {
  __shared__ double mats[kEigPerBlock][104];
  const int g = threadIdx.x / kEigGroup, glane = threadIdx.x % kEigGroup;
  const int k = blockIdx.x * kEigPerBlock + g;
  if (k >= count) return;
  __builtin_amdgcn_s_setprio(3);
  solve5_stage_b1_group(R, k, &mats[g][0], glane);
}
// Stage B2: one lane per (problem, eigenvalue): its 10 x 10 elimination in LDS (element-major, lane-minor)
__global__ __launch_bounds__(kWave) void rp_vec5_kernel(Rounds R, int count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int q = blockIdx.x * kWave + threadIdx.x;
  if (q >= count * kMaxModels) return;
  __builtin_amdgcn_s_setprio(3);
  solve5_stage_b2(R, q, LaneArr<double, kWave>{(double *)smem + threadIdx.x});
}

// One lane per non-minimal problem.  The 9 x 9 matrix and its eigenvectors (162 doubles) stay in the register file: the rotation
// order of the cyclic Jacobi is the same for every lane, so with the 36 rotations of a sweep unrolled every index is a constant.
__global__ __launch_bounds__(kWave) void rp_solveN_kernel(Rounds R, int count) {
  const int k = blockIdx.x * kWave + threadIdx.x;
  if (k >= count) return;
  double AtA[81], V[81], w[9];
  solveN_problem<double *>(R, k, AtA, V, w);
}

// One lane per essential matrix: the first count5 * 10 lanes belong to the five-point list, the next countN to the N-point list
__global__ __launch_bounds__(256, 4) void rp_pose_kernel(Rounds R, int count5, int countN) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < count5 * kMaxModels)
    pose5_item(R, q);
  else if (q - count5 * kMaxModels < countN)
    poseN_item(R, q - count5 * kMaxModels);
}

// After the rounds: the inlier mask of the RANSAC (mode 0) or the refinement stage of robust_match_calibrated (mode 1), results out
template <int WPE>
__global__ __launch_bounds__(kWave, WPE) void rp_finish_kernel(Rounds R, int mode, double threshold_angle, int refine_iterations, int *sub_ws,
                                                          uint8_t *mask, PairOut *out) {
  __shared__ RefineShared sh;
  const int p = (int)blockIdx.x;
  if (p >= R.n_pairs) return;
  GpuWave w{(int)threadIdx.x};
  const int64_t o = R.offsets[p];
  const int n = (int)(R.offsets[p + 1] - o);
  const PairState &S = R.st[p];
  for (int i = w.lane; i < n; i += kWave) mask[o + i] = 0;
  // the result goes out in pieces (lane 0): a PairOut kept in registers across the refinement is 72 of them in every lane
  __shared__ double Rt[12];
  if (w.lane == 0) {
    PairOut &r = out[p];
    for (int i = 0; i < 12; i++) {
      r.model[i] = S.model[i];
      r.lo_model[i] = S.lo_model[i];
    }
    r.score = S.best_score;
    r.iterations = S.it;
    r.pad = 0;
  }
  if (w.lane < 12) Rt[w.lane] = 0.0;
  __syncthreads();
  const int *list = R.inliers + o;
  int count = S.best_score;
  if (mode == 1) {
    count = 0;
    list = sub_ws + o;
    if (!S.rejected)
      count = robust_match_finish_wave(w, sh, R.b1 + 3 * o, R.b2 + 3 * o, n, S.lo_model, threshold_angle, refine_iterations, sub_ws + o, Rt, Rt + 9);
  }
  __syncthreads();
  for (int i = w.lane; i < count; i += kWave) mask[o + list[i]] = 1;
  if (w.lane == 0) {
    PairOut &r = out[p];
    for (int i = 0; i < 9; i++) r.R[i] = Rt[i];
    for (int i = 0; i < 3; i++) r.t[i] = Rt[9 + i];
    r.n_inliers = count;
  }
}

struct CameraParams {
  double v[16];  // native order [projection][distortion][affine]
};

__global__ void pixel_bearings_kernel(int model, CameraParams cam, const double *__restrict__ px, int n, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double b[3];
  pixel_bearing_generic(model, cam.v, px[2 * i], px[2 * i + 1], b);
  out[3 * i] = b[0];
  out[3 * i + 1] = b[1];
  out[3 * i + 2] = b[2];
}

struct DevBuf {  // frees on scope exit
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

struct Sub {  // a slice of a DevBuf
  void *p;
  template <class T>
  T *as() const {
    return (T *)p;
  }
};

constexpr int kRngTableSize = 1 << 21;  // raw outputs of std::mt19937(42): 1000 iterations with a full LO chain each need ~1.3e5

// the context's device copy of the generator stream, made on first use
int rng_table(osfm_ctx *ctx, RngTable *out) {
  if (!ctx->d_rng_table) {
    std::vector<uint32_t> t((size_t)kRngTableSize);
    std::mt19937 g(42);  // RandomSamplesGenerator(int seed = 42), robust/random_sampler.h:10
    for (auto &v : t) v = (uint32_t)g();
    void *d = nullptr;
    OSFM_HIP(hipMalloc(&d, t.size() * sizeof(uint32_t)));
    const hipError_t e = hipMemcpy(d, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(d);
      OSFM_HIP(e);
    }
    ctx->d_rng_table = d;
  }
  out->tab = (const uint32_t *)ctx->d_rng_table;
  out->size = kRngTableSize;
  return OSFM_OK;
}

int ensure_relpose_attributes(int device) {
  static OsfmPerDeviceOnce once;
  return once.run(device, []() -> int {
    OSFM_HIP(hipFuncSetAttribute((const void *)rp_solve5a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OSFM_HIP(hipFuncSetAttribute((const void *)rp_vec5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    return OSFM_OK;
  });
}

}  // namespace

// The rounds over device-resident bearings: d_b1 / d_b2 (total x 3), d_off (n_pairs + 1), host copy `offsets`.  Fills d_mask (total) and
// d_out (n_pairs results).  Everything is enqueued on `st`; the host waits once per round for three counters.
int osfm_relpose_run_device(osfm_ctx *ctx, hipStream_t st, const double *d_b1, const double *d_b2, const int64_t *d_off, const int64_t *offsets,
                            int n_pairs, const osfm_relpose_params *prm, int mode, uint8_t *d_mask, void *d_out, int *rounds_out) {
  if (rounds_out) *rounds_out = 0;
  if (n_pairs == 0) return OSFM_OK;
  const int64_t total = offsets[n_pairs];
  {
    const int rc = ensure_relpose_attributes(ctx->device);
    if (rc != OSFM_OK) return rc;
  }
  RngTable rng;
  {
    const int rc = rng_table(ctx, &rng);
    if (rc != OSFM_OK) return rc;
  }
  // ShouldStop's bound for every possible best inlier count, with the host libm (see relpose_core.h): one table per distinct pair
  // size n, shared by all pairs of that size (n + 1 pow / log evaluations each)
  std::vector<double> stop;
  std::vector<int64_t> stop_off((size_t)n_pairs);
  {
    if (ctx->stop_probability != prm->probability) {  // the context's tables are for one probability
      ctx->stop_tables.clear();
      ctx->stop_probability = prm->probability;
    }
    if (ctx->stop_tables.size() > 65536) ctx->stop_tables.clear();
    std::unordered_map<int, int64_t> table_of_n;
    for (int p = 0; p < n_pairs; p++) {
      const int n = (int)(offsets[p + 1] - offsets[p]);
      auto it = table_of_n.find(n);
      if (it == table_of_n.end()) {
        it = table_of_n.emplace(n, (int64_t)stop.size()).first;
        auto ct = ctx->stop_tables.find(n);
        if (ct == ctx->stop_tables.end()) {
          std::vector<double> t((size_t)n + 1);
          for (int c = 0; c <= n; c++) t[(size_t)c] = max_iterations_for(c, n > 0 ? n : 1, prm->probability);
          ct = ctx->stop_tables.emplace(n, std::move(t)).first;
        }
        stop.insert(stop.end(), ct->second.begin(), ct->second.end());
      }
      stop_off[(size_t)p] = it->second;
    }
  }
  const int lo = prm->lo_iterations > 0 ? prm->lo_iterations : 1;
  // one device allocation for all work buffers of the batch (hipMalloc / hipFree are expensive and synchronising)
  const size_t cap5 = ((size_t)n_pairs * kMaxSlots + kWave - 1) / kWave * kWave;
  const size_t sizes[] = {
      (size_t)total * 24, (size_t)total * 24, stop.size() * 8, stop_off.size() * 8, (size_t)n_pairs * sizeof(PairState),
      (size_t)n_pairs * kMaxSlots * 5 * 4, (size_t)n_pairs * kMaxSlots * 4, (size_t)n_pairs * kMaxSlots * 4, (size_t)n_pairs * kMaxSlots * 4,
      (size_t)n_pairs * kMaxSlots * kMaxModels * 12 * 8, (size_t)n_pairs * lo * kLoSampleMax * 4, (size_t)n_pairs * lo * 4,
      (size_t)n_pairs * lo * 4, (size_t)n_pairs * lo * 12 * 8, (size_t)total * 4, (size_t)total * 4, (size_t)n_pairs * kMaxSlots * 4,
      (size_t)n_pairs * lo * 4, 4 * 4, cap5 * 60 * 8, cap5 * 36 * 8, cap5 * 4, cap5 * kMaxModels * 9 * 8, (size_t)n_pairs * lo * 9 * 8,
      cap5 * kMaxModels * 8, cap5 * 4, cap5 * kMaxModels * 4};
  constexpr int kBuffers = sizeof(sizes) / sizeof(sizes[0]);
  size_t offs[kBuffers], arena_bytes = 0;
  for (int i = 0; i < kBuffers; i++) {
    offs[i] = arena_bytes;
    arena_bytes += (sizes[i] + 255) / 256 * 256;
  }
  OsfmPoolBuf arena;  // from the context's cache of blocks (the caller holds the context lock): a hipMalloc / hipFree of this size is ~2 ms per call
  OSFM_HIP(arena.alloc(ctx, arena_bytes));
  Sub sub[kBuffers];
  for (int i = 0; i < kBuffers; i++) sub[i].p = (char *)arena.p + offs[i];
  const Sub &d_u1 = sub[0], &d_u2 = sub[1], &d_stop = sub[2], &d_stopoff = sub[3], &d_st = sub[4], &d_sidx = sub[5], &d_posb = sub[6], &d_pos = sub[7],
            &d_nm = sub[8], &d_models = sub[9], &d_lidx = sub[10], &d_lopos = sub[11], &d_look = sub[12], &d_lort = sub[13], &d_inl = sub[14],
            &d_sub = sub[15], &d_l5 = sub[16], &d_lN = sub[17], &d_cnt = sub[18], &d_at6 = sub[19], &d_bas = sub[20], &d_ok5 = sub[21],
            &d_E5 = sub[22], &d_loE = sub[23], &d_wr = sub[24], &d_nreal = sub[25], &d_valid = sub[26];
  static_assert(kBuffers == 27, "buffer list and names must match");
  OSFM_HIP(hipMemcpyAsync(d_stop.p, stop.data(), stop.size() * 8, hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(d_stopoff.p, stop_off.data(), stop_off.size() * 8, hipMemcpyHostToDevice, st));
  Rounds R{d_b1, d_b2, d_u1.as<double>(), d_u2.as<double>(), d_off, n_pairs, d_stop.as<double>(), d_stopoff.as<int64_t>(), rng, 1.0 - cos(prm->threshold), (int)prm->iterations, (int)prm->use_lo,
           (int)prm->lo_iterations, mode == OSFM_RELPOSE_MATCH ? 8 : 5, kMaxSlots, d_st.as<PairState>(), d_sidx.as<int>(), d_posb.as<int>(), d_pos.as<int>(),
           d_nm.as<int>(), d_models.as<double>(), d_lidx.as<int>(), d_lopos.as<int>(), d_look.as<int>(), d_lort.as<double>(), d_inl.as<int>(),
           d_at6.as<double>(), d_bas.as<double>(), d_ok5.as<int>(), d_E5.as<double>(), d_loE.as<double>(), d_l5.as<int>(), d_lN.as<int>(), d_cnt.as<int>(),
           d_wr.as<double>(), d_nreal.as<int>(), d_valid.as<int>()};
  // the side stream and its two events live in the context (creating and destroying them per call is ~1 ms: most of a single pair's call)
  if (!ctx->stream_c) OSFM_HIP(hipStreamCreateWithFlags(&ctx->stream_c, hipStreamNonBlocking));
  for (int q = 0; q < 2; q++)
    if (!ctx->ev_rp[q]) OSFM_HIP(hipEventCreateWithFlags(&ctx->ev_rp[q], hipEventDisableTiming));
  struct Stream2 {
    hipStream_t s = nullptr;
  } st2;
  st2.s = ctx->stream_c;
  hipEvent_t ev_fork = ctx->ev_rp[0], ev_join = ctx->ev_rp[1];
  {
    const long first = (long)offsets[0], count = (long)(total - offsets[0]);  // this batch's correspondences
    const long items = std::max<long>(n_pairs, count);
    hipLaunchKernelGGL(rp_init_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, R, first, count);
  }
  OSFM_HIP(hipMemsetAsync(d_cnt.p, 0, 16, st));
  int rounds = 0;
  for (;;) {
    int h[4] = {0, 0, 0, 0};
    hipLaunchKernelGGL(rp_walk_kernel, dim3(n_pairs), dim3(kWave), 0, st, R);
    OSFM_HIP(hipMemcpyAsync(h, d_cnt.p, 16, hipMemcpyDeviceToHost, st));
    OSFM_HIP(hipMemsetAsync(d_cnt.p, 0, 12, st));  // [3] (table overflow) is sticky
    OSFM_HIP(hipStreamSynchronize(st));
    rounds++;
    OSFM_REQUIRE(h[3] == 0, OSFM_E_UNSUPPORTED, "osfm_relpose_pairs: the tabulated mt19937 stream (%d outputs) is too short for this input",
                 kRngTableSize);
    if (h[2] == 0) break;
    // the two work lists of a round are independent: the N-point problems (registers only) run on a second stream next to the
    // five-point ones (one LDS-filling wavefront per CU), and the next walk waits for both
    if (h[1] > 0) {
      OSFM_HIP(hipEventRecord(ev_fork, st));
      OSFM_HIP(hipStreamWaitEvent(st2.s, ev_fork, 0));
      hipLaunchKernelGGL(rp_solveN_kernel, dim3((h[1] + kWave - 1) / kWave), dim3(kWave), 0, st2.s, R, h[1]);
      OSFM_HIP(hipEventRecord(ev_join, st2.s));
    }
    if (h[0] > 0) {
      hipLaunchKernelGGL(rp_solve5a_kernel, dim3((h[0] + kWave - 1) / kWave), dim3(kWave), kStageALds, st, R, h[0]);
      hipLaunchKernelGGL(rp_eig5_kernel, dim3((h[0] + kEigPerBlock - 1) / kEigPerBlock), dim3(kEigBlock), 0, st, R, h[0]);
      hipLaunchKernelGGL(rp_vec5_kernel, dim3((h[0] * kMaxModels + kWave - 1) / kWave), dim3(kWave), kStageBLds, st, R, h[0]);
    }
    if (h[1] > 0) OSFM_HIP(hipStreamWaitEvent(st, ev_join, 0));
    {
      const long lanes = (long)h[0] * kMaxModels + h[1];
      if (lanes > 0) hipLaunchKernelGGL(rp_pose_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, R, h[0], h[1]);
    }
    OSFM_HIP(hipGetLastError());
    OSFM_REQUIRE(rounds < 100000, OSFM_E_NUMERIC, "osfm_relpose_pairs: the rounds do not terminate");
  }
  static const int finish_wpe = getenv("OSFM_RP_FINISH_WPE") ? atoi(getenv("OSFM_RP_FINISH_WPE")) : 2;
  if (finish_wpe == 2)
    hipLaunchKernelGGL(rp_finish_kernel<2>, dim3(n_pairs), dim3(kWave), 0, st, R, mode, prm->threshold, (int)prm->refine_iterations, d_sub.as<int>(), d_mask,
                       (PairOut *)d_out);
  else
    hipLaunchKernelGGL(rp_finish_kernel<1>, dim3(n_pairs), dim3(kWave), 0, st, R, mode, prm->threshold, (int)prm->refine_iterations, d_sub.as<int>(), d_mask,
                       (PairOut *)d_out);
  OSFM_HIP(hipGetLastError());
  OSFM_HIP(hipStreamSynchronize(st));  // the work buffers above are released on return
  if (rounds_out) *rounds_out = rounds;
  return OSFM_OK;
}

extern "C" int osfm_pixel_bearings(osfm_ctx *ctx, int model, const double *cam, const double *px, int n, double *bearings) {
  OSFM_REQUIRE(ctx && (cam || model == OSFM_CAMERA_SPHERICAL) && (n == 0 || (px && bearings)), OSFM_E_INVALID,
               "osfm_pixel_bearings: null argument");
  OSFM_REQUIRE(model >= OSFM_CAMERA_PERSPECTIVE && model <= OSFM_CAMERA_SPHERICAL, OSFM_E_INVALID, "osfm_pixel_bearings: camera model %d", model);
  OSFM_REQUIRE(n >= 0, OSFM_E_INVALID, "osfm_pixel_bearings: n < 0");
  CameraParams cp;
  {
    int proj, kind, nd, na;
    camera_layout(model, &proj, &kind, &nd, &na);
    const int np = (proj == 2 ? 1 : 0) + nd + na;
    for (int i = 0; i < 16; i++) cp.v[i] = i < np ? cam[i] : 0.0;
  }
  if (n == 0) return OSFM_OK;
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  DevBuf d_px, d_out;
  OSFM_HIP(d_px.alloc((size_t)n * 16));
  OSFM_HIP(d_out.alloc((size_t)n * 24));
  OSFM_HIP(hipMemcpyAsync(d_px.p, px, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(pixel_bearings_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, model, cp, d_px.as<double>(), n,
                     d_out.as<double>());
  OSFM_HIP(hipGetLastError());
  OSFM_HIP(hipMemcpyAsync(bearings, d_out.p, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
  OSFM_HIP(hipStreamSynchronize(ctx->stream));
  return OSFM_OK;
}

extern "C" int osfm_relpose_pairs(osfm_ctx *ctx, const double *b1, const double *b2, const int64_t *offsets, int n_pairs,
                                  const osfm_relpose_params *prm, int mode, osfm_relpose_result *results, uint8_t *mask,
                                  double *kernel_ms) {
  OSFM_REQUIRE(ctx && offsets && prm && (n_pairs == 0 || results), OSFM_E_INVALID, "osfm_relpose_pairs: null argument");
  OSFM_REQUIRE(n_pairs >= 0, OSFM_E_INVALID, "osfm_relpose_pairs: n_pairs < 0");
  OSFM_REQUIRE(mode == OSFM_RELPOSE_RANSAC || mode == OSFM_RELPOSE_MATCH, OSFM_E_INVALID, "osfm_relpose_pairs: mode %d", mode);
  OSFM_REQUIRE(prm->lo_iterations <= kLoIterMax, OSFM_E_UNSUPPORTED, "osfm_relpose_pairs: more than %d local optimisation iterations", kLoIterMax);
  OSFM_REQUIRE(prm->iterations >= 0 && prm->lo_iterations >= 0 && prm->refine_iterations >= 0 && prm->threshold > 0 &&
                   prm->probability > 0 && prm->probability < 1,
               OSFM_E_INVALID, "osfm_relpose_pairs: bad parameters");
  if (kernel_ms) *kernel_ms = 0.0;
  if (n_pairs == 0) return OSFM_OK;
  OSFM_REQUIRE(offsets[0] == 0, OSFM_E_INVALID, "osfm_relpose_pairs: offsets[0] must be 0");
  for (int p = 0; p < n_pairs; p++)
    OSFM_REQUIRE(offsets[p + 1] >= offsets[p] && offsets[p + 1] - offsets[p] <= (1 << 24), OSFM_E_INVALID,
                 "osfm_relpose_pairs: offsets must ascend (pair %d)", p);
  const int64_t total = offsets[n_pairs];
  OSFM_REQUIRE(total == 0 || (b1 && b2 && mask), OSFM_E_INVALID, "osfm_relpose_pairs: null bearings / mask");
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  DevBuf d_b1, d_b2, d_off, d_mask, d_out;
  OSFM_HIP(d_b1.alloc((size_t)total * 24));
  OSFM_HIP(d_b2.alloc((size_t)total * 24));
  OSFM_HIP(d_off.alloc((size_t)(n_pairs + 1) * 8));
  OSFM_HIP(d_mask.alloc((size_t)total));
  OSFM_HIP(d_out.alloc((size_t)n_pairs * sizeof(PairOut)));
  if (total) {
    OSFM_HIP(hipMemcpyAsync(d_b1.p, b1, (size_t)total * 24, hipMemcpyHostToDevice, st));
    OSFM_HIP(hipMemcpyAsync(d_b2.p, b2, (size_t)total * 24, hipMemcpyHostToDevice, st));
  }
  OSFM_HIP(hipMemcpyAsync(d_off.p, offsets, (size_t)(n_pairs + 1) * 8, hipMemcpyHostToDevice, st));
  OSFM_HIP(hipEventRecord(ctx->ev[0], st));
  // batches of pairs bound the work space (30 KiB of speculative models and solver hand-over per pair).  (Running several
  // batches side by side on their own streams was measured and is slower: every kernel of a round already occupies its SIMDs'
  // register file or the CUs' LDS, so kernels of different batches queue instead of overlapping.)
  constexpr int kBatch = 65536;
  for (int p0 = 0; p0 < n_pairs; p0 += kBatch) {
    const int np = std::min(kBatch, n_pairs - p0);
    // pair p of a batch is addressed through offsets[p0 + p]: the rounds get the shifted offset table with absolute positions
    const int rc = osfm_relpose_run_device(ctx, st, d_b1.as<double>(), d_b2.as<double>(), d_off.as<int64_t>() + p0, offsets + p0, np, prm, mode,
                                           d_mask.as<uint8_t>(), d_out.as<PairOut>() + p0, nullptr);
    if (rc != OSFM_OK) return rc;
  }
  OSFM_HIP(hipEventRecord(ctx->ev[1], st));
  OSFM_HIP(hipMemcpyAsync(results, d_out.p, (size_t)n_pairs * sizeof(PairOut), hipMemcpyDeviceToHost, st));
  if (total) OSFM_HIP(hipMemcpyAsync(mask, d_mask.p, (size_t)total, hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  if (kernel_ms) {
    float ms = 0.f;
    OSFM_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    *kernel_ms = ms;
  }
  return OSFM_OK;
}
