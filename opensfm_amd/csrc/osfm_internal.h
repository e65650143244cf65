// Internal declarations shared by the translation units of libosfm_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>
#include <string>
#include <cstdlib>
#include <vector>

#include "../../include/osfm_mi355.h"

void osfm_set_error(const char *fmt, ...);
// bumped by every osfm_set_error on this thread: a pooled device block released after an error was recorded may still be in use by
// kernels of the failed call, so its release waits for the device before the block goes back to the cache (OsfmPoolBuf::release)
extern thread_local unsigned osfm_error_epoch;
struct osfm_ctx;
// hipMalloc for the allocations that do not come from the context's block cache: when the device is out of memory while blocks sit
// idle in the cache, the cache is dropped (osfm_ctx_trim_pool) and the allocation retried
hipError_t osfm_malloc_retry(osfm_ctx *ctx, void **p, size_t bytes);

#define OSFM_HIP(call)                                                                      \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      osfm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));  \
      return OSFM_E_HIP;                                                                    \
    }                                                                                       \
  } while (0)

#define OSFM_REQUIRE(cond, code, ...) \
  do {                                \
    if (!(cond)) {                    \
      osfm_set_error(__VA_ARGS__);    \
      return (code);                  \
    }                                 \
  } while (0)

struct osfm_ctx {
  // Every entry point that takes the context holds this lock for the duration of the call: the stream, the events and the
  // device buffers a call allocates are then used by one host thread at a time.  Threads that want concurrency on one GPU
  // create one context each (the Python layer keeps a context per thread, _lib.default_context).  Recursive: the leaf
  // functions are implemented on top of the batched entry points.
  std::recursive_mutex mu;
  int device = 0;
  int num_cus = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void *d_rng_table = nullptr;  // relpose.hip: the tabulated std::mt19937(42) stream, made on first use
  void *d_fr_scratch = nullptr;  // ransac.hip: per-pair states + the list the first kernel hands to the long-run kernel
  size_t fr_scratch_bytes = 0;
  // device buffers of the batched matching calls, kept between calls (hipMalloc / hipFree of the chunk buffers cost ~0.5 ms per call):
  // a free block is reused when it is large enough and at most twice the request; at most kPoolBytes stay cached
  struct PoolBlock {
    void *p;
    size_t bytes;
  };
  std::vector<PoolBlock> pool;
  size_t pool_bytes = 0;
  std::mutex pool_mu;  // the block cache itself: osfm_hahog_extract_batch's worker threads take and return blocks while the caller holds `mu`
  std::vector<hipStream_t> aux_streams;  // hahog.hip: one stream per concurrent image of a batch, made on first use
  static constexpr size_t kPoolBytes = (size_t)24 << 30;  // (of 288 GB: a bundle adjustment at 5 M observations keeps ~16 GB of slabs between calls)
  // what THIS context may keep cached between calls: kPoolBytes unless OSFM_POOL_BYTES says otherwise (several processes or contexts on one
  // GPU -- OpenSfM's multi-process stages -- each hold their own cache, and an out-of-memory in one cannot reclaim another's: give each a share; osfm_ctx_trim_pool empties it)
  size_t pool_limit = pool_limit_from_env();
  static size_t pool_limit_from_env() {
    const char *e = getenv("OSFM_POOL_BYTES");
    return e ? (size_t)strtoull(e, nullptr, 10) : kPoolBytes;
  }
  hipStream_t stream_b = nullptr;  // second stream of the batched matching calls (gather + D2H of chunk k under the matcher of k + 1)
  size_t match_hint = 0;        // int32 entries of the last batched call's match list: the next call reserves that much up front
  // relpose.hip: ShouldStop's iteration bound for every (pair size n, best inlier count c <= n), tabulated with the host's libm (pow, log);
  // kept between calls -- the pair sizes of a data set repeat, and a cold table for ~300 sizes is ~5 ms of libm calls
  double stop_probability = -1.0;
  std::unordered_map<int, std::vector<double>> stop_tables;
  void *h_pinned = nullptr;     // ba.hip: pinned host memory the LM loop's scalars come back through, made on first use
  hipEvent_t ev_side[2] = {nullptr, nullptr};  // ba.hip: fork / join of the solver's side stream (= stream_b), made on first use
  hipStream_t stream_c = nullptr;              // relpose.hip: side stream of the LO-RANSAC rounds, with its fork / join events; made on first use
  hipEvent_t ev_rp[2] = {nullptr, nullptr};
  size_t h_pinned_bytes = 0;
  void *h_stage = nullptr;      // ba.hip: pinned staging memory of a solve's small uploads (4 MiB), made on first use
  void *d_hahog_tables = nullptr;  // hahog.hip: the orientation mask, the exp table and the descriptor's per-pixel table (constants), made on first use
  std::mutex hahog_mu;             // ... under this lock (the batch's worker threads arrive together)
};

// Tile = 32 descriptors x 128 int8 in MFMA-operand order (4 KiB):
//   byte offset of (row r, k) = (k/32)*1024 + (((k%32)/16)*32 + r)*16 + (k%16)
// i.e. [ks 0..3][lane 0..63][16 B] with lane = half*32 + row, so one wave loads an A/B operand
// of v_mfma_i32_32x32x32_i8 with a single fully coalesced 1 KiB dwordx4 load per k-step.
constexpr int OSFM_TILE_ROWS = 32;
constexpr int OSFM_TILE_BYTES = 4096;
constexpr int OSFM_PAD_NORM = 1 << 23;  // norm of padding rows: never wins a top-2

struct osfm_store {
  osfm_ctx *ctx = nullptr;
  int n_images = 0;
  std::vector<int32_t> counts;      // features per image
  std::vector<int64_t> row_off;     // unpadded row offsets (n_images + 1)
  std::vector<int64_t> tile_off;    // tile offsets (n_images + 1)
  int max_count = 0;
  int8_t *d_tiles = nullptr;        // total_tiles * 4096 : (u8 - 128) in tile order
  int32_t *d_norms = nullptr;       // total_tiles * 32 : sum (u8-128)^2, padding = OSFM_PAD_NORM
  int32_t *d_hneg = nullptr;        // total_tiles * 32 : -ceil(norm / 2), the accumulator seed of the matcher (match.hip)
  float *d_descf = nullptr;         // float store only: total_tiles * 32 rows x 128 floats (padding rows zero), row = tile * 32 + r
  bool is_float = false;            // the descriptors are not integers in [0, 255] (root-SIFT ...): d_descf holds them, d_tiles an 8-bit
                                    // quantisation x^ = round((v - lo) * 255 / (hi - lo)) - 128 whose distances BOUND the float ones (match.hip)
  bool quantised = false;           // float store with a usable quantisation (finite values, sane dynamic range): fused kernel, FQ mode
  float *d_qerr = nullptr;          // float store: per image max_row ||x - x^||_2 in quantised units, rounded up (n_images + 1)
  uint32_t *d_bin = nullptr;        // binary store (AKAZE MLDB / ORB bit strings, matched by Hamming distance): 16 dwords per row
                                    // (tile * 32 + r), rows shorter than 64 bytes and padding rows zero-filled
  bool is_binary = false;
  float *d_seg = nullptr;           // matching_use_segmentation: the 129th descriptor column (label x 35) per row (tile * 32 + r), padding 0
  double *d_pts = nullptr;          // total_tiles * 32 * 2 (padded rows zero)
  int32_t *d_counts = nullptr;      // n_images
  int64_t *d_tile_off = nullptr;    // n_images + 1
  int64_t bytes = 0;
};

// growable array without value-initialisation (the match list is tens of MB per call and every entry is overwritten by a D2H copy)
template <typename T>
struct OsfmRawVec {
  T *p = nullptr;
  size_t n = 0, cap = 0;
  OsfmRawVec() = default;
  OsfmRawVec(const OsfmRawVec &) = delete;
  OsfmRawVec &operator=(const OsfmRawVec &) = delete;
  ~OsfmRawVec() { free(p); }
  bool reserve(size_t want) {
    if (want <= cap) return true;
    T *q = (T *)realloc(p, want * sizeof(T));
    if (!q) return false;
    p = q;
    cap = want;
    return true;
  }
  bool resize(size_t want) {
    if (want > cap && !reserve(want + want / 2)) return false;
    n = want;
    return true;
  }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T *data() { return p; }
  const T *data() const { return p; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};

struct osfm_match_result {
  std::vector<int32_t> counts;
  OsfmRawVec<int32_t> matches;  // total x 2
  // OSFM_MATCH_KEEP_DEVICE: the match rows stay in HBM (the exchange step of the multi-GPU path all-gathers from there) and only
  // reach the host when osfm_result_fetch asks for them; the counts are in both places.
  bool on_device = false;
  int device = 0;
  int32_t *d_counts = nullptr;   // n_pairs
  int32_t *d_matches = nullptr;  // d_total x 2 used, d_cap x 2 allocated
  size_t d_total = 0, d_cap = 0;
  ~osfm_match_result() {
    if (d_counts || d_matches) (void)hipSetDevice(device);
    if (d_counts) (void)hipFree(d_counts);
    if (d_matches) (void)hipFree(d_matches);
  }
};

// A device buffer that comes from / goes back to the context's cache of blocks (the caller holds the context lock), or a plain
// hipMalloc'd one
struct OsfmPoolBuf {
  void *p = nullptr;
  size_t bytes = 0;
  osfm_ctx *pool = nullptr;  // non-null: taken from / returned to the context's cache (the caller holds the context lock)
  unsigned epoch = 0;        // osfm_error_epoch when the block was taken
  ~OsfmPoolBuf() { release(); }
  void release() {
    if (!p) return;
    // error path (an early return after OSFM_HIP / OSFM_REQUIRE): kernels of the failed call may still read or write the block
    if (epoch != osfm_error_epoch) (void)hipDeviceSynchronize();
    bool cached = false;
    if (pool) {
      std::lock_guard<std::mutex> g(pool->pool_mu);
      if (pool->pool_bytes + bytes <= pool->pool_limit && pool->pool.size() < 64) {
        pool->pool.push_back({p, bytes});
        pool->pool_bytes += bytes;
        cached = true;
      }
    }
    if (!cached) (void)hipFree(p);
    p = nullptr;
  }
  hipError_t alloc(size_t want) {
    bytes = want ? want : 16;
    epoch = osfm_error_epoch;
    return hipMalloc(&p, bytes);
  }
  hipError_t alloc(osfm_ctx *ctx, size_t want) {
    want = want ? want : 16;
    pool = ctx;
    epoch = osfm_error_epoch;
    std::lock_guard<std::mutex> g(ctx->pool_mu);
    int best = -1;
    for (int i = 0; i < (int)ctx->pool.size(); ++i)
      if (ctx->pool[i].bytes >= want && ctx->pool[i].bytes <= 2 * want + 4096 && (best < 0 || ctx->pool[i].bytes < ctx->pool[best].bytes)) best = i;
    if (best >= 0) {
      p = ctx->pool[best].p;
      bytes = ctx->pool[best].bytes;
      ctx->pool_bytes -= bytes;
      ctx->pool.erase(ctx->pool.begin() + best);
      return hipSuccess;
    }
    bytes = want;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess && !ctx->pool.empty()) {  // out of memory with blocks cached: drop the cache and try again
      for (auto &b : ctx->pool) (void)hipFree(b.p);
      ctx->pool.clear();
      ctx->pool_bytes = 0;
      (void)hipGetLastError();
      e = hipMalloc(&p, bytes);
    }
    return e;
  }
  template <typename T>
  T *as() {
    return (T *)p;
  }
};

#define OSFM_CTX_LOCK(ctx) std::lock_guard<std::recursive_mutex> osfm_ctx_lock_((ctx)->mu)

// Per-device one-time hipFuncSetAttribute bookkeeping (attributes are per device, the flags must not be process-wide booleans)
struct OsfmPerDeviceOnce {
  std::mutex mu;
  bool done[64] = {false};
  template <class F>
  int run(int device, F f) {
    std::lock_guard<std::mutex> lk(mu);
    if (device >= 0 && device < 64 && done[device]) return OSFM_OK;
    const int rc = f();
    if (rc == OSFM_OK && device >= 0 && device < 64) done[device] = true;
    return rc;
  }
};

// match.hip
int osfm_launch_match(osfm_ctx *ctx, const osfm_store *store, const int32_t *d_pairs, int64_t n_pairs,
                      double ratio, int symmetric, int squared_ratio, int cap, int32_t *d_counts, uint32_t *d_matches,
                      int32_t *d_flags, bool exact_kernel, hipStream_t stream);
// relpose.hip: the LO-RANSAC rounds (+ refinement in MATCH mode) over device-resident bearings; fills d_mask and d_out
int osfm_relpose_run_device(osfm_ctx *ctx, hipStream_t st, const double *d_b1, const double *d_b2, const int64_t *d_off, const int64_t *offsets,
                            int n_pairs, const osfm_relpose_params *prm, int mode, uint8_t *d_mask, void *d_out, int *rounds_out);
// calib.hip: the geometric stage of the calibrated branch on a chunk's device buffers (in place, like osfm_launch_ransac_pairs)
struct OsfmCalibStage {
  const double *d_bearings;  // unit bearings of every feature of the store, padded tile layout (tile * 32 + row) x 3
  const osfm_relpose_params *relpose;
};
int osfm_calibrated_filter_chunk(osfm_ctx *ctx, const osfm_store *store, const OsfmCalibStage &cs, const int32_t *h_pairs, const int32_t *d_pairs,
                                 int64_t n_pairs, int cap, int min_match, int32_t *d_counts, uint32_t *d_matches, hipStream_t stream,
                                 int64_t *pairs_filtered);
int osfm_store_bearings(osfm_ctx *ctx, const osfm_store *store, const int32_t *cam_model, const double *cam_params, double **d_out);
// guided.hip: the descriptor stage of guided matching on a chunk's device buffers (fills counts / matches like osfm_launch_match)
struct OsfmGuidedStage {
  const float *d_bearings;  // float32 bearings of every feature of the store, padded tile layout (tile * 32 + row) x 3
  const double *h_poses;    // n_pairs x 12 on the host: R (row-major) and t of every pair's relative pose
  double cstar;             // osfm_guided_cos_threshold(guided_matching_threshold)
};
double osfm_guided_cos_threshold(double threshold);
int osfm_guided_scratch_bytes(int cap, int64_t n_pairs, size_t *six_bytes, size_t *good_bytes);
int osfm_launch_guided_pairs(osfm_ctx *ctx, const osfm_store *store, const OsfmGuidedStage &gs, const int32_t *d_pairs, const double *d_poses,
                             int64_t n_pairs, double ratio, int symmetric, int cap, int32_t *d_counts, uint32_t *d_matches, int32_t *d_flags,
                             double *d_six, int32_t *d_good, hipStream_t stream);
// ransac.hip
// in place: counts/matches of each pair are replaced by the inliers (or 0 when the pair fails a gate)
int osfm_launch_ransac_pairs(osfm_ctx *ctx, const osfm_store *store, const int32_t *d_pairs,
                             int64_t n_pairs, int cap, int min_match, double thr, double conf,
                             int max_iters, int32_t *d_counts, uint32_t *d_matches, double *d_F_or_null, hipStream_t stream,
                             unsigned long long *d_work_or_null = nullptr);
// correspondences the RANSAC kernels stage in LDS; pairs / calls with more read them from HBM
int osfm_ransac_lds_points();        // the single-problem kernel
int osfm_ransac_pairs_lds_points();  // the batched kernel
int osfm_launch_ransac_single(osfm_ctx *ctx, const double *d_p1, const double *d_p2, int n, double thr,
                              double conf, int max_iters, double *d_F, uint8_t *d_mask,
                              int32_t *d_info);

#ifdef __HIPCC__
// cv2's normL2Sqr_ on an AVX2 build as the oracle restates it (oracle/match_oracle.c l2sqr_f32): four 8-lane accumulators over
// blocks of 32 dimensions, (d0 + d1) + (d2 + d3), then ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)); no contraction
__device__ __forceinline__ float l2sqr_rows_f32(const float *a, const float *b) {
  float acc[4][8];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int l = 0; l < 8; ++l) acc[v][l] = 0.f;
#pragma unroll
  for (int jb = 0; jb < 128; jb += 32)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int l4 = 0; l4 < 8; l4 += 4) {
        const float4 x = *(const float4 *)(a + jb + 8 * v + l4);
        const float4 y = *(const float4 *)(b + jb + 8 * v + l4);
        const float xx[4] = {x.x, x.y, x.z, x.w}, yy[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xx[e] - yy[e];
          const float sq = d * d;
          acc[v][l4 + e] = acc[v][l4 + e] + sq;
        }
      }
  float sv[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) sv[l] = (acc[0][l] + acc[1][l]) + (acc[2][l] + acc[3][l]);
  return ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
}
#endif
