// calib.hip -- the calibrated branch of matching.match (opensfm/matching.py:563-634 with robust_match -> robust_match_calibrated,
// :871-929) for a whole chunk of pairs, device-resident between the descriptor stage and the geometric stage:
//
//   osfm_store_bearings           camera.pixel_bearing_many of every feature of the store, once per (store, cameras)
//   osfm_calibrated_filter_chunk  the robust_matching_min_match gate, the gather of the matched bearings, the LO-RANSAC +
//                                 refinement rounds (relpose.hip), the ordered compaction of the surviving matches and the second
//                                 gate -- in place on the chunk's count / match buffers, like the fundamental-matrix stage.
// The only host traffic between the two stages is the per-pair match count (4 bytes per pair) that sizes the gather.
#include <vector>

#include "osfm_internal.h"
#include "relpose_core.h"

using namespace osfm_rp;

namespace {

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

// one workgroup per image: bearings of its features in the store's padded layout
__global__ void store_bearings_kernel(const double *__restrict__ pts, const int64_t *__restrict__ tile_off, const int32_t *__restrict__ counts,
                                      const int32_t *__restrict__ cam_model, const double *__restrict__ cam_params, double *__restrict__ out) {
  const int im = blockIdx.x;
  const int n = counts[im];
  const int64_t row0 = tile_off[im] * 32;
  double par[16];
  for (int i = 0; i < 16; i++) par[i] = cam_params[(size_t)im * 16 + i];
  const int model = cam_model[im];
  for (int r = threadIdx.x; r < n; r += blockDim.x) {
    double b[3];
    pixel_bearing_generic(model, par, pts[(row0 + r) * 2], pts[(row0 + r) * 2 + 1], b);
    out[(row0 + r) * 3] = b[0];
    out[(row0 + r) * 3 + 1] = b[1];
    out[(row0 + r) * 3 + 2] = b[2];
  }
}

// one workgroup per kept pair: b1[k] = bearing of feature i_k of the first image, b2[k] = of feature j_k of the second
__global__ void gather_bearings_kernel(const int32_t *__restrict__ kept, const int64_t *__restrict__ off, const int32_t *__restrict__ pairs,
                                       const int64_t *__restrict__ tile_off, const uint32_t *__restrict__ matches, int cap,
                                       const double *__restrict__ bearings, double *__restrict__ b1, double *__restrict__ b2) {
  const int q = blockIdx.x;
  const int p = kept[q];
  const int64_t o = off[q];
  const int n = (int)(off[q + 1] - o);
  const int64_t r1 = tile_off[pairs[2 * p]] * 32, r2 = tile_off[pairs[2 * p + 1]] * 32;
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    const uint32_t m = matches[(size_t)p * cap + k];
    const double *s1 = bearings + (r1 + (m & 0xFFFFu)) * 3, *s2 = bearings + (r2 + (m >> 16)) * 3;
    for (int a = 0; a < 3; a++) {
      b1[(o + k) * 3 + a] = s1[a];
      b2[(o + k) * 3 + a] = s2[a];
    }
  }
}

// one wavefront per pair of the chunk: matches[inliers] in order (matching.py:903), then the second gate (matching.py:632-634);
// pairs that did not reach the geometric stage end with a count of 0 (matching.py:590-598)
__global__ __launch_bounds__(64) void compact_by_mask_kernel(const int32_t *__restrict__ slot_of_pair, const int64_t *__restrict__ off,
                                                             const uint8_t *__restrict__ mask, int cap, int min_match, int32_t *counts,
                                                             uint32_t *matches) {
  const int p = blockIdx.x, lane = threadIdx.x;
  const int q = slot_of_pair[p];
  if (q < 0) {
    if (lane == 0) counts[p] = 0;
    return;
  }
  const int64_t o = off[q];
  const int n = (int)(off[q + 1] - o);
  uint32_t *mp = matches + (size_t)p * cap;
  int c = 0;
  for (int base = 0; base < n; base += 64) {
    const int k = base + lane;
    const bool keep = k < n && mask[o + k] != 0;
    const uint32_t m = k < n ? mp[k] : 0u;
    const unsigned long long bal = __ballot(keep);
    __syncthreads();  // every read of this step precedes its writes (they land at or below the positions just read)
    if (keep) mp[c + __popcll(bal & ((1ull << lane) - 1ull))] = m;
    c += __popcll(bal);
    __syncthreads();
  }
  if (lane == 0) counts[p] = (c >= min_match && c > 0) ? c : 0;
}

}  // namespace

int osfm_store_bearings(osfm_ctx *ctx, const osfm_store *store, const int32_t *cam_model, const double *cam_params, double **d_out) {
  *d_out = nullptr;
  const int n = store->n_images;
  for (int i = 0; i < n; i++)
    OSFM_REQUIRE(cam_model[i] >= OSFM_CAMERA_PERSPECTIVE && cam_model[i] <= OSFM_CAMERA_SPHERICAL, OSFM_E_INVALID, "image %d: camera model %d", i,
                 cam_model[i]);
  const int64_t rows = (store->tile_off[n] + 4) * 32;
  DevBuf d_model, d_par;
  double *d_b = nullptr;
  OSFM_HIP(d_model.alloc((size_t)n * 4));
  OSFM_HIP(d_par.alloc((size_t)n * 16 * 8));
  OSFM_HIP(hipMalloc((void **)&d_b, (size_t)rows * 3 * 8));
  hipError_t e = hipSuccess;
  if (n > 0) {
    e = hipMemcpyAsync(d_model.p, cam_model, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_par.p, cam_params, (size_t)n * 16 * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(store_bearings_kernel, dim3(n), dim3(256), 0, ctx->stream, store->d_pts, store->d_tile_off, store->d_counts,
                         d_model.as<int32_t>(), d_par.as<double>(), d_b);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  if (e != hipSuccess) {
    (void)hipFree(d_b);
    OSFM_HIP(e);
  }
  *d_out = d_b;
  return OSFM_OK;
}

int osfm_calibrated_filter_chunk(osfm_ctx *ctx, const osfm_store *store, const OsfmCalibStage &cs, const int32_t *h_pairs, const int32_t *d_pairs,
                                 int64_t n_pairs, int cap, int min_match, int32_t *d_counts, uint32_t *d_matches, hipStream_t stream,
                                 int64_t *pairs_filtered) {
  (void)h_pairs;
  if (pairs_filtered) *pairs_filtered = 0;
  if (n_pairs == 0) return OSFM_OK;
  std::vector<int32_t> counts((size_t)n_pairs), kept, slot((size_t)n_pairs, -1);
  OSFM_HIP(hipMemcpyAsync(counts.data(), d_counts, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, stream));
  OSFM_HIP(hipStreamSynchronize(stream));
  std::vector<int64_t> off(1, 0);
  for (int64_t p = 0; p < n_pairs; p++) {
    const int c = counts[(size_t)p] > cap ? cap : counts[(size_t)p];
    if (c >= min_match && c > 0) {  // matching.py:590-598
      slot[(size_t)p] = (int32_t)kept.size();
      kept.push_back((int32_t)p);
      off.push_back(off.back() + c);
    }
  }
  const int nk = (int)kept.size();
  const int64_t total = off.back();
  if (pairs_filtered) *pairs_filtered = nk;
  OsfmPoolBuf d_kept, d_slot, d_off, d_b1, d_b2, d_mask, d_out;  // blocks of the context's cache (released after the stream has drained)
  OSFM_HIP(d_slot.alloc(ctx, (size_t)n_pairs * 4));
  OSFM_HIP(hipMemcpyAsync(d_slot.p, slot.data(), (size_t)n_pairs * 4, hipMemcpyHostToDevice, stream));
  OSFM_HIP(d_off.alloc(ctx, (size_t)(nk + 1) * 8));
  OSFM_HIP(hipMemcpyAsync(d_off.p, off.data(), (size_t)(nk + 1) * 8, hipMemcpyHostToDevice, stream));
  OSFM_HIP(d_mask.alloc(ctx, (size_t)total));
  if (nk > 0) {
    OSFM_HIP(d_kept.alloc(ctx, (size_t)nk * 4));
    OSFM_HIP(hipMemcpyAsync(d_kept.p, kept.data(), (size_t)nk * 4, hipMemcpyHostToDevice, stream));
    OSFM_HIP(d_b1.alloc(ctx, (size_t)total * 24));
    OSFM_HIP(d_b2.alloc(ctx, (size_t)total * 24));
    OSFM_HIP(d_out.alloc(ctx, (size_t)nk * sizeof(osfm_relpose_result)));
    hipLaunchKernelGGL(gather_bearings_kernel, dim3(nk), dim3(128), 0, stream, d_kept.as<int32_t>(), d_off.as<int64_t>(), d_pairs, store->d_tile_off,
                       d_matches, cap, cs.d_bearings, d_b1.as<double>(), d_b2.as<double>());
    OSFM_HIP(hipGetLastError());
    const int rc = osfm_relpose_run_device(ctx, stream, d_b1.as<double>(), d_b2.as<double>(), d_off.as<int64_t>(), off.data(), nk, cs.relpose,
                                           OSFM_RELPOSE_MATCH, d_mask.as<uint8_t>(), d_out.p, nullptr);
    if (rc != OSFM_OK) return rc;
  }
  hipLaunchKernelGGL(compact_by_mask_kernel, dim3((unsigned)n_pairs), dim3(64), 0, stream, d_slot.as<int32_t>(), d_off.as<int64_t>(),
                     d_mask.as<uint8_t>(), cap, min_match, d_counts, d_matches);
  OSFM_HIP(hipGetLastError());
  OSFM_HIP(hipStreamSynchronize(stream));  // the buffers above are released on return
  return OSFM_OK;
}
