// api.hip -- C ABI entry points of libosfm_mi355.so (context, descriptor store, batched matching).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "osfm_internal.h"

static thread_local char g_err[1024] = "";
thread_local unsigned osfm_error_epoch = 0;

void osfm_set_error(const char *fmt, ...) {
  ++osfm_error_epoch;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *osfm_last_error(void) { return g_err; }
extern "C" const char *osfm_version(void) { return "osfm-mi355 0.1 (gfx950)"; }

extern "C" int osfm_ctx_create(int device, osfm_ctx **out) {
  OSFM_REQUIRE(out != nullptr, OSFM_E_INVALID, "osfm_ctx_create: out is null");
  *out = nullptr;
  int ndev = 0;
  OSFM_HIP(hipGetDeviceCount(&ndev));
  OSFM_REQUIRE(device >= 0 && device < ndev, OSFM_E_INVALID, "device %d out of range (%d visible)", device, ndev);
  OSFM_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  OSFM_HIP(hipGetDeviceProperties(&prop, device));
  osfm_ctx *c = new (std::nothrow) osfm_ctx();
  OSFM_REQUIRE(c != nullptr, OSFM_E_NOMEM, "out of host memory");
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  OSFM_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (int i = 0; i < 8; ++i) OSFM_HIP(hipEventCreate(&c->ev[i]));
  *out = c;
  return OSFM_OK;
}

extern "C" void osfm_ctx_destroy(osfm_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < 8; ++i)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->d_rng_table) (void)hipFree(c->d_rng_table);
  if (c->d_fr_scratch) (void)hipFree(c->d_fr_scratch);
  for (auto &b : c->pool) (void)hipFree(b.p);
  if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
  for (hipStream_t a : c->aux_streams) (void)hipStreamDestroy(a);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->d_hahog_tables) (void)hipFree(c->d_hahog_tables);
  for (int i = 0; i < 2; ++i)
    if (c->ev_side[i]) (void)hipEventDestroy(c->ev_side[i]);
  for (int i = 0; i < 2; ++i)
    if (c->ev_rp[i]) (void)hipEventDestroy(c->ev_rp[i]);
  if (c->stream_c) (void)hipStreamDestroy(c->stream_c);
  delete c;
}

extern "C" int64_t osfm_ctx_trim_pool(osfm_ctx *c) {
  if (!c) return 0;
  OSFM_CTX_LOCK(c);
  (void)hipSetDevice(c->device);
  std::lock_guard<std::mutex> g(c->pool_mu);
  if (!c->pool.empty()) (void)hipDeviceSynchronize();
  const int64_t freed = (int64_t)c->pool_bytes;
  for (auto &b : c->pool) (void)hipFree(b.p);
  c->pool.clear();
  c->pool_bytes = 0;
  return freed;
}

hipError_t osfm_malloc_retry(osfm_ctx *ctx, void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  if (e != hipSuccess && ctx && osfm_ctx_trim_pool(ctx) > 0) {
    (void)hipGetLastError();
    e = hipMalloc(p, bytes ? bytes : 16);
  }
  return e;
}

extern "C" int osfm_ctx_device(const osfm_ctx *c) { return c ? c->device : -1; }
extern "C" int osfm_ctx_num_cus(const osfm_ctx *c) { return c ? c->num_cus : 0; }

// ------------------------------------------------------------------------------------------
// store
// ------------------------------------------------------------------------------------------
extern "C" int osfm_store_create(osfm_ctx *ctx, int n_images, const int32_t *counts, osfm_store **out) {
  OSFM_REQUIRE(ctx && counts && out, OSFM_E_INVALID, "osfm_store_create: null argument");
  OSFM_REQUIRE(n_images >= 0, OSFM_E_INVALID, "n_images < 0");
  OSFM_CTX_LOCK(ctx);
  *out = nullptr;
  osfm_store *s = new (std::nothrow) osfm_store();
  OSFM_REQUIRE(s != nullptr, OSFM_E_NOMEM, "out of host memory");
  s->ctx = ctx;
  s->n_images = n_images;
  s->counts.assign(counts, counts + n_images);
  s->row_off.resize(n_images + 1);
  s->tile_off.resize(n_images + 1);
  s->row_off[0] = 0;
  s->tile_off[0] = 0;
  for (int i = 0; i < n_images; ++i) {
    if (counts[i] < 0 || counts[i] > OSFM_MAX_FEATURES) {
      osfm_set_error("image %d has %d features (supported: 0..%d)", i, counts[i], OSFM_MAX_FEATURES);
      delete s;
      return OSFM_E_UNSUPPORTED;
    }
    s->row_off[i + 1] = s->row_off[i] + counts[i];
    s->tile_off[i + 1] = s->tile_off[i] + (counts[i] + OSFM_TILE_ROWS - 1) / OSFM_TILE_ROWS;
    if (counts[i] > s->max_count) s->max_count = counts[i];
  }
  (void)hipSetDevice(ctx->device);
  const int64_t nt = s->tile_off[n_images] + 4;  // +4 tiles of slack: chunk loads never run off the end
  hipError_t e;
  bool ok = true;
  ok = ok && (e = osfm_malloc_retry(ctx, (void **)&s->d_tiles, (size_t)nt * OSFM_TILE_BYTES)) == hipSuccess;
  ok = ok && (e = osfm_malloc_retry(ctx, (void **)&s->d_norms, (size_t)nt * 32 * sizeof(int32_t))) == hipSuccess;
  ok = ok && (e = osfm_malloc_retry(ctx, (void **)&s->d_hneg, (size_t)nt * 32 * sizeof(int32_t))) == hipSuccess;
  ok = ok && (e = osfm_malloc_retry(ctx, (void **)&s->d_pts, (size_t)nt * 32 * 2 * sizeof(double))) == hipSuccess;
  ok = ok && (e = osfm_malloc_retry(ctx, (void **)&s->d_counts, (size_t)(n_images + 1) * sizeof(int32_t))) == hipSuccess;
  ok = ok && (e = osfm_malloc_retry(ctx, (void **)&s->d_tile_off, (size_t)(n_images + 1) * sizeof(int64_t))) == hipSuccess;
  if (!ok) {
    osfm_set_error("hipMalloc failed for a store of %lld tiles: %s", (long long)nt, hipGetErrorString(e));
    osfm_store_destroy(s);
    return OSFM_E_NOMEM;
  }
  s->bytes = nt * (OSFM_TILE_BYTES + 32 * 8 + 32 * 16) + (int64_t)(n_images + 1) * 12;
  *out = s;
  return OSFM_OK;
}

extern "C" void osfm_store_destroy(osfm_store *s) {
  if (!s) return;
  if (s->ctx) (void)hipSetDevice(s->ctx->device);
  (void)hipFree(s->d_tiles);
  (void)hipFree(s->d_norms);
  (void)hipFree(s->d_hneg);
  (void)hipFree(s->d_descf);
  (void)hipFree(s->d_seg);
  (void)hipFree(s->d_bin);
  (void)hipFree(s->d_qerr);
  (void)hipFree(s->d_pts);
  (void)hipFree(s->d_counts);
  (void)hipFree(s->d_tile_off);
  delete s;
}

extern "C" int64_t osfm_store_bytes(const osfm_store *s) { return s ? s->bytes : 0; }

template <typename T>
static int store_upload(osfm_store *s, const T *desc, const double *pts) {
  OSFM_REQUIRE(s && desc && pts, OSFM_E_INVALID, "osfm_store_upload: null argument");
  OSFM_CTX_LOCK(s->ctx);
  (void)hipSetDevice(s->ctx->device);
  const int64_t nt = s->tile_off[s->n_images] + 4;
  std::vector<int8_t> tiles((size_t)nt * OSFM_TILE_BYTES, 0);
  std::vector<int32_t> norms((size_t)nt * 32, OSFM_PAD_NORM);
  std::vector<double> hp((size_t)nt * 64, 0.0);
  // integer-valued descriptors in [0, 255] (the HAHOG / SIFT uchar round trip, features.py:526-534) take the exact int8 MFMA path;
  // anything else (root-SIFT floats, ...) is kept as float32 and matched by the exact float kernel
  bool integral = true;
  for (int64_t k = 0, n = s->row_off[s->n_images] * OSFM_DESC_DIM; k < n && integral; ++k) {
    const double v = (double)desc[k];
    integral = (v >= 0.0 && v <= 255.0) && v == std::floor(v);
  }
  s->is_float = !integral;
  // a segmentation column (osfm_store_set_segmentation) is defined on integer-valued descriptors only: the exact int8 kernel it selects
  // would run on the QUANTISED tiles of a float store and return wrong distances without a word
  OSFM_REQUIRE(integral || !s->d_seg, OSFM_E_UNSUPPORTED,
               "osfm_store_upload: the store holds a segmentation column and the new descriptors are not integer-valued (feature_loading.py:126-133)");
  std::vector<float> descf;
  // float store: next to the float rows an 8-bit quantisation on the store's value range [lo, hi]; ||x^_q - x^_t|| differs from the
  // (scaled) float distance by at most the two rows' residual norms, which the fused kernel turns into rigorous accept / reject
  // bounds -- everything it cannot decide is evaluated in float (match.hip, FQ mode)
  double lo = 0.0, scale = 1.0;
  std::vector<float> qerr;
  s->quantised = false;
  if (!integral) {
    descf.assign((size_t)nt * 32 * OSFM_DESC_DIM, 0.0f);
    double hi = 0.0;
    bool finite = true, first = true;
    for (int64_t k = 0, n = s->row_off[s->n_images] * OSFM_DESC_DIM; k < n; ++k) {
      const double v = (double)(float)desc[k];
      if (!std::isfinite(v)) finite = false;
      if (first || v < lo) lo = v;
      if (first || v > hi) hi = v;
      first = false;
    }
    OSFM_REQUIRE(finite, OSFM_E_INVALID, "osfm_store_upload: non-finite descriptor value");
    // outside this range the float32 squares of the reference's own computation lose their relative accuracy (underflow /
    // overflow) and the bounds below would not hold: such stores stay on the exact float kernel
    s->quantised = (hi - lo) > 1e-6 && std::fabs(lo) < 1e15 && std::fabs(hi) < 1e15;
    if (s->quantised) scale = 255.0 / (hi - lo);
    qerr.assign((size_t)s->n_images + 1, 0.0f);
  }
  for (int im = 0; im < s->n_images; ++im) {
    const int n = s->counts[im];
    const T *d = desc + s->row_off[im] * OSFM_DESC_DIM;
    const double *pp = pts + s->row_off[im] * 2;
    double emax = 0.0;
    for (int r = 0; r < n; ++r) {
      const int64_t tile = s->tile_off[im] + r / 32;
      int8_t *t = tiles.data() + tile * OSFM_TILE_BYTES;
      const int rr = r & 31;
      int32_t nrm = 0;
      if (integral) {
        for (int k = 0; k < OSFM_DESC_DIM; ++k) {
          const int q = (int)d[(size_t)r * OSFM_DESC_DIM + k] - 128;
          nrm += q * q;
          t[(k >> 5) * 1024 + ((((k & 31) >> 4) * 32) + rr) * 16 + (k & 15)] = (int8_t)q;
        }
      } else {
        float *f = descf.data() + ((size_t)tile * 32 + rr) * OSFM_DESC_DIM;
        double e2 = 0.0;
        for (int k = 0; k < OSFM_DESC_DIM; ++k) {
          f[k] = (float)d[(size_t)r * OSFM_DESC_DIM + k];
          if (s->quantised) {
            const double x = ((double)f[k] - lo) * scale - 128.0;
            double xq = std::nearbyint(x);
            xq = xq < -128.0 ? -128.0 : (xq > 127.0 ? 127.0 : xq);
            e2 += (x - xq) * (x - xq);
            const int q = (int)xq;
            nrm += q * q;
            t[(k >> 5) * 1024 + ((((k & 31) >> 4) * 32) + rr) * 16 + (k & 15)] = (int8_t)q;
          }
        }
        const double e = std::sqrt(e2) * (1.0 + 1e-9) + 1e-9;
        if (e > emax) emax = e;
      }
      if (integral || s->quantised) norms[(size_t)tile * 32 + rr] = nrm;
      hp[((size_t)tile * 32 + rr) * 2] = pp[2 * r];
      hp[((size_t)tile * 32 + rr) * 2 + 1] = pp[2 * r + 1];
    }
    if (!integral) qerr[im] = std::nextafterf((float)emax, INFINITY);
  }
  if (!integral) {
    if (!s->d_descf) {
      OSFM_REQUIRE(osfm_malloc_retry(s->ctx, (void **)&s->d_descf, descf.size() * sizeof(float)) == hipSuccess, OSFM_E_NOMEM,
                   "osfm_store_upload: out of device memory for %lld float descriptors", (long long)(descf.size() / OSFM_DESC_DIM));
      s->bytes += (int64_t)descf.size() * (int64_t)sizeof(float);
    }
    OSFM_HIP(hipMemcpy(s->d_descf, descf.data(), descf.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!s->d_qerr)
      OSFM_REQUIRE(osfm_malloc_retry(s->ctx, (void **)&s->d_qerr, qerr.size() * sizeof(float)) == hipSuccess, OSFM_E_NOMEM, "osfm_store_upload: out of device memory");
    OSFM_HIP(hipMemcpy(s->d_qerr, qerr.data(), qerr.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  OSFM_HIP(hipMemcpy(s->d_tiles, tiles.data(), tiles.size(), hipMemcpyHostToDevice));
  OSFM_HIP(hipMemcpy(s->d_norms, norms.data(), norms.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  {
    std::vector<int32_t> hneg(norms.size());
    for (size_t k = 0; k < norms.size(); ++k) hneg[k] = -((norms[k] + 1) >> 1);
    OSFM_HIP(hipMemcpy(s->d_hneg, hneg.data(), hneg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  OSFM_HIP(hipMemcpy(s->d_pts, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice));
  std::vector<int32_t> cnt(s->counts);
  cnt.push_back(0);
  OSFM_HIP(hipMemcpy(s->d_counts, cnt.data(), cnt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  OSFM_HIP(hipMemcpy(s->d_tile_off, s->tile_off.data(), s->tile_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  return OSFM_OK;
}

// matching_use_segmentation (feature_loading.py:123-155): the reference appends one column, 35 x the feature's segmentation label,
// to the HAHOG uchar descriptors, so d^2 = d^2_128 + (35 (s1 - s2))^2.  The store keeps the column; stores that have one are matched
// by the exact kernel, which adds the term exactly as cv2's normL2Sqr_ accumulates a 129th element (scalar tail: d += t * t).
extern "C" int osfm_store_set_segmentation(osfm_store *s, const float *column129) {
  OSFM_REQUIRE(s && column129, OSFM_E_INVALID, "osfm_store_set_segmentation: null argument");
  OSFM_REQUIRE(!s->is_float && !s->is_binary, OSFM_E_UNSUPPORTED,
               "segmentation in the descriptor is defined for integer-valued (HAHOG uchar) descriptors only (feature_loading.py:126-133)");
  OSFM_CTX_LOCK(s->ctx);
  OSFM_HIP(hipSetDevice(s->ctx->device));
  const int64_t nt = s->tile_off[s->n_images] + 4;
  std::vector<float> seg((size_t)nt * 32, 0.0f);
  for (int i = 0; i < s->n_images; ++i)
    for (int r = 0; r < s->counts[i]; ++r) {
      const float v = column129[s->row_off[i] + r];
      OSFM_REQUIRE(std::isfinite(v), OSFM_E_INVALID, "osfm_store_set_segmentation: non-finite label of feature %d of image %d", r, i);
      seg[(size_t)s->tile_off[i] * 32 + r] = v;
    }
  if (!s->d_seg) {
    OSFM_REQUIRE(osfm_malloc_retry(s->ctx, (void **)&s->d_seg, seg.size() * sizeof(float)) == hipSuccess, OSFM_E_NOMEM,
                 "osfm_store_set_segmentation: out of device memory");
    s->bytes += (int64_t)seg.size() * 4;
  }
  OSFM_HIP(hipMemcpy(s->d_seg, seg.data(), seg.size() * sizeof(float), hipMemcpyHostToDevice));
  return OSFM_OK;
}

// Binary descriptors (uint8 bit strings: AKAZE MLDB 61 bytes, ORB 32 bytes): the reference's match_brute_force switches cv2 to
// "BruteForce-Hamming" for uint8 arrays (matching.py:737-740).  Rows are zero-padded to 64 bytes (zeros add nothing to a Hamming
// distance); the keypoints go where the other stores keep them, so the gates and the robust stage are unchanged.
extern "C" int osfm_store_upload_binary(osfm_store *s, const uint8_t *desc, int width_bytes, const double *pts) {
  OSFM_REQUIRE(s && desc && pts, OSFM_E_INVALID, "osfm_store_upload_binary: null argument");
  OSFM_REQUIRE(width_bytes >= 1 && width_bytes <= 64, OSFM_E_UNSUPPORTED, "osfm_store_upload_binary: %d bytes per descriptor (1..64)", width_bytes);
  OSFM_REQUIRE(!s->d_seg, OSFM_E_UNSUPPORTED, "osfm_store_upload_binary: the store holds a segmentation column (integer-valued L2 descriptors only)");
  OSFM_CTX_LOCK(s->ctx);
  OSFM_HIP(hipSetDevice(s->ctx->device));
  const int64_t nt = s->tile_off[s->n_images] + 4;
  std::vector<uint32_t> bin((size_t)nt * 32 * 16, 0u);
  std::vector<double> hp((size_t)nt * 64, 0.0);
  for (int i = 0; i < s->n_images; ++i)
    for (int r = 0; r < s->counts[i]; ++r) {
      const int64_t src = s->row_off[i] + r, dst = s->tile_off[i] * 32 + r;
      memcpy((uint8_t *)(bin.data() + (size_t)dst * 16), desc + (size_t)src * width_bytes, (size_t)width_bytes);
      hp[(size_t)dst * 2] = pts[(size_t)src * 2];
      hp[(size_t)dst * 2 + 1] = pts[(size_t)src * 2 + 1];
    }
  if (!s->d_bin) {
    OSFM_REQUIRE(osfm_malloc_retry(s->ctx, (void **)&s->d_bin, bin.size() * sizeof(uint32_t)) == hipSuccess, OSFM_E_NOMEM,
                 "osfm_store_upload_binary: out of device memory");
    s->bytes += (int64_t)bin.size() * 4;
  }
  OSFM_HIP(hipMemcpy(s->d_bin, bin.data(), bin.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  OSFM_HIP(hipMemcpy(s->d_pts, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice));
  std::vector<int32_t> cnt(s->counts);
  cnt.push_back(0);
  OSFM_HIP(hipMemcpy(s->d_counts, cnt.data(), cnt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  OSFM_HIP(hipMemcpy(s->d_tile_off, s->tile_off.data(), s->tile_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  s->is_binary = true;
  s->is_float = false;
  return OSFM_OK;
}

extern "C" int osfm_store_upload_f32(osfm_store *s, const float *desc, const double *pts) {
  return store_upload<float>(s, desc, pts);
}
extern "C" int osfm_store_upload_u8(osfm_store *s, const uint8_t *desc, const double *pts) {
  return store_upload<uint8_t>(s, desc, pts);
}

// ------------------------------------------------------------------------------------------
// batched matching
// ------------------------------------------------------------------------------------------
extern "C" void osfm_match_params_default(osfm_match_params *p) {
  if (!p) return;
  p->lowes_ratio = 0.8;
  p->symmetric = 1;
  p->robust = 1;
  p->robust_matching_min_match = 20;
  p->robust_matching_threshold = 0.004;
  p->ransac_confidence = 0.9999;
  p->ransac_max_iters = 1000;
  p->flags = 0;
}

namespace {
__global__ void gather_matches_kernel(const int32_t *counts, const int64_t *offsets, const uint32_t *matches, int cap,
                                      int32_t *out, long n_pairs, int32_t *counts_out) {
  const long p = blockIdx.x;
  if (p >= n_pairs) return;
  const int n = min(counts[p], cap);
  if (counts_out && threadIdx.x == 0) counts_out[p] = n;
  const int64_t o = offsets[p];
  for (int k = threadIdx.x; k < n; k += blockDim.x) {
    const uint32_t m = matches[p * cap + k];
    out[2 * (o + k)] = (int32_t)(m & 0xFFFFu);
    out[2 * (o + k) + 1] = (int32_t)(m >> 16);
  }
}

using DevBuf = OsfmPoolBuf;
}  // namespace

static int match_pairs_impl(osfm_ctx *ctx, const osfm_store *store, const int32_t *pairs, int64_t n_pairs,
                            const osfm_match_params *params, const OsfmCalibStage *calib, osfm_match_result **out,
                            osfm_match_timings *tm, const OsfmGuidedStage *guided = nullptr) {
  OSFM_REQUIRE(ctx && store && params && out && (pairs || n_pairs == 0), OSFM_E_INVALID, "osfm_match_pairs: null argument");
  OSFM_REQUIRE(n_pairs >= 0, OSFM_E_INVALID, "n_pairs < 0");
  OSFM_REQUIRE(calib || !params->robust || params->robust_matching_min_match >= 15, OSFM_E_UNSUPPORTED,
               "robust_matching_min_match < 15 would take cv2's LMedS branch, which is not implemented");
  *out = nullptr;
  for (int64_t k = 0; k < 2 * n_pairs; ++k)
    OSFM_REQUIRE(pairs[k] >= 0 && pairs[k] < store->n_images, OSFM_E_INVALID, "pair %lld references image %d (store has %d)",
                 (long long)(k / 2), pairs[k], store->n_images);
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  if (tm) memset(tm, 0, sizeof(*tm));

  const int cap = store->max_count > 0 ? store->max_count : 1;
  // 131072 pairs per chunk at cap <= 2048 (1 GiB of match slots per buffer set), fewer for larger images
  const int64_t chunk_pairs = std::max<int64_t>(4096, std::min<int64_t>(1 << 17, ((int64_t)1 << 28) / cap));
  // at least two chunks once there is enough work for that (>= 4096 pairs each): gather + D2H of chunk k run on stream B underneath the
  // matcher of chunk k + 1.  More chunks than memory asks for only add per-chunk fixed costs (measured on the 15 864-pair neighbour
  // list, profiles/r03_prof_neighbour.json: 1 chunk 13.05 ms, 2 chunks 12.54 ms, 4 chunks 13.43 ms).
  int64_t cp = n_pairs < chunk_pairs ? n_pairs : chunk_pairs;
  {
    const char *e = getenv("OSFM_MATCH_CHUNKS");  // measurement knob
    const int64_t nch = e ? std::max(1, atoi(e)) : 2;
    // (the calibrated branch keeps one chunk while memory allows: its geometric stage is a chain of ~12 rounds of latency-bound kernels,
    // ~17 ms whatever the number of pairs below ~10^5 -- two chunks pay for it twice, 59 -> 35 ms on the 15 864-pair neighbour list)
    if (n_pairs >= 8192 && (!calib || e)) cp = std::min<int64_t>(cp, std::max<int64_t>(4096, (n_pairs + nch - 1) / nch));
  }
  if (guided) cp = std::min<int64_t>(cp, 8192);  // 96 B of epipolar vectors + 8 B of results per feature and pair in the chunk's scratch
  const int64_t nchunks = (n_pairs + cp - 1) / (cp > 0 ? cp : 1);
  // Two chunk buffer sets: while stream B runs RANSAC + gather + D2H of chunk k (a few thousand
  // workgroups, latency bound, plus two host round trips), stream A already runs the fused matcher
  // of chunk k + 1, which fills the machine on its own.
  struct ChunkSet {
    DevBuf pairs, counts, matches, flags, offsets, gather;
    DevBuf poses, six, good;  // guided matching: relative poses, per-pair epipolar vectors, per-direction results
    size_t gather_cap = 0;
    hipEvent_t m0 = nullptr, m1 = nullptr, r0 = nullptr, r1 = nullptr, q0 = nullptr, q1 = nullptr;
    ~ChunkSet() {
      for (hipEvent_t ev : {m0, m1, r0, r1, q0, q1})
        if (ev) (void)hipEventDestroy(ev);
    }
  } sets[2];
  const int nsets = nchunks > 1 ? 2 : 1;
  hipError_t e = hipSuccess;
  bool ok = true;
  for (int q = 0; q < nsets; ++q) {
    ok = ok && (e = sets[q].pairs.alloc(ctx, (size_t)cp * 2 * sizeof(int32_t))) == hipSuccess;
    ok = ok && (e = sets[q].counts.alloc(ctx, (size_t)cp * sizeof(int32_t))) == hipSuccess;
    ok = ok && (e = sets[q].flags.alloc(ctx, (size_t)cp * sizeof(int32_t))) == hipSuccess;
    ok = ok && (e = sets[q].offsets.alloc(ctx, (size_t)cp * sizeof(int64_t))) == hipSuccess;
    ok = ok && (e = sets[q].matches.alloc(ctx, (size_t)cp * cap * sizeof(uint32_t))) == hipSuccess;
    if (guided) {
      size_t six_bytes = 0, good_bytes = 0;
      osfm_guided_scratch_bytes(cap, cp, &six_bytes, &good_bytes);
      ok = ok && (e = sets[q].poses.alloc(ctx, (size_t)cp * 12 * sizeof(double))) == hipSuccess;
      ok = ok && (e = sets[q].six.alloc(ctx, six_bytes)) == hipSuccess;
      ok = ok && (e = sets[q].good.alloc(ctx, good_bytes)) == hipSuccess;
    }
    ok = ok && (e = hipEventCreate(&sets[q].m0)) == hipSuccess && (e = hipEventCreate(&sets[q].m1)) == hipSuccess;
    ok = ok && (e = hipEventCreate(&sets[q].r0)) == hipSuccess && (e = hipEventCreate(&sets[q].r1)) == hipSuccess;
    ok = ok && (e = hipEventCreate(&sets[q].q0)) == hipSuccess && (e = hipEventCreate(&sets[q].q1)) == hipSuccess;
  }
  OSFM_REQUIRE(ok, OSFM_E_NOMEM, "hipMalloc failed for match buffers: %s", hipGetErrorString(e));
  if (!ctx->stream_b) OSFM_HIP(hipStreamCreateWithFlags(&ctx->stream_b, hipStreamNonBlocking));
  // measurement knob: OSFM_MATCH_ONE_STREAM=1 puts everything on the matcher's stream (no overlap between chunks)
  hipStream_t stA = ctx->stream, stB = getenv("OSFM_MATCH_ONE_STREAM") ? ctx->stream : ctx->stream_b;
  // The fundamental-matrix RANSAC runs on the MATCHER's stream, right behind its chunk: the matcher owns every SIMD's registers and
  // nearly all LDS (2 x 78 KiB per CU), so kernels of a second stream only get in where a matcher workgroup retires -- measured on the
  // neighbour list, the robust stage takes 1.0 ms on its own and 9 ms of stream time underneath the matcher, which it slows down in
  // turn (tools/prof_neighbour.py, profiles/r03_prof_neighbour.json).  Stream B keeps what does overlap for free: counts D2H, the
  // gather of the match rows and their D2H copy.  The calibrated branch (host-driven rounds) stays on stream B.
  const bool ransac_on_a = !calib && params->robust && !getenv("OSFM_MATCH_RANSAC_STREAM_B");

  DevBuf d_work;  // models x correspondences scored by the RANSAC kernel (osfm_match_timings::ransac_model_points)
  OSFM_HIP(d_work.alloc(ctx, 8));
  OSFM_HIP(hipMemsetAsync(d_work.p, 0, 8, stA));
  OSFM_HIP(hipStreamSynchronize(stA));
  osfm_match_result *res = new (std::nothrow) osfm_match_result();
  OSFM_REQUIRE(res != nullptr, OSFM_E_NOMEM, "out of host memory");
  struct Guard {
    osfm_match_result *r;
    ~Guard() { delete r; }
  } guard{res};
  res->counts.assign((size_t)n_pairs, 0);
  const bool keep_dev = (params->flags & OSFM_MATCH_KEEP_DEVICE) != 0;
  if (keep_dev) {  // the match rows stay in HBM: one growing device buffer instead of the host vector
    res->on_device = true;
    res->device = ctx->device;
    res->d_cap = (ctx->match_hint + ctx->match_hint / 8) / 2 + 1024;  // rows; what the previous call on this context produced
    OSFM_REQUIRE(osfm_malloc_retry(ctx, (void **)&res->d_counts, (size_t)(n_pairs > 0 ? n_pairs : 1) * sizeof(int32_t)) == hipSuccess &&
                     osfm_malloc_retry(ctx, (void **)&res->d_matches, res->d_cap * 2 * sizeof(int32_t)) == hipSuccess,
                 OSFM_E_NOMEM, "hipMalloc failed for the device-resident result of %lld pairs", (long long)n_pairs);
    OSFM_HIP(hipMemsetAsync(res->d_counts, 0, (size_t)(n_pairs > 0 ? n_pairs : 1) * sizeof(int32_t), stA));
    OSFM_HIP(hipStreamSynchronize(stA));
  } else {
    (void)res->matches.reserve(ctx->match_hint + ctx->match_hint / 8);  // what the previous call on this context produced
  }
  std::vector<int32_t> hflags((size_t)cp);
  std::vector<int64_t> hoff((size_t)cp);
  double ms_match = 0.0, ms_ransac = 0.0;
  OSFM_HIP(hipEventRecord(ctx->ev[0], stA));

  auto enqueue_match = [&](int64_t k) -> int {  // stream A: pairs H2D + fused matcher (+ rare exact re-run)
    ChunkSet &S = sets[k & 1];
    const int64_t p0 = k * cp, np = (n_pairs - p0) < cp ? (n_pairs - p0) : cp;
    OSFM_HIP(hipMemcpyAsync(S.pairs.p, pairs + 2 * p0, (size_t)np * 2 * sizeof(int32_t), hipMemcpyHostToDevice, stA));
    OSFM_HIP(hipEventRecord(S.m0, stA));
    int rc;
    if (guided) {  // epipolar-masked descriptor stage (guided.hip); the robust stage below is the same
      OSFM_HIP(hipMemcpyAsync(S.poses.p, guided->h_poses + 12 * p0, (size_t)np * 12 * sizeof(double), hipMemcpyHostToDevice, stA));
      rc = osfm_launch_guided_pairs(ctx, store, *guided, S.pairs.as<int32_t>(), S.poses.as<double>(), np, params->lowes_ratio, params->symmetric, cap,
                                    S.counts.as<int32_t>(), S.matches.as<uint32_t>(), S.flags.as<int32_t>(), S.six.as<double>(), S.good.as<int32_t>(), stA);
      if (rc != OSFM_OK) return rc;
      OSFM_HIP(hipEventRecord(S.m1, stA));
    } else if (params->flags & OSFM_MATCH_EXACT_KERNEL) {
      // debug/cross-check mode: every pair on the exact VALU kernel
      OSFM_HIP(hipMemsetAsync(S.flags.p, 0, (size_t)np * sizeof(int32_t), stA));
      rc = osfm_launch_match(ctx, store, S.pairs.as<int32_t>(), np, params->lowes_ratio, params->symmetric, (params->flags & OSFM_MATCH_SQUARED_RATIO) ? 1 : 0, cap,
                             S.counts.as<int32_t>(), S.matches.as<uint32_t>(), nullptr, true, stA);
      if (rc != OSFM_OK) return rc;
      OSFM_HIP(hipEventRecord(S.m1, stA));
    } else {
      rc = osfm_launch_match(ctx, store, S.pairs.as<int32_t>(), np, params->lowes_ratio, params->symmetric, (params->flags & OSFM_MATCH_SQUARED_RATIO) ? 1 : 0, cap,
                             S.counts.as<int32_t>(), S.matches.as<uint32_t>(), S.flags.as<int32_t>(), false, stA);
      if (rc != OSFM_OK) return rc;
      OSFM_HIP(hipEventRecord(S.m1, stA));
      // rare exact path: pairs whose second-nearest d^2 >= 2^22 (sqrtf is not injective there)
      rc = osfm_launch_match(ctx, store, S.pairs.as<int32_t>(), np, params->lowes_ratio, params->symmetric, (params->flags & OSFM_MATCH_SQUARED_RATIO) ? 1 : 0, cap,
                             S.counts.as<int32_t>(), S.matches.as<uint32_t>(), S.flags.as<int32_t>(), true, stA);
      if (rc != OSFM_OK) return rc;
    }
    if (ransac_on_a) {
      OSFM_HIP(hipEventRecord(S.q0, stA));
      rc = osfm_launch_ransac_pairs(ctx, store, S.pairs.as<int32_t>(), np, cap, params->robust_matching_min_match, params->robust_matching_threshold,
                                    params->ransac_confidence, params->ransac_max_iters, S.counts.as<int32_t>(), S.matches.as<uint32_t>(), nullptr, stA,
                                    d_work.as<unsigned long long>());
      if (rc != OSFM_OK) return rc;
      OSFM_HIP(hipEventRecord(S.q1, stA));
    }
    OSFM_HIP(hipEventRecord(S.r0, stA));  // everything of the descriptor stage (and the F-RANSAC) is enqueued
    return OSFM_OK;
  };

  if (nchunks > 0) {
    const int rc0 = enqueue_match(0);
    if (rc0 != OSFM_OK) return rc0;
  }
  for (int64_t k = 0; k < nchunks; ++k) {
    ChunkSet &S = sets[k & 1];
    const int64_t p0 = k * cp, np = (n_pairs - p0) < cp ? (n_pairs - p0) : cp;
    if (k + 1 < nchunks) {  // its buffer set was released when chunk k - 1 finished (host synchronised on B)
      const int rc1 = enqueue_match(k + 1);
      if (rc1 != OSFM_OK) return rc1;
    }
    OSFM_HIP(hipStreamWaitEvent(stB, S.r0, 0));
    hipEvent_t rb0 = ctx->ev[3], rb1 = ctx->ev[4];
    OSFM_HIP(hipEventRecord(rb0, stB));
    if (calib) {  // essential-matrix branch of robust_match (matching.py:871-929), device-resident between the two stages
      int64_t nf = 0;
      const int rc = osfm_calibrated_filter_chunk(ctx, store, *calib, pairs + 2 * p0, S.pairs.as<int32_t>(), np, cap, params->robust_matching_min_match,
                                                  S.counts.as<int32_t>(), S.matches.as<uint32_t>(), stB, &nf);
      if (rc != OSFM_OK) return rc;
      if (tm) tm->pairs_ransac += nf;
    } else if (params->robust && !ransac_on_a) {
      const int rc = osfm_launch_ransac_pairs(ctx, store, S.pairs.as<int32_t>(), np, cap, params->robust_matching_min_match,
                                              params->robust_matching_threshold, params->ransac_confidence,
                                              params->ransac_max_iters, S.counts.as<int32_t>(), S.matches.as<uint32_t>(), nullptr, stB, d_work.as<unsigned long long>());
      if (rc != OSFM_OK) return rc;
    }
    OSFM_HIP(hipEventRecord(rb1, stB));
    OSFM_HIP(hipMemcpyAsync(res->counts.data() + p0, S.counts.p, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, stB));
    OSFM_HIP(hipMemcpyAsync(hflags.data(), S.flags.p, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, stB));
    OSFM_HIP(hipStreamSynchronize(stB));
    int64_t total = 0;
    for (int64_t q = 0; q < np; ++q) {
      int32_t &cnt = res->counts[(size_t)(p0 + q)];
      if (cnt > cap) cnt = cap;
      hoff[(size_t)q] = total;
      total += cnt;
      if (tm) tm->pairs_exact_path += hflags[(size_t)q] != 0;
    }
    if (keep_dev) {
      const size_t base = res->d_total;
      if (base + (size_t)total > res->d_cap) {  // grow: D2D copy of what is there (stream B is idle, the host waited on it above)
        const size_t ncap = (base + (size_t)total) + (base + (size_t)total) / 2;
        int32_t *nb = nullptr;
        OSFM_REQUIRE(osfm_malloc_retry(ctx, (void **)&nb, ncap * 2 * sizeof(int32_t)) == hipSuccess, OSFM_E_NOMEM, "hipMalloc failed for %lld device-resident matches",
                     (long long)ncap);
        hipError_t ce = base ? hipMemcpyAsync(nb, res->d_matches, base * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, stB) : hipSuccess;
        if (ce == hipSuccess) ce = hipStreamSynchronize(stB);
        if (ce != hipSuccess) {
          (void)hipFree(nb);
          osfm_set_error("growing the device-resident result: %s", hipGetErrorString(ce));
          return OSFM_E_HIP;
        }
        (void)hipFree(res->d_matches);
        res->d_matches = nb;
        res->d_cap = ncap;
      }
      OSFM_HIP(hipMemcpyAsync(S.offsets.p, hoff.data(), (size_t)np * sizeof(int64_t), hipMemcpyHostToDevice, stB));
      hipLaunchKernelGGL(gather_matches_kernel, dim3((unsigned)np), dim3(64), 0, stB, S.counts.as<int32_t>(), S.offsets.as<int64_t>(),
                         S.matches.as<uint32_t>(), cap, res->d_matches + 2 * base, (long)np, res->d_counts + p0);
      OSFM_HIP(hipGetLastError());
      OSFM_HIP(hipStreamSynchronize(stB));  // the chunk's buffer set is handed back to the matcher
      res->d_total = base + (size_t)total;
    } else {
    const size_t base = res->matches.size();
    OSFM_REQUIRE(res->matches.resize(base + (size_t)total * 2), OSFM_E_NOMEM, "out of host memory for %lld matches", (long long)total);
    if (total > 0) {
      if ((size_t)total > S.gather_cap) {
        S.gather.release();
        S.gather_cap = (size_t)total + (size_t)total / 4;
        OSFM_REQUIRE(S.gather.alloc(ctx, S.gather_cap * 2 * sizeof(int32_t)) == hipSuccess, OSFM_E_NOMEM,
                     "hipMalloc failed for gathered matches");
      }
      OSFM_HIP(hipMemcpyAsync(S.offsets.p, hoff.data(), (size_t)np * sizeof(int64_t), hipMemcpyHostToDevice, stB));
      hipLaunchKernelGGL(gather_matches_kernel, dim3((unsigned)np), dim3(64), 0, stB, S.counts.as<int32_t>(),
                         S.offsets.as<int64_t>(), S.matches.as<uint32_t>(), cap, S.gather.as<int32_t>(), (long)np, (int32_t *)nullptr);
      OSFM_HIP(hipGetLastError());
      OSFM_HIP(hipMemcpyAsync(res->matches.data() + base, S.gather.p, (size_t)total * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, stB));
      OSFM_HIP(hipStreamSynchronize(stB));
    }
    }
    if (tm) {
      float ms = 0.f;
      OSFM_HIP(hipEventElapsedTime(&ms, S.m0, S.m1));
      ms_match += ms;
      if (ransac_on_a)
        OSFM_HIP(hipEventElapsedTime(&ms, S.q0, S.q1));
      else
        OSFM_HIP(hipEventElapsedTime(&ms, rb0, rb1));
      ms_ransac += ms;
      tm->match_launches += 1;
    }
  }
  OSFM_HIP(hipEventRecord(ctx->ev[5], stA));
  OSFM_HIP(hipStreamSynchronize(stA));
  if (tm) {
    float ms = 0.f;
    OSFM_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[5]));
    tm->ms_total = ms;
    tm->ms_match_kernel = ms_match;
    tm->ms_ransac_kernel = ms_ransac;
    tm->pairs = n_pairs;
    unsigned long long work = 0;
    OSFM_HIP(hipMemcpy(&work, d_work.p, 8, hipMemcpyDeviceToHost));
    tm->ransac_model_points = (int64_t)work;
  }
  ctx->match_hint = keep_dev ? res->d_total * 2 : res->matches.size();
  guard.r = nullptr;
  *out = res;
  return OSFM_OK;
}

extern "C" int osfm_match_pairs(osfm_ctx *ctx, const osfm_store *store, const int32_t *pairs, int64_t n_pairs,
                                const osfm_match_params *params, osfm_match_result **out, osfm_match_timings *tm) {
  return match_pairs_impl(ctx, store, pairs, n_pairs, params, nullptr, out, tm);
}

extern "C" int osfm_match_pairs_calibrated(osfm_ctx *ctx, const osfm_store *store, const int32_t *cam_model, const double *cam_params,
                                           const int32_t *pairs, int64_t n_pairs, const osfm_match_params *params,
                                           const osfm_relpose_params *relpose, osfm_match_result **out, osfm_match_timings *tm) {
  OSFM_REQUIRE(ctx && store && cam_model && cam_params && params && relpose && out, OSFM_E_INVALID, "osfm_match_pairs_calibrated: null argument");
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  double *d_bearings = nullptr;
  int rc = osfm_store_bearings(ctx, store, cam_model, cam_params, &d_bearings);
  if (rc != OSFM_OK) return rc;
  const OsfmCalibStage cs{d_bearings, relpose};
  rc = match_pairs_impl(ctx, store, pairs, n_pairs, params, &cs, out, tm);
  (void)hipFree(d_bearings);
  return rc;
}

// Guided matching for every pair (matching.py:204-207,260-337: _match_descriptors_guided_impl per pair, then robust_match as usual)
extern "C" int osfm_match_pairs_guided(osfm_ctx *ctx, const osfm_store *store, const float *bearings, const int32_t *pairs, int64_t n_pairs,
                                       const double *poses, double threshold, const osfm_match_params *params, const int32_t *cam_model,
                                       const double *cam_params, const osfm_relpose_params *relpose, osfm_match_result **out,
                                       osfm_match_timings *tm) {
  OSFM_REQUIRE(ctx && store && params && out && (bearings || store->row_off[store->n_images] == 0) && (poses || n_pairs == 0), OSFM_E_INVALID,
               "osfm_match_pairs_guided: null argument");
  OSFM_REQUIRE((cam_model && cam_params && relpose) || (!cam_model && !cam_params && !relpose), OSFM_E_INVALID,
               "osfm_match_pairs_guided: give all of (cam_model, cam_params, relpose) for the calibrated robust stage, or none");
  OSFM_REQUIRE(!(params->flags & (OSFM_MATCH_SQUARED_RATIO | OSFM_MATCH_EXACT_KERNEL)), OSFM_E_UNSUPPORTED,
               "guided matching goes with the BRUTEFORCE matcher (matching.py:272-279)");
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  // float32 bearings in the store's padded row order
  const int64_t rows = (store->tile_off[store->n_images] + 4) * 32;
  std::vector<float> hb((size_t)rows * 3, 0.0f);
  for (int im = 0; im < store->n_images; ++im)
    if (store->counts[im] > 0)
      memcpy(hb.data() + (size_t)store->tile_off[im] * 32 * 3, bearings + (size_t)store->row_off[im] * 3, (size_t)store->counts[im] * 3 * sizeof(float));
  DevBuf d_b;
  OSFM_REQUIRE(d_b.alloc(hb.size() * sizeof(float)) == hipSuccess, OSFM_E_NOMEM, "hipMalloc failed for the bearings");
  OSFM_HIP(hipMemcpy(d_b.p, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
  const OsfmGuidedStage gs{d_b.as<float>(), poses, osfm_guided_cos_threshold(threshold)};
  if (cam_model) {
    double *d_bearings = nullptr;
    int rc = osfm_store_bearings(ctx, store, cam_model, cam_params, &d_bearings);
    if (rc != OSFM_OK) return rc;
    const OsfmCalibStage cs{d_bearings, relpose};
    rc = match_pairs_impl(ctx, store, pairs, n_pairs, params, &cs, out, tm, &gs);
    (void)hipFree(d_bearings);
    return rc;
  }
  return match_pairs_impl(ctx, store, pairs, n_pairs, params, nullptr, out, tm, &gs);
}

extern "C" int64_t osfm_result_num_pairs(const osfm_match_result *r) { return r ? (int64_t)r->counts.size() : 0; }
extern "C" int64_t osfm_result_total_matches(const osfm_match_result *r) {
  return r ? (int64_t)(r->on_device ? r->d_total : r->matches.size() / 2) : 0;
}
extern "C" int osfm_result_fetch(const osfm_match_result *r, int32_t *counts, int32_t *matches) {
  OSFM_REQUIRE(r != nullptr, OSFM_E_INVALID, "osfm_result_fetch: null result");
  if (counts && !r->counts.empty()) memcpy(counts, r->counts.data(), r->counts.size() * sizeof(int32_t));
  if (r->on_device) {  // kept in HBM (OSFM_MATCH_KEEP_DEVICE): one D2H copy on demand
    if (matches && r->d_total > 0) {
      OSFM_HIP(hipSetDevice(r->device));
      OSFM_HIP(hipMemcpy(matches, r->d_matches, r->d_total * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return OSFM_OK;
  }
  if (matches && !r->matches.empty()) memcpy(matches, r->matches.data(), r->matches.size() * sizeof(int32_t));
  return OSFM_OK;
}
extern "C" int osfm_result_host_ptrs(const osfm_match_result *r, const int32_t **counts, const int32_t **matches) {
  OSFM_REQUIRE(r && counts && matches, OSFM_E_INVALID, "osfm_result_host_ptrs: null argument");
  OSFM_REQUIRE(!r->on_device, OSFM_E_INVALID, "osfm_result_host_ptrs: the match rows of this result were kept on the device");
  *counts = r->counts.data();
  *matches = r->matches.data();
  return OSFM_OK;
}
extern "C" int osfm_result_dev_ptrs(const osfm_match_result *r, const int32_t **d_counts, const int32_t **d_matches) {
  OSFM_REQUIRE(r && d_counts && d_matches, OSFM_E_INVALID, "osfm_result_dev_ptrs: null argument");
  OSFM_REQUIRE(r->on_device, OSFM_E_INVALID, "osfm_result_dev_ptrs: the result was not made with OSFM_MATCH_KEEP_DEVICE");
  *d_counts = r->d_counts;
  *d_matches = r->d_matches;
  return OSFM_OK;
}
extern "C" int osfm_result_device(const osfm_match_result *r) { return r ? r->device : -1; }
extern "C" void osfm_result_destroy(osfm_match_result *r) { delete r; }

// Leaf: one pair from host buffers (matching.py:723-777).
extern "C" int osfm_match_l2_ratio(osfm_ctx *ctx, const float *A, int nA, const float *B, int nB, int dim, double ratio,
                                   int symmetric, int32_t *out_pairs, int cap, int *out_n) {
  return osfm_match_l2_ratio_ex(ctx, A, nA, B, nB, dim, ratio, symmetric, 0, out_pairs, cap, out_n);
}

extern "C" int osfm_match_l2_ratio_ex(osfm_ctx *ctx, const float *A, int nA, const float *B, int nB, int dim, double ratio,
                                      int symmetric, int flags, int32_t *out_pairs, int cap, int *out_n) {
  OSFM_REQUIRE(ctx && out_n && (out_pairs || cap == 0), OSFM_E_INVALID, "osfm_match_l2_ratio: null argument");
  // 129 = 128 + the segmentation column (matching_use_segmentation, feature_loading.py:123-155)
  OSFM_REQUIRE(dim == OSFM_DESC_DIM || dim == OSFM_DESC_DIM + 1, OSFM_E_UNSUPPORTED, "descriptor dim %d (128, or 129 with the segmentation column)", dim);
  OSFM_REQUIRE(nA >= 0 && nB >= 0 && (A || nA == 0) && (B || nB == 0), OSFM_E_INVALID, "bad descriptor arrays");
  *out_n = 0;
  OSFM_CTX_LOCK(ctx);
  // knnMatch returns < 2 neighbours when the TRAIN set has < 2 rows -> no match (matching.py:750); a single QUERY is fine in
  // one-way matching (symmetric matching then has an empty second direction).  The batched kernels implement match()'s rule
  // (both images need two features, matching.py:363-374), so a lone query is presented twice and its copy dropped.
  const bool query_is_b = !symmetric && (flags & OSFM_MATCH_SQUARED_RATIO);  // match_flann(index1, f2): f2 queries
  const int n_query = query_is_b ? nB : nA, n_train = query_is_b ? nA : nB;
  if (n_train < 2 || n_query < 1 || (symmetric && n_query < 2)) return OSFM_OK;
  const bool lone = n_query == 1;
  const int mA = (lone && !query_is_b) ? 2 : nA, mB = (lone && query_is_b) ? 2 : nB;
  const int32_t counts[2] = {mA, mB};
  osfm_store *st = nullptr;
  int rc = osfm_store_create(ctx, 2, counts, &st);
  if (rc != OSFM_OK) return rc;
  std::vector<float> desc((size_t)(mA + mB) * OSFM_DESC_DIM);
  std::vector<float> seg(dim > OSFM_DESC_DIM ? (size_t)(mA + mB) : 0);
  for (int r = 0; r < mA; ++r) {
    memcpy(desc.data() + (size_t)r * OSFM_DESC_DIM, A + (size_t)(r < nA ? r : 0) * dim, OSFM_DESC_DIM * sizeof(float));
    if (!seg.empty()) seg[(size_t)r] = A[(size_t)(r < nA ? r : 0) * dim + OSFM_DESC_DIM];
  }
  for (int r = 0; r < mB; ++r) {
    memcpy(desc.data() + (size_t)(mA + r) * OSFM_DESC_DIM, B + (size_t)(r < nB ? r : 0) * dim, OSFM_DESC_DIM * sizeof(float));
    if (!seg.empty()) seg[(size_t)(mA + r)] = B[(size_t)(r < nB ? r : 0) * dim + OSFM_DESC_DIM];
  }
  std::vector<double> pts((size_t)(mA + mB) * 2, 0.0);
  rc = osfm_store_upload_f32(st, desc.data(), pts.data());
  if (rc == OSFM_OK && !seg.empty()) rc = osfm_store_set_segmentation(st, seg.data());
  osfm_match_result *res = nullptr;
  if (rc == OSFM_OK) {
    osfm_match_params prm;
    osfm_match_params_default(&prm);
    prm.lowes_ratio = ratio;
    prm.symmetric = symmetric;
    prm.flags = flags;
    prm.robust = 0;
    const int32_t pair[2] = {0, 1};
    rc = osfm_match_pairs(ctx, st, pair, 1, &prm, &res, nullptr);
  }
  if (rc == OSFM_OK) {
    int n = 0;
    for (int k = 0; k < res->counts[0]; ++k) {
      const int i = res->matches[2 * k], j = res->matches[2 * k + 1];
      if (i >= nA || j >= nB) continue;  // the copy of a lone query
      if (n < cap) {
        out_pairs[2 * n] = i;
        out_pairs[2 * n + 1] = j;
      }
      ++n;
    }
    *out_n = n;
  }
  osfm_result_destroy(res);
  osfm_store_destroy(st);
  return rc;
}

// match_brute_force[_symmetric] on uint8 arrays: cv2 BruteForce-Hamming (matching.py:737-740); same lone-query handling as the L2 leaf
extern "C" int osfm_match_hamming_ratio_ex(osfm_ctx *ctx, const uint8_t *A, int nA, const uint8_t *B, int nB, int width_bytes, double ratio,
                                           int symmetric, int flags, int32_t *out_pairs, int cap, int *out_n) {
  OSFM_REQUIRE(ctx && out_n && (out_pairs || cap == 0), OSFM_E_INVALID, "osfm_match_hamming_ratio: null argument");
  OSFM_REQUIRE(width_bytes >= 1 && width_bytes <= 64, OSFM_E_UNSUPPORTED, "osfm_match_hamming_ratio: %d bytes per descriptor (1..64)", width_bytes);
  OSFM_REQUIRE(nA >= 0 && nB >= 0 && (A || nA == 0) && (B || nB == 0), OSFM_E_INVALID, "bad descriptor arrays");
  *out_n = 0;
  OSFM_CTX_LOCK(ctx);
  // (the lone-query rule and match_flann's query side: as osfm_match_l2_ratio_ex)
  const bool query_is_b = !symmetric && (flags & OSFM_MATCH_SQUARED_RATIO);
  const int n_query = query_is_b ? nB : nA, n_train = query_is_b ? nA : nB;
  if (n_train < 2 || n_query < 1 || (symmetric && n_query < 2)) return OSFM_OK;  // knnMatch returns < 2 neighbours for a train set of < 2 rows
  const bool lone = n_query == 1;
  const int mA = (lone && !query_is_b) ? 2 : nA, mB = (lone && query_is_b) ? 2 : nB;
  const int32_t counts[2] = {mA, mB};
  osfm_store *st = nullptr;
  int rc = osfm_store_create(ctx, 2, counts, &st);
  if (rc != OSFM_OK) return rc;
  std::vector<uint8_t> desc((size_t)(mA + mB) * width_bytes);
  for (int r = 0; r < mA; ++r) memcpy(desc.data() + (size_t)r * width_bytes, A + (size_t)(r < nA ? r : 0) * width_bytes, (size_t)width_bytes);
  for (int r = 0; r < mB; ++r) memcpy(desc.data() + (size_t)(mA + r) * width_bytes, B + (size_t)(r < nB ? r : 0) * width_bytes, (size_t)width_bytes);
  std::vector<double> pts((size_t)(mA + mB) * 2, 0.0);
  rc = osfm_store_upload_binary(st, desc.data(), width_bytes, pts.data());
  osfm_match_result *res = nullptr;
  if (rc == OSFM_OK) {
    osfm_match_params prm;
    osfm_match_params_default(&prm);
    prm.lowes_ratio = ratio;
    prm.symmetric = symmetric;
    prm.flags = flags;
    prm.robust = 0;
    const int32_t pair[2] = {0, 1};
    rc = osfm_match_pairs(ctx, st, pair, 1, &prm, &res, nullptr);
  }
  if (rc == OSFM_OK) {
    int n = 0;
    for (int k = 0; k < res->counts[0]; ++k) {
      const int i = res->matches[2 * k], j = res->matches[2 * k + 1];
      if (i >= nA || j >= nB) continue;  // the copy of a lone query
      if (n < cap) {
        out_pairs[2 * n] = i;
        out_pairs[2 * n + 1] = j;
      }
      ++n;
    }
    *out_n = n;
  }
  osfm_result_destroy(res);
  osfm_store_destroy(st);
  return rc;
}

extern "C" int osfm_match_hamming_ratio(osfm_ctx *ctx, const uint8_t *A, int nA, const uint8_t *B, int nB, int width_bytes, double ratio,
                                        int symmetric, int32_t *out_pairs, int cap, int *out_n) {
  return osfm_match_hamming_ratio_ex(ctx, A, nA, B, nB, width_bytes, ratio, symmetric, 0, out_pairs, cap, out_n);
}
