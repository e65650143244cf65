// tracks.hip -- linking matches into tracks on the GPU.
//
// Replaces tracking.create_tracks_manager's grouping (opensfm/tracking.py:68-98): a union-find over
// every match (im1, f1) -- (im2, f2), the sets listed in the order of their first-inserted member,
// members in insertion order, then _good_track (tracking.py:238-244): at least min_length members
// and no image twice.  track_id = index in that filtered list.
//
// GPU formulation (no sequential union-find):
//   rank[node]  = position of the node's first appearance in the flattened edge list a0 b0 a1 b1 ...
//                 (atomicMin; unique per node) -- this IS the reference's dict insertion order;
//   components  = lock-free hooking (ECL-CC style: CAS the larger root under the smaller one, the
//                 order being `rank`), so every component's root is its member of minimum rank;
//   tracks      = one stable 64-bit radix sort by (rank of the root, rank): components come out in
//                 the order of their first-inserted member, members in insertion order;
//   _good_track = a second sort by (root rank, image) exposes repeated images as equal neighbours.
// Everything else is scans (rocPRIM).  Bit-identical to oracle/tracks_oracle.c.
#include <cstring>  // rocprim's texture iterator calls the host memset
#include <memory>

#include <rocprim/rocprim.hpp>

#include <vector>

#include "osfm_internal.h"

namespace {
constexpr unsigned kNoRank = 0xFFFFFFFFu;
constexpr int TPB = 256;
inline unsigned nblk(long n) { return (unsigned)((n + TPB - 1) / TPB); }

__global__ void fill_u32_kernel(unsigned *p, long n, unsigned v) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void iota_kernel(int *p, long n) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) p[i] = (int)i;
}
__global__ void rank_kernel(const int *ea, const int *eb, long E, unsigned *rank) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= E) return;
  atomicMin(&rank[ea[i]], (unsigned)(2 * i));
  atomicMin(&rank[eb[i]], (unsigned)(2 * i + 1));
}
__device__ __forceinline__ int representative(int idx, int *parent) {
  int curr = parent[idx];
  if (curr != idx) {
    int next, prev = idx;
    while (curr != (next = parent[curr])) {  // pointer jumping on the way up
      parent[prev] = next;
      prev = curr;
      curr = next;
    }
  }
  return curr;
}
__global__ void hook_kernel(const int *ea, const int *eb, long E, const unsigned *rank, int *parent) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= E) return;
  int u = representative(ea[i], parent), v = representative(eb[i], parent);
  bool repeat;
  do {
    repeat = false;
    if (u != v) {
      int ret;
      if (rank[v] < rank[u]) {  // hook the larger (later-inserted) root under the smaller one
        if ((ret = atomicCAS(&parent[u], u, v)) != u) {
          u = ret;
          repeat = true;
        }
      } else {
        if ((ret = atomicCAS(&parent[v], v, u)) != v) {
          v = ret;
          repeat = true;
        }
      }
    }
  } while (repeat);
}
__global__ void key_kernel(const unsigned *rank, int *parent, long N, unsigned long long *keys, int *count) {
  const long v = (long)blockIdx.x * TPB + threadIdx.x;
  if (v >= N) return;
  unsigned long long k = ~0ull;
  if (rank[v] != kNoRank) {
    int r = parent[v];
    while (r != parent[r]) r = parent[r];
    k = ((unsigned long long)rank[r] << 32) | rank[v];
    atomicAdd(count, 1);
  }
  keys[v] = k;
}
__global__ void head_kernel(const unsigned long long *keys, long Na, int *head) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= Na) return;
  head[i] = (i == 0 || (keys[i] >> 32) != (keys[i - 1] >> 32)) ? 1 : 0;
}
__global__ void segstart_kernel(const int *head, const int *segincl, long Na, int *segstart) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= Na) return;
  if (head[i]) segstart[segincl[i] - 1] = (int)i;
}
__device__ __forceinline__ int image_of(const long long *off, int n_images, long node) {
  int lo = 0, hi = n_images;  // off[lo] <= node < off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= node) lo = mid;
    else hi = mid;
  }
  return lo;
}
__global__ void key2_kernel(const unsigned long long *keys, const int *nodes, long Na, const long long *off, int n_images,
                            unsigned long long *keys2) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= Na) return;
  keys2[i] = (keys[i] & 0xFFFFFFFF00000000ull) | (unsigned)image_of(off, n_images, nodes[i]);
}
__global__ void dup_kernel(const unsigned long long *keys2, const int *segincl, long Na, int *bad) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i + 1 >= Na) return;
  if (keys2[i] == keys2[i + 1]) bad[segincl[i] - 1] = 1;
}
__global__ void good_kernel(const int *segstart, const int *bad, int nseg, long Na, int min_length, int *good) {
  const int s = blockIdx.x * TPB + threadIdx.x;
  if (s >= nseg) return;
  const long len = (s + 1 < nseg ? segstart[s + 1] : Na) - segstart[s];
  good[s] = (len >= min_length && !bad[s]) ? 1 : 0;
}
__global__ void obsflag_kernel(const int *segincl, const int *good, long Na, int *flag) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= Na) return;
  flag[i] = good[segincl[i] - 1];
}
__global__ void emit_kernel(const int *segincl, const int *good, const int *trackid, const int *obsidx, const int *nodes, long Na,
                            const long long *off, int n_images, int *o_track, int *o_image, int *o_feature) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= Na) return;
  const int s = segincl[i] - 1;
  if (!good[s]) return;
  const int k = obsidx[i], node = nodes[i];
  const int im = image_of(off, n_images, node);
  o_track[k] = trackid[s];
  o_image[k] = im;
  o_feature[k] = (int)(node - off[im]);
}

struct DevBuf {
  std::vector<void *> ptrs;
  hipError_t err = hipSuccess;
  template <class T>
  T *alloc(size_t n) {
    void *p = nullptr;
    if (err == hipSuccess) err = hipMalloc(&p, (n ? n : 1) * sizeof(T));
    if (p) ptrs.push_back(p);
    return (T *)p;
  }
  ~DevBuf() {
    for (void *p : ptrs) (void)hipFree(p);
  }
};
}  // namespace

struct osfm_tracks {
  int64_t n_tracks = 0, n_obs = 0;
  std::vector<int32_t> track, image, feature;
  double ms_device = 0;
};

extern "C" int osfm_tracks_create(osfm_ctx *ctx, const int32_t *edge_a, const int32_t *edge_b, int64_t n_edges,
                                  const int64_t *node_offsets, int32_t n_images, int32_t min_length, osfm_tracks **out) {
  OSFM_REQUIRE(ctx && out && node_offsets && n_images >= 0 && n_edges >= 0 && (n_edges == 0 || (edge_a && edge_b)), OSFM_E_INVALID,
               "osfm_tracks_create: bad argument");
  *out = nullptr;
  OSFM_REQUIRE(n_edges < (1ll << 31) - 1, OSFM_E_UNSUPPORTED, "more than 2^31 matches");
  const long N = node_offsets[n_images], E = n_edges;
  OSFM_REQUIRE(N >= 0 && N < (1ll << 31), OSFM_E_UNSUPPORTED, "more than 2^31 features");
  for (int64_t i = 0; i < E; ++i)
    OSFM_REQUIRE(edge_a[i] >= 0 && edge_a[i] < N && edge_b[i] >= 0 && edge_b[i] < N, OSFM_E_INVALID, "match %lld references node %d / %d",
                 (long long)i, edge_a[i], edge_b[i]);
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // the result object is handed to the caller only on success: no leak when a later step fails
  std::unique_ptr<osfm_tracks> owner(new osfm_tracks());
  osfm_tracks *T = owner.get();
  if (E == 0 || N == 0) {  // no images or no matches: an empty table, as tracking.create_tracks_manager returns an empty manager
    *out = owner.release();
    return OSFM_OK;
  }
  DevBuf B;
  int *ea = B.alloc<int>(E), *eb = B.alloc<int>(E);
  long long *off = B.alloc<long long>((size_t)n_images + 1);
  unsigned *rank = B.alloc<unsigned>(N);
  int *parent = B.alloc<int>(N), *nodes_in = B.alloc<int>(N), *nodes = B.alloc<int>(N);
  unsigned long long *keys_in = B.alloc<unsigned long long>(N), *keys = B.alloc<unsigned long long>(N);
  unsigned long long *keys2_in = B.alloc<unsigned long long>(N), *keys2 = B.alloc<unsigned long long>(N);
  int *head = B.alloc<int>(N), *segincl = B.alloc<int>(N), *segstart = B.alloc<int>(N), *bad = B.alloc<int>(N);
  int *good = B.alloc<int>(N), *trackid = B.alloc<int>(N), *flag = B.alloc<int>(N), *obsidx = B.alloc<int>(N);
  int *o_track = B.alloc<int>(N), *o_image = B.alloc<int>(N), *o_feature = B.alloc<int>(N);
  int *counter = B.alloc<int>(4);
  size_t tmp_bytes = 0, need = 0;
  (void)rocprim::radix_sort_pairs(nullptr, need, keys_in, keys, nodes_in, nodes, (size_t)N, 0u, 64u, st);
  tmp_bytes = need;
  (void)rocprim::radix_sort_keys(nullptr, need, keys2_in, keys2, (size_t)N, 0u, 64u, st);
  tmp_bytes = std::max(tmp_bytes, need);
  (void)rocprim::inclusive_scan(nullptr, need, head, segincl, (size_t)N, rocprim::plus<int>(), st);
  tmp_bytes = std::max(tmp_bytes, need);
  (void)rocprim::exclusive_scan(nullptr, need, flag, obsidx, 0, (size_t)N, rocprim::plus<int>(), st);
  tmp_bytes = std::max(tmp_bytes, need);
  unsigned char *tmp = B.alloc<unsigned char>(tmp_bytes + 256);
  OSFM_REQUIRE(B.err == hipSuccess, OSFM_E_NOMEM, "tracks: device allocation failed: %s", hipGetErrorString(B.err));
  OSFM_HIP(hipMemcpyAsync(ea, edge_a, (size_t)E * 4, hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(eb, edge_b, (size_t)E * 4, hipMemcpyHostToDevice, st));
  OSFM_HIP(hipMemcpyAsync(off, node_offsets, ((size_t)n_images + 1) * 8, hipMemcpyHostToDevice, st));
  OSFM_HIP(hipEventRecord(ctx->ev[4], st));
  hipLaunchKernelGGL(fill_u32_kernel, dim3(nblk(N)), dim3(TPB), 0, st, rank, N, kNoRank);
  hipLaunchKernelGGL(iota_kernel, dim3(nblk(N)), dim3(TPB), 0, st, parent, N);
  hipLaunchKernelGGL(iota_kernel, dim3(nblk(N)), dim3(TPB), 0, st, nodes_in, N);
  OSFM_HIP(hipMemsetAsync(counter, 0, 16, st));
  hipLaunchKernelGGL(rank_kernel, dim3(nblk(E)), dim3(TPB), 0, st, ea, eb, E, rank);
  hipLaunchKernelGGL(hook_kernel, dim3(nblk(E)), dim3(TPB), 0, st, ea, eb, E, rank, parent);
  hipLaunchKernelGGL(key_kernel, dim3(nblk(N)), dim3(TPB), 0, st, rank, parent, N, keys_in, counter);
  size_t tb = tmp_bytes;
  OSFM_HIP(rocprim::radix_sort_pairs(tmp, tb, keys_in, keys, nodes_in, nodes, (size_t)N, 0u, 64u, st));
  int h_count[4] = {0, 0, 0, 0};
  OSFM_HIP(hipMemcpyAsync(h_count, counter, 4, hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  const long Na = h_count[0];  // nodes that appear in a match, now the first Na sorted entries
  if (Na == 0) {
    *out = owner.release();
    return OSFM_OK;
  }
  hipLaunchKernelGGL(head_kernel, dim3(nblk(Na)), dim3(TPB), 0, st, keys, Na, head);
  tb = tmp_bytes;
  OSFM_HIP(rocprim::inclusive_scan(tmp, tb, head, segincl, (size_t)Na, rocprim::plus<int>(), st));
  int nseg = 0;
  OSFM_HIP(hipMemcpyAsync(&nseg, segincl + (Na - 1), 4, hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  hipLaunchKernelGGL(segstart_kernel, dim3(nblk(Na)), dim3(TPB), 0, st, head, segincl, Na, segstart);
  hipLaunchKernelGGL(key2_kernel, dim3(nblk(Na)), dim3(TPB), 0, st, keys, nodes, Na, off, n_images, keys2_in);
  tb = tmp_bytes;
  OSFM_HIP(rocprim::radix_sort_keys(tmp, tb, keys2_in, keys2, (size_t)Na, 0u, 64u, st));
  OSFM_HIP(hipMemsetAsync(bad, 0, (size_t)nseg * 4, st));
  hipLaunchKernelGGL(dup_kernel, dim3(nblk(Na)), dim3(TPB), 0, st, keys2, segincl, Na, bad);
  hipLaunchKernelGGL(good_kernel, dim3(nblk(nseg)), dim3(TPB), 0, st, segstart, bad, nseg, Na, min_length, good);
  tb = tmp_bytes;
  OSFM_HIP(rocprim::exclusive_scan(tmp, tb, good, trackid, 0, (size_t)nseg, rocprim::plus<int>(), st));
  hipLaunchKernelGGL(obsflag_kernel, dim3(nblk(Na)), dim3(TPB), 0, st, segincl, good, Na, flag);
  tb = tmp_bytes;
  OSFM_HIP(rocprim::exclusive_scan(tmp, tb, flag, obsidx, 0, (size_t)Na, rocprim::plus<int>(), st));
  int last[4] = {0, 0, 0, 0};  // trackid[nseg-1], good[nseg-1], obsidx[Na-1], flag[Na-1]
  OSFM_HIP(hipMemcpyAsync(&last[0], trackid + (nseg - 1), 4, hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipMemcpyAsync(&last[1], good + (nseg - 1), 4, hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipMemcpyAsync(&last[2], obsidx + (Na - 1), 4, hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipMemcpyAsync(&last[3], flag + (Na - 1), 4, hipMemcpyDeviceToHost, st));
  hipLaunchKernelGGL(emit_kernel, dim3(nblk(Na)), dim3(TPB), 0, st, segincl, good, trackid, obsidx, nodes, Na, off, n_images, o_track,
                     o_image, o_feature);
  OSFM_HIP(hipEventRecord(ctx->ev[5], st));
  OSFM_HIP(hipStreamSynchronize(st));
  T->n_tracks = (int64_t)last[0] + last[1];
  T->n_obs = (int64_t)last[2] + last[3];
  T->track.resize((size_t)T->n_obs);
  T->image.resize((size_t)T->n_obs);
  T->feature.resize((size_t)T->n_obs);
  if (T->n_obs > 0) {
    OSFM_HIP(hipMemcpyAsync(T->track.data(), o_track, (size_t)T->n_obs * 4, hipMemcpyDeviceToHost, st));
    OSFM_HIP(hipMemcpyAsync(T->image.data(), o_image, (size_t)T->n_obs * 4, hipMemcpyDeviceToHost, st));
    OSFM_HIP(hipMemcpyAsync(T->feature.data(), o_feature, (size_t)T->n_obs * 4, hipMemcpyDeviceToHost, st));
    OSFM_HIP(hipStreamSynchronize(st));
  }
  float ms = 0.f;
  OSFM_HIP(hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5]));
  T->ms_device = ms;
  *out = owner.release();
  return OSFM_OK;
}
extern "C" int64_t osfm_tracks_num_tracks(const osfm_tracks *t) { return t ? t->n_tracks : 0; }
extern "C" int64_t osfm_tracks_num_observations(const osfm_tracks *t) { return t ? t->n_obs : 0; }
extern "C" double osfm_tracks_device_ms(const osfm_tracks *t) { return t ? t->ms_device : 0.0; }
extern "C" int osfm_tracks_fetch(const osfm_tracks *t, int32_t *obs_track, int32_t *obs_image, int32_t *obs_feature) {
  OSFM_REQUIRE(t && (t->n_obs == 0 || (obs_track && obs_image && obs_feature)), OSFM_E_INVALID, "osfm_tracks_fetch: null argument");
  for (int64_t i = 0; i < t->n_obs; ++i) {
    obs_track[i] = t->track[(size_t)i];
    obs_image[i] = t->image[(size_t)i];
    obs_feature[i] = t->feature[(size_t)i];
  }
  return OSFM_OK;
}
extern "C" void osfm_tracks_destroy(osfm_tracks *t) { delete t; }
