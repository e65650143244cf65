// hip_runtime.h -- a HOST EMULATION of the small part of HIP that opensfm_amd/csrc/ba*.hip use: TEST INFRASTRUCTURE ONLY.
//
// tests/native/build_emu.py compiles the product's own bundle-adjustment sources (ba.hip and its .inc files, unmodified apart from three
// mechanical substitutions listed there) against this header with the host clang++, so that every kernel, every launch and the whole
// Levenberg-Marquardt driver run on the CPU and can be compared with the oracle in the `-m "not gpu"` suite -- including under
// AddressSanitizer.  Nothing in opensfm_amd/ ever includes it; the GPU library is built by hipcc against the real runtime.
//
// Execution model: a launch runs its workgroups one after the other; the threads of a workgroup are ucontext fibres on one OS thread.
// A fibre runs until it reaches a synchronisation point (__syncthreads, a wavefront shuffle, an MFMA, a wavefront barrier) or returns.
// Wavefronts (64 consecutive threads) are swept lane by lane: after a sweep every lane of the wavefront stands at a synchronisation
// point, the wavefront collectives are resolved from what the lanes posted, and the sweep repeats until the whole wavefront waits at
// the workgroup barrier (or has returned).  Global and LDS memory are ordinary host memory: __shared__ becomes `static` (workgroups never
// overlap in time), dynamic LDS a buffer that is poisoned with NaNs before every workgroup, hipMalloc'd memory is poisoned as well, so a
// read of something never written shows up as a NaN instead of a lucky zero.
#pragma once
#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 {
  double x, y;
};
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct int2 {
  int x, y;
};
struct uint4 {
  unsigned x, y, z, w;
};
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

using std::max;
using std::min;

typedef struct ihipStream_t *hipStream_t;
typedef struct ihipEvent_t *hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
  int multiProcessorCount = 256;
  char name[64] = "hipemu";
};

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipErrorInvalidValue"; }
static inline hipError_t hipMalloc(void **p, size_t n) {
  const size_t bytes = (n + 255) / 256 * 256 + 256;
  *p = aligned_alloc(256, bytes);
  if (!*p) return hipErrorOutOfMemory;
  memset(*p, 0xFF, bytes);  // NaN / -1: an unwritten entry must not look like a zero
  return hipSuccess;
}
template <class T>
static inline hipError_t hipMalloc(T **p, size_t n) {
  return hipMalloc((void **)p, n);
}
static inline hipError_t hipFree(void *p) {
  free(p);
  return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
  *free_b = *total_b = (size_t)64 << 30;
  return hipSuccess;
}
constexpr unsigned hipHostMallocDefault = 0;
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { return hipFree(p); }
static inline void __threadfence_system() {}
#ifndef __HIP_MEMORY_SCOPE_SYSTEM
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) {
  *d = h;
  return hipSuccess;
}
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
  if (n) memmove(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t = nullptr) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipMemset(void *d, int v, size_t n) {
  if (n) memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { return hipMemset(d, v, n); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) {
  *s = (hipStream_t)malloc(8);
  return hipSuccess;
}
static inline hipError_t hipStreamDestroy(hipStream_t s) {
  free(s);
  return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) {
  *e = (hipEvent_t)malloc(8);
  return hipSuccess;
}
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) {
  free(e);
  return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
  *ms = 1.0f;
  return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) {
  *n = 1;
  return hipSuccess;
}
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  *p = hipDeviceProp_t();
  return hipSuccess;
}
static inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }

namespace hipemu {
enum { ST_RUN = 0, ST_BAR = 1, ST_WAVE = 2, ST_DONE = 3 };
enum { OP_NONE = 0, OP_SHFL = 1, OP_MFMA_F64_16X16X4 = 2, OP_WAVE_BARRIER = 3 };
struct Lane {
  ucontext_t ctx;
  int state, op, src;
  dim3 tid;
  uint64_t post, result;
  double ma, mb, mc[4], md[4];
};
constexpr size_t kStackBytes = 512 * 1024;
constexpr int kMaxThreads = 1024;
inline Lane *g_lanes = nullptr;
inline char *g_stacks = nullptr;
inline Lane *g_cur = nullptr;
inline ucontext_t g_sched;
inline dim3 g_tid, g_bid, g_bdim, g_gdim;
inline std::vector<char> g_dyn;
inline const std::function<void()> *g_body = nullptr;
inline long g_launches = 0;

inline void *dyn_lds() { return g_dyn.data(); }
inline void yield(int st) {
  Lane *me = g_cur;
  me->state = st;
  swapcontext(&me->ctx, &g_sched);
}
inline void fiber_entry() {
  (*g_body)();
  g_cur->state = ST_DONE;
  swapcontext(&g_cur->ctx, &g_sched);
}
inline void resolve_wave(Lane *w, int n) {
  bool mfma = false;
  for (int l = 0; l < n; l++)
    if (w[l].state == ST_WAVE && w[l].op == OP_MFMA_F64_16X16X4) mfma = true;
  if (mfma) {  // D (16 x 16) = A (16 x 4) B (4 x 16) + C; lane l holds A[l % 16][l / 16], B[l / 16][l % 16] and, in register r, C / D[4 r + l / 16][l % 16]
    double A[16][4], B[4][16];
    for (int l = 0; l < 64; l++) {
      const bool on = l < n && w[l].state == ST_WAVE && w[l].op == OP_MFMA_F64_16X16X4;
      A[l % 16][l / 16] = on ? w[l].ma : 0.0;
      B[l / 16][l % 16] = on ? w[l].mb : 0.0;
    }
    for (int l = 0; l < n; l++) {
      if (!(w[l].state == ST_WAVE && w[l].op == OP_MFMA_F64_16X16X4)) continue;
      for (int r = 0; r < 4; r++) {
        const int row = 4 * r + l / 16, col = l % 16;
        double acc = w[l].mc[r];
        for (int k = 0; k < 4; k++) acc = __builtin_fma(A[row][k], B[k][col], acc);
        w[l].md[r] = acc;
      }
    }
  }
  for (int l = 0; l < n; l++)
    if (w[l].state == ST_WAVE && w[l].op == OP_SHFL) {
      const int s = w[l].src;
      w[l].result = (s >= 0 && s < n && w[s].state == ST_WAVE && w[s].op == OP_SHFL) ? w[s].post : w[l].post;
    }
  for (int l = 0; l < n; l++)
    if (w[l].state == ST_WAVE) {
      w[l].state = ST_RUN;
      w[l].op = OP_NONE;
    }
}
inline void run_block(int nthreads) {
  for (int t = 0; t < nthreads; t++) {
    Lane &L = g_lanes[t];
    getcontext(&L.ctx);
    L.ctx.uc_stack.ss_sp = g_stacks + (size_t)t * kStackBytes;
    L.ctx.uc_stack.ss_size = kStackBytes;
    L.ctx.uc_link = nullptr;
    makecontext(&L.ctx, (void (*)())fiber_entry, 0);
    L.state = ST_RUN;
    L.op = OP_NONE;
    L.tid = dim3(t % g_bdim.x, (t / g_bdim.x) % g_bdim.y, t / (g_bdim.x * g_bdim.y));
  }
  const int nw = (nthreads + 63) / 64;
  for (;;) {
    for (int w = 0; w < nw; w++) {
      Lane *wl = g_lanes + 64 * w;
      const int n = std::min(64, nthreads - 64 * w);
      for (;;) {
        for (int l = 0; l < n; l++)
          if (wl[l].state == ST_RUN) {
            g_cur = &wl[l];
            g_tid = wl[l].tid;
            swapcontext(&g_sched, &wl[l].ctx);
          }
        bool coll = false;
        for (int l = 0; l < n; l++) coll = coll || wl[l].state == ST_WAVE;
        if (!coll) break;
        resolve_wave(wl, n);
      }
    }
    int waiting = 0;
    for (int t = 0; t < nthreads; t++)
      if (g_lanes[t].state == ST_BAR) {
        g_lanes[t].state = ST_RUN;
        waiting++;
      }
    if (!waiting) break;
  }
}
template <class F>
inline void launch(dim3 grid, dim3 block, size_t lds, F f) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads <= 0 || nthreads > kMaxThreads || grid.x * (size_t)grid.y * grid.z == 0) {
    fprintf(stderr, "hipemu: bad launch configuration (%u %u %u) x (%u %u %u)\n", grid.x, grid.y, grid.z, block.x, block.y, block.z);
    abort();
  }
  if (!g_lanes) {
    g_lanes = new Lane[kMaxThreads];
    g_stacks = (char *)mmap(nullptr, kStackBytes * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (g_stacks == (char *)MAP_FAILED) abort();
  }
  const std::function<void()> body = f;
  g_body = &body;
  g_bdim = block;
  g_gdim = grid;
  g_dyn.resize(lds + 64);
  g_launches++;
  for (unsigned z = 0; z < grid.z; z++)
    for (unsigned y = 0; y < grid.y; y++)
      for (unsigned x = 0; x < grid.x; x++) {
        g_bid = dim3(x, y, z);
        memset(g_dyn.data(), 0xFF, g_dyn.size());
        run_block(nthreads);
      }
  g_body = nullptr;
}
template <class T>
inline T shfl(T v, int src) {
  static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
  Lane *me = g_cur;
  me->post = 0;
  memcpy(&me->post, &v, sizeof(T));
  me->op = OP_SHFL;
  me->src = src;
  yield(ST_WAVE);
  T out;
  memcpy(&out, &me->result, sizeof(T));
  return out;
}
inline int lane_id() { return (int)((g_tid.x + g_bdim.x * (g_tid.y + g_bdim.y * g_tid.z)) & 63); }
}  // namespace hipemu

#define threadIdx (hipemu::g_tid)
#define blockIdx (hipemu::g_bid)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hipemu::launch(dim3(grid), dim3(block), (size_t)(lds), [=]() { (kernel)(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::yield(hipemu::ST_BAR); }
template <class T>
static inline T __shfl_xor(T v, int mask, int = 64) { return hipemu::shfl(v, hipemu::lane_id() ^ mask); }
template <class T>
static inline T __shfl(T v, int src, int = 64) { return hipemu::shfl(v, src); }
template <class T>
static inline T __shfl_down(T v, unsigned delta, int = 64) { return hipemu::shfl(v, hipemu::lane_id() + (int)delta); }
template <class T>
static inline T __shfl_up(T v, unsigned delta, int = 64) { return hipemu::shfl(v, hipemu::lane_id() - (int)delta); }
static inline void hipemu_wave_barrier() {
  hipemu::g_cur->op = hipemu::OP_WAVE_BARRIER;
  hipemu::yield(hipemu::ST_WAVE);
}
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_barrier() __syncthreads()
typedef double hipemu_v4d __attribute__((ext_vector_type(4)));
static inline hipemu_v4d hipemu_mfma_f64_16x16x4(double a, double b, hipemu_v4d c, int, int, int) {
  hipemu::Lane *me = hipemu::g_cur;
  me->ma = a;
  me->mb = b;
  for (int r = 0; r < 4; r++) me->mc[r] = c[r];
  me->op = hipemu::OP_MFMA_F64_16X16X4;
  hipemu::yield(hipemu::ST_WAVE);
  hipemu_v4d d;
  for (int r = 0; r < 4; r++) d[r] = me->md[r];
  return d;
}
#define __builtin_amdgcn_mfma_f64_16x16x4f64 hipemu_mfma_f64_16x16x4

template <class T>
static inline T atomicAdd(T *p, T v) {
  const T old = *p;
  *p = old + v;
  return old;
}
static inline double atomicAdd(double *p, double v) {
  const double old = *p;
  *p = old + v;
  return old;
}
template <class T>
static inline T atomicMax(T *p, T v) {
  const T old = *p;
  if (v > old) *p = v;
  return old;
}
template <class T>
static inline T atomicMin(T *p, T v) {
  const T old = *p;
  if (v < old) *p = v;
  return old;
}
template <class T>
static inline T atomicOr(T *p, T v) {
  const T old = *p;
  *p = old | v;
  return old;
}
template <class T>
static inline T atomicCAS(T *p, T cmp, T v) {
  const T old = *p;
  if (old == cmp) *p = v;
  return old;
}
static inline long long __double_as_longlong(double v) {
  long long o;
  memcpy(&o, &v, 8);
  return o;
}
static inline double __longlong_as_double(long long v) {
  double o;
  memcpy(&o, &v, 8);
  return o;
}
