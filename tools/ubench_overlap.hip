// Microbenchmark: do int8 MFMAs and integer VALU ops of the SAME or of DIFFERENT wavefronts overlap
// on a gfx950 SIMD?  Each workgroup = 256 threads (one wave per SIMD); blocks-per-CU 1 or 2.
//   mode 0: MFMA only (4 x 32x32x32 i8 per iteration, two independent accumulators)
//   mode 1: VALU only (V x {v_lshl_add_u32, v_max3_i32} per iteration, 8 independent chains)
//   mode 2: both in the same wave
//   mode 3: even waves (by block) MFMA only, odd blocks VALU only  (needs 2 blocks / CU)
// build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/ubench_overlap.hip -o tools/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE, int V>
__global__ void __launch_bounds__(256, 2) k(int *out, int iters, int seed) {
  __shared__ v4i ldsb[2 * 4 * 64];
  if (MODE >= 6) {
    for (int i = threadIdx.x; i < 2 * 4 * 64; i += 256) ldsb[i] = v4i{seed + i, seed, i, 1};
    __syncthreads();
  }
  v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed * 5, seed * 7, seed * 9};
  v16i acc0 = {0}, acc1 = {0};
  int x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * (i + 1) + seed;
  int c0 = seed * 11, c1 = seed * 13;
  const bool do_m = MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7 || (MODE == 3 && (blockIdx.x & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6 || (MODE == 3 && (blockIdx.x & 1) == 1);
  for (int it = 0; it < iters; it++) {
    if (do_m && MODE >= 6) {  // B operands come from LDS every iteration, accumulators restart from zero (like a tile)
      const int lane = threadIdx.x & 63;
      v4i b0[4], b1[4];
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        b0[ks] = ldsb[(ks * 64 + lane + it) & 511];
        b1[ks] = ldsb[(256 + ks * 64 + lane + it) & 511];
      }
      v16i z = {0};
      acc0 = z;
      acc1 = z;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1[ks], acc1, 0, 0, 0);
      }
      x[0] += b0[2][0] + b0[3][1] + b1[2][2] + b1[3][3];
    } else if (do_m) {
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc1, 0, 0, 0);
    }
    if (do_v && MODE == 6) {  // the matcher's epilogue shape: keys from the accumulators just produced
#pragma unroll
      for (int q = 0; q < V / 16; q++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          int t;
          asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"((q & 1) ? acc1[(q * 8 + i) & 15] : acc0[(q * 8 + i) & 15]), "v"(c0));
          asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(t), "v"(c1));
        }
      }
    } else if (do_v && MODE == 4) {  // epilogue-like: keys built from the accumulator of the PREVIOUS iteration's MFMAs
      v16i prev = acc1;
#pragma unroll
      for (int q = 0; q < V / 16; q++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          int t;
          asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(t) : "v"(prev[(q * 8 + i) & 15]), "v"(c0));
          asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(t), "v"(c1));
        }
      }
    } else if (do_v && MODE == 5) {  // keys from the accumulator just produced (true dependency)
#pragma unroll
      for (int q = 0; q < V / 16; q++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          int t;
          asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(t) : "v"(acc0[(q * 8 + i) & 15]), "v"(c0));
          asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(t), "v"(c1));
        }
      }
    } else if (do_v) {
#pragma unroll
      for (int q = 0; q < V / 16; q++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          int t;
          asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(t) : "v"(x[i]), "v"(c0));
          asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(t), "v"(c1));
        }
      }
    }
  }
  int r = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) r += acc0[i] + acc1[i];
#pragma unroll
  for (int i = 0; i < 8; i++) r += x[i];
  if (r == 0x12345678) out[threadIdx.x] = r;
}

// Software-pipelined variant of mode 6: the LDS reads of iteration it+1 are issued first, the 4 MFMAs of
// iteration it (operands loaded one iteration earlier) are interleaved one by one with 12 VALU ops of the
// epilogue of iteration it-1, pinned with sched_barrier.
__global__ void __launch_bounds__(256, 2) kpipe(int *out, int iters, int seed) {
  __shared__ v4i ldsb[2 * 4 * 64];
  for (int i = threadIdx.x; i < 2 * 4 * 64; i += 256) ldsb[i] = v4i{seed + i, seed, i, 1};
  __syncthreads();
  const int lane = threadIdx.x & 63;
  v4i a = {seed, seed + 1, seed + 2, seed + 3};
  int x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * (i + 1) + seed;
  const int c0 = seed * 11, c1 = seed * 13;
  v4i bc0[2], bc1[2], bn0[2], bn1[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    bc0[ks] = ldsb[(ks * 64 + lane) & 511];
    bc1[ks] = ldsb[(256 + ks * 64 + lane) & 511];
  }
  v16i accP0 = {0}, accP1 = {0}, accN0, accN1;
  // one pipeline stage, operands and accumulators passed by name so that the two-fold unrolled loop
  // below ping-pongs between two register sets without copies
#define STAGE(BC0, BC1, BN0, BN1, AP0, AP1, AN0, AN1, IT)                                             \
  {                                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ks++) {                                                 \
      BN0[ks] = ldsb[(ks * 64 + lane + (IT) + 1) & 511];                                               \
      BN1[ks] = ldsb[(256 + ks * 64 + lane + (IT) + 1) & 511];                                         \
    }                                                                                                  \
    const v16i z16 = {0};                                                                              \
    AN0 = z16;                                                                                         \
    AN1 = z16;                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                    \
      if (i & 1) AN1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, BC1[i >> 1], AN1, 0, 0, 0);            \
      else AN0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, BC0[i >> 1], AN0, 0, 0, 0);                  \
      __builtin_amdgcn_sched_barrier(0);                                                               \
      _Pragma("unroll") for (int j = 0; j < 6; j++) {                                                  \
        const int e = i * 6 + j;                                                                       \
        int t;                                                                                         \
        asm volatile("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"((e & 1) ? AP1[e & 15] : AP0[e & 15]), "v"(c0)); \
        asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(x[e & 7]) : "v"(x[e & 7]), "v"(t), "v"(c1)); \
      }                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }                                                                                                  \
  }
  for (int it = 0; it < iters; it += 2) {
    STAGE(bc0, bc1, bn0, bn1, accP0, accP1, accN0, accN1, it)
    STAGE(bn0, bn1, bc0, bc1, accN0, accN1, accP0, accP1, it + 1)
  }
#undef STAGE
  int r = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) r += accP0[i] + accP1[i];
#pragma unroll
  for (int i = 0; i < 8; i++) r += x[i];
  if (r == 0x12345678) out[threadIdx.x] = r;
}
float run_pipe(int blocks, int iters) {
  int *d;
  hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kpipe, dim3(blocks), dim3(256), 0, 0, d, iters / 10, 3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kpipe, dim3(blocks), dim3(256), 0, 0, d, iters, 3);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms;
}

template <int MODE, int V>
float run(int blocks, int iters) {
  int *d;
  hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, V>), dim3(blocks), dim3(256), 0, 0, d, iters / 10, 3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, V>), dim3(blocks), dim3(256), 0, 0, d, iters, 3);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(d);
  return ms;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int iters = 200000;
  printf("CUs %d, clock %d kHz\n", cus, p.clockRate);
  for (int bpc = 1; bpc <= 2; bpc++) {
    const int blocks = cus * bpc;
    const float m = run<0, 48>(blocks, iters), v48 = run<1, 48>(blocks, iters), v96 = run<1, 96>(blocks, iters);
    const float b48 = run<2, 48>(blocks, iters), b96 = run<2, 96>(blocks, iters);
    printf("waves/SIMD %d: MFMA-only %.2f ms | VALU48 %.2f | VALU96 %.2f | same-wave MFMA+VALU48 %.2f | +VALU96 %.2f\n", bpc, m, v48,
           v96, b48, b96);
    const double cyc = 1e-3 * p.clockRate * 1e3 / iters;  // cycles per ms per iteration
    printf("   cycles/iter/wave-slot: MFMA(4) %.0f  VALU48 %.0f  VALU96 %.0f  both48 %.0f  both96 %.0f\n", m * cyc / bpc * bpc, v48 * cyc,
           v96 * cyc, b48 * cyc, b96 * cyc);
  }
  for (int bpc = 1; bpc <= 2; bpc++) {
    const float p48 = run<4, 48>(cus * bpc, iters), d48 = run<5, 48>(cus * bpc, iters);
    printf("waves/SIMD %d: MFMA + VALU48 reading the OTHER accumulator %.2f ms | reading the accumulator just produced %.2f ms\n", bpc, p48, d48);
  }
  for (int bpc = 1; bpc <= 2; bpc++) {
    const float l48 = run<6, 48>(cus * bpc, iters), lm = run<7, 48>(cus * bpc, iters);
    printf("waves/SIMD %d: LDS-fed MFMA(4, restarting accumulators) + VALU48 on the fresh accumulators %.2f ms | LDS-fed MFMA only %.2f ms | software-pipelined %.2f ms\n", bpc,
           l48, lm, run_pipe(cus * bpc, iters));
  }
  {
    const float x48 = run<3, 48>(cus * 2, iters), x96 = run<3, 96>(cus * 2, iters);
    printf("2 waves/SIMD, one MFMA-only + one VALU-only: VALU48 %.2f ms, VALU96 %.2f ms\n", x48, x96);
  }
  return 0;
}
