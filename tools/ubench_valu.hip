// VALU issue-rate microbenchmark for gfx950: wave64 instructions per cycle per SIMD for the integer
// ops of the matcher epilogue.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
template <int OP>
__global__ void __launch_bounds__(256) k(int *out, int seed) {
  int a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x;
  int b = seed * 3 + 1, c = seed + 7;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (OP == 0) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 1) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 3) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 5) { float f = __int_as_float(a[i]); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(__int_as_float(b)), "v"(__int_as_float(c))); a[i] = __float_as_int(f); }
      if (OP == 6) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, int *d, int waves_per_simd) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_simd = (double)ITER * 16 * waves_per_simd;  // wave-instructions per SIMD
  printf("%-16s waves/SIMD %d : %8.3f ms -> %6.2f ns per wave-instr per SIMD  (%.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}
int main() {
  int *d; hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
  for (int w = 1; w <= 2; w++) {
    run<0>("v_max3_i32", d, w); run<1>("v_lshl_add_u32", d, w); run<2>("v_med3_i32", d, w); run<3>("v_max_i32", d, w);
    run<4>("v_add_u32", d, w); run<5>("v_fma_f32", d, w); run<6>("v_pk_max_i16", d, w); run<7>("v_mov_b32", d, w);
  }
  return 0;
}
