// which lane holds which element of v_mfma_f64_16x16x4_f64's operands and result (run on gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D) {  // A 16x4 row-major, B 4x16 row-major; D[lane*4+r]
  const int lane = threadIdx.x;
  const double a = A[(lane & 15) * 4 + (lane >> 4)], b = B[(lane >> 4) * 16 + (lane & 15)];
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[lane * 4 + r] = c[r];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) hA[i * 4 + k] = 1 + i + 100 * k;
  for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) hB[k * 16 + j] = (k == 0 ? 1 : 0) * (1 + j) + (k == 1 ? 0.001 * (j * j + 1) : 0) + (k == 2 ? 7 + j * 3 : 0) + (k == 3 ? -2.0 * j : 0);
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int lane = 0; lane < 64; lane++) for (int r = 0; r < 4; r++) {
    const double v = hD[lane * 4 + r];
    if (v != ref[((lane >> 4) + 4 * r) * 16 + (lane & 15)]) bad1++;
    if (v != ref[((lane >> 4) * 4 + r) * 16 + (lane & 15)]) bad2++;
  }
  printf("row=(lane>>4)+4r: %d mismatches; row=4(lane>>4)+r: %d mismatches\n", bad1, bad2);
  if (bad1 && bad2) for (int lane = 0; lane < 64; lane += 7) for (int r = 0; r < 4; r++) {
    const double v = hD[lane * 4 + r];
    int fi = -1, fj = -1;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) if (ref[i * 16 + j] == v) { fi = i; fj = j; }
    printf("lane %d reg %d -> (%d, %d)\n", lane, r, fi, fj);
  }
  return 0;
}
