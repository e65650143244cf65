// Microbenchmark of the cyclic-reduction level kernel: duration against the number of workgroups, with parts of the kernel switched off
// (bit 1: no Gauss-Jordan, 2: no output stores, 4: no neighbour products, 8: no global loads), and of one solve walk.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOSFM_BCR_UBENCH -I opensfm_amd/csrc -I include tools/ubench_bcr.hip \
//         -o tools/ubench_bcr -L opensfm_amd/csrc -losfm_mi355 -Wl,-rpath,'$ORIGIN/../opensfm_amd/csrc'
#include "ba.hip"

int main() {
  constexpr int CS = 9, n = 54, n2 = n * n, N = 556;
  Dev d;
  memset(&d, 0, sizeof(d));
  d.S = N * CS; d.cs = CS; d.ncd = n; d.ncl = N; d.NC = 1; d.cam0 = 6 * d.S;
  const size_t nb = (size_t)N * n2;
  std::vector<double> hD(nb), hE(nb);
  for (int c = 0; c < N; c++)
    for (int r = 0; r < n; r++)
      for (int q = 0; q < n; q++) {
        hD[(size_t)c * n2 + r * n + q] = (r == q) ? 8.0 + 0.01 * (c % 7) : 0.02 * std::cos(0.1 * (r + q) + c);
        hE[(size_t)c * n2 + r * n + q] = 0.015 * std::sin(0.07 * r - 0.05 * q + 0.3 * c);
      }
  for (int c = 0; c < N; c++)  // symmetric D
    for (int r = 0; r < n; r++)
      for (int q = 0; q < r; q++) hD[(size_t)c * n2 + q * n + r] = hD[(size_t)c * n2 + r * n + q];
  double **arr[] = {&d.bD, &d.bD2, &d.bE, &d.bG, &d.bH, &d.bGt, &d.bHt};
  for (auto a : arr) hipMalloc((void **)a, nb * sizeof(double));
  hipMalloc((void **)&d.bx, (size_t)6 * N * n * sizeof(double));
  double *rin, *z;
  hipMalloc((void **)&rin, (size_t)N * n * sizeof(double));
  hipMalloc((void **)&z, (size_t)(N * n + 16) * sizeof(double));
  hipMalloc((void **)&d.Binv, (size_t)(36 * d.S + 16) * sizeof(double));
  hipMemset(rin, 0, (size_t)N * n * sizeof(double));
  int *status;
  hipMalloc((void **)&status, 16);
  const BcrLaunch lv = bcr_level_for(CS);
  hipFuncSetAttribute((const void *)lv.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto reset = [&]() {
    hipMemcpy(d.bD, hD.data(), nb * sizeof(double), hipMemcpyHostToDevice);
    hipMemcpy(d.bE, hE.data(), nb * sizeof(double), hipMemcpyHostToDevice);
    hipMemset(d.bD2, 0, nb * sizeof(double));
  };
  const int variants[] = {0, 8, 4, 2, 6, 14, 1, 15};
  for (int v : variants) {
    hipMemcpyToSymbol(HIP_SYMBOL(osfm_bcr_variant), &v, sizeof(int));
    printf("variant %2d:", v);
    for (int wgs : {1, 8, 18, 35, 70, 139, 278}) {
      reset();
      hipDeviceSynchronize();
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(lv.fn, dim3(wgs), dim3(lv.threads), lv.lds_bytes, 0, d, 1, 0, status);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
      }
      printf("  %3d wg %7.1f us", wgs, best * 1e3f);
    }
    printf("\n");
  }
  // the whole factorisation and one solve walk, as the solver issues them
  int v0 = 0;
  hipMemcpyToSymbol(HIP_SYMBOL(osfm_bcr_variant), &v0, sizeof(int));
  Solver sv;
  sv.d = d;
  sv.st = 0;
  for (int rep = 0; rep < 3; rep++) {
    reset();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int stq = 1; stq < N; stq *= 2) hipLaunchKernelGGL(lv.fn, dim3((N + 2 * stq - 1) / (2 * stq)), dim3(lv.threads), lv.lds_bytes, 0, d, stq, 0, status);
    hipLaunchKernelGGL(lv.fn, dim3(1), dim3(lv.threads), lv.lds_bytes, 0, d, 1, 1, status);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int hst = -1;
    hipMemcpy(&hst, status, sizeof(int), hipMemcpyDeviceToHost);
    hipEventRecord(e0, 0);
    for (int q = 0; q < 10; q++) sv.bcr_solve_set(RhsSet{rin, 0, z, 0, 1, nullptr, nullptr, -1, -1});
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms2;
    hipEventElapsedTime(&ms2, e0, e1);
    printf("factor (11 levels + root) %.1f us, status %d; solve walk %.1f us\n", ms * 1e3f, hst, ms2 * 1e2f);
  }
  return 0;
}
