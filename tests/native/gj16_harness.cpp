// gj16_harness.cpp -- TEST INFRASTRUCTURE (tests/test_emu_ba.py builds and runs it): the product's in-LDS Gauss-Jordan inverse with
// 16 x 16 pivots on the matrix cores (opensfm_amd/csrc/ba.hip: wide_gj_inverse_mfma, inv16_spd_wave -- the pivot step of the
// dense-cluster cyclic reduction) run by the HIP host emulation on random SPD blocks of the orders the solver meets (a full panel, the
// block survey's 90 / 72, a remainder of 36 / 18, a single shot), against the definition: max |A A^-1 - I|.
// The generated source _build/ba_emu.cpp (tests/native/build_emu.py) is included whole: the functions live in its anonymous namespace.
#include "ba_emu.cpp"
#include <random>
namespace {
__global__ void gj_test_kernel(const double *A, double *out, int w, int *status) {
  double *lds = (double *)hipemu::dyn_lds();
  double *X = lds;
  const int tid = threadIdx.x;
  for (int t = tid; t < kWB * kWB; t += 256) {
    const int c = t / kWB, r = t - c * kWB;
    X[r * kWLd + c] = (r < w && c < w) ? A[c * w + r] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  int bad = 0;
  wide_gj_inverse_mfma(X, lds + kWB * kWLd, tid, bad, (w + 15) / 16);
  if (bad) status[0] = 1;
  for (int t = tid; t < w * w; t += 256) {
    const int c = t / w, r = t - c * w;
    out[c * w + r] = X[r * kWLd + c];
  }
}
}
int main() {
  for (int w : {90, 96, 72, 36, 18, 6}) {
    std::mt19937 g(w);
    std::normal_distribution<double> nd;
    std::vector<double> B(w * w), A(w * w, 0.0), Inv(w * w);
    for (auto &x : B) x = nd(g);
    for (int i = 0; i < w; i++) for (int j = 0; j < w; j++) { double s = 0; for (int k = 0; k < w; k++) s += B[i * w + k] * B[j * w + k]; A[j * w + i] = s + (i == j ? w : 0); }
    double *dA, *dO; int *dS;
    hipMalloc(&dA, w * w * 8); hipMalloc(&dO, w * w * 8); hipMalloc(&dS, 16);
    memcpy(dA, A.data(), w * w * 8); memset(dS, 0, 16);
    hipLaunchKernelGGL(gj_test_kernel, dim3(1), dim3(256), (size_t)(kWB * kWLd + kGj16Scratch) * sizeof(double), 0, dA, dO, w, dS);
    // residual |A Inv - I|
    double err = 0;
    for (int i = 0; i < w; i++) for (int j = 0; j < w; j++) { double s = 0; for (int k = 0; k < w; k++) s += A[k * w + i] * dO[j * w + k]; err = std::max(err, std::fabs(s - (i == j))); }
    printf("w %d: max |A inv - I| = %.3e status %d\n", w, err, dS[0]);
  }
}
