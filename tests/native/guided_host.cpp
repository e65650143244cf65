// guided_host.cpp -- TEST INFRASTRUCTURE: the product's guided-matching numerics and wavefront orchestration
// (opensfm_amd/csrc/guided_wave.h) compiled for the HOST with a loop-based wave policy, for bit-for-bit comparison with the
// CPU oracle (tests/test_guided_host.py).  Nothing in the product links or loads this file.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../opensfm_amd/csrc/guided_wave.h"

using namespace osfm_gm;

struct LoopWave {
  template <class F> void single(F f) { f(); }
  template <class F> void parallel_for(int n, F f) { for (int i = 0; i < n; i++) f(i); }
};

extern "C" int host_match_guided(const float* f1, int n1, const float* f2, int n2, const uint8_t* mask, const float* b1, const float* b2,
                                 const double* R, const double* t, double threshold, double ratio, int symmetric, int* out_pairs, int cap) {
  std::vector<uint8_t> u1((size_t)n1 * kDim), u2((size_t)n2 * kDim);
  std::vector<int> nr1(n1 + 1), nr2(n2 + 1);
  auto conv = [](const float* f, int n, std::vector<uint8_t>& u, std::vector<int>& nr) {
    for (int i = 0; i < n; i++) {
      int s = 0;
      for (int k = 0; k < kDim; k++) {
        const int v = (int)f[(size_t)i * kDim + k];
        u[(size_t)i * kDim + k] = (uint8_t)v;
        s += v * v;
      }
      nr[i] = s;
    }
  };
  conv(f1, n1, u1, nr1);
  conv(f2, n2, u2, nr2);
  std::vector<double> first6((size_t)n1 * 6 + 6), second6((size_t)n2 * 6 + 6);
  Allowed allowed{mask, nullptr, nullptr, threshold, n2};
  if (!mask) {
    double tn[3] = {t[0], t[1], t[2]};
    normalized3(tn);
    for (int i = 0; i < n1; i++) epipolar_precompute(0, b1 + 3 * (size_t)i, R, tn, first6.data() + 6 * (size_t)i);
    for (int j = 0; j < n2; j++) epipolar_precompute(1, b2 + 3 * (size_t)j, R, tn, second6.data() + 6 * (size_t)j);
    allowed.first6 = first6.data();
    allowed.second6 = second6.data();
  }
  LoopWave w;
  GuidedShared s;
  std::vector<int> g12(n1 + 1), g21(n2 + 1);
  for (int i = 0; i < n1; i++)
    g12[i] = match_query_wave(w, s, u1.data() + (size_t)i * kDim, nr1[i], i, u2.data(), nr2.data(), n2, allowed, 1, ratio);
  for (int j = 0; j < n2; j++)
    g21[j] = match_query_wave(w, s, u2.data() + (size_t)j * kDim, nr2[j], j, u1.data(), nr1.data(), n1, allowed, 0, ratio);
  int c = 0;
  for (int i = 0; i < n1; i++) {
    int j = g12[i];
    if (j >= 0 && symmetric && g21[j] != i) j = -1;
    if (j >= 0) {
      if (c < cap) { out_pairs[2 * c] = i; out_pairs[2 * c + 1] = j; }
      c++;
    }
  }
  return c;
}
