// relpose_core_host.cpp -- TEST INFRASTRUCTURE: compiles the product's per-lane numerics (opensfm_amd/csrc/relpose_core.h)
// and its wavefront orchestration (relpose_wave.h) for the HOST with a loop-based wave policy, so that
// tests/test_relpose_core_host.py can compare them bit for bit with the CPU oracle without a GPU.
// Nothing in the product links or loads this file.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../opensfm_amd/csrc/relpose_coop.h"

using namespace osfm_rp;

static int g_batch0 = 64;  // speculative batch schedule (RansacParams::batch0); results must not depend on it
extern "C" void host_set_batch0(int b) { g_batch0 = b; }

template <int WIDTH>
struct LoopWave {  // "lanes" are loop iterations; single() runs once
  static constexpr int width = WIDTH;
  template <class F> void single(F f) { f(); }
  template <class F> void parallel_for(int n, F f) { for (int i = 0; i < n; i++) f(i); }
  template <class P> int count_if(int n, P p) { int c = 0; for (int i = 0; i < n; i++) c += p(i) ? 1 : 0; return c; }
  template <class P> int compact(int n, P p, int* out) { int c = 0; for (int i = 0; i < n; i++) if (p(i)) out[c++] = i; return c; }
};

template <int WIDTH>
static int ransac_impl(const double* b1, const double* b2, int n, double thr, int iterations, double probability, int use_lo, int lo_it,
                       double* model, double* lo_model, int* inliers, int* iters_run) {
  LoopWave<WIDTH> w;
  std::vector<double> models((size_t)kWave * kMaxModels * 12);
  std::vector<int> inl(n > 0 ? n : 1), sub(n > 0 ? n : 1);
  WaveShared* s = new WaveShared;
  std::vector<double> stop(n + 1);
  for (int c = 0; c <= n; c++) stop[c] = max_iterations_for(c, n > 0 ? n : 1, probability);
  PairWork P{b1, b2, n, models.data(), inl.data(), sub.data(), stop.data()};
  RansacParams prm{thr, 1.0 - cos(thr), iterations, probability, use_lo, lo_it, g_batch0};
  RansacResult r;
  ransac_relative_pose_wave(w, *s, P, prm, r);
  delete s;
  memcpy(model, r.model, sizeof(r.model));
  memcpy(lo_model, r.lo_model, sizeof(r.lo_model));
  for (int i = 0; i < r.best_score; i++) inliers[i] = inl[i];
  *iters_run = r.iterations_run;
  return r.best_score;
}
// ---- the cooperative organisation (relpose_coop.h): items of a parallel step in a chosen order, to expose any dependence ----
static int g_order = 0;  // 0 forwards, 1 backwards, 2 shuffled
static unsigned g_lcg = 12345u;
struct OrderWave {
  static constexpr int width = 64;
  template <class F> void single(F f) { f(); }
  template <class F> void parallel_for(int n, F f) {
    if (g_order == 0) {
      for (int i = 0; i < n; i++) f(i);
    } else if (g_order == 1) {
      for (int i = n - 1; i >= 0; i--) f(i);
    } else {
      std::vector<int> idx(n);
      for (int i = 0; i < n; i++) idx[i] = i;
      for (int i = n - 1; i > 0; i--) {
        g_lcg = g_lcg * 1664525u + 1013904223u;
        const int j = (int)((g_lcg >> 8) % (unsigned)(i + 1));
        const int t = idx[i]; idx[i] = idx[j]; idx[j] = t;
      }
      for (int i = 0; i < n; i++) f(idx[i]);
    }
  }
  template <class P> int count_if(int n, P p) { int c = 0; for (int i = 0; i < n; i++) c += p(i) ? 1 : 0; return c; }
  template <class P> int compact(int n, P p, int* out) { int c = 0; for (int i = 0; i < n; i++) if (p(i)) out[c++] = i; return c; }
};

extern "C" {

void host_set_item_order(int order) { g_order = order; }

// real eigenvalues of a general 10 x 10: per-lane routine (which = 0) or the cooperative one (which = 1)
int host_real_eigenvalues10(const double* a_in, int which, double* wr) {
  if (which == 0) {
    double a[100];
    memcpy(a, a_in, sizeof(a));
    return real_eigenvalues10(a, wr);
  }
  OrderWave w;
  CoopShared* c = new CoopShared;
  memset(c, 0xff, sizeof(CoopShared));
  memcpy(c->Aq, a_in, sizeof(double) * 100);
  const int n = real_eigenvalues10_coop(w, *c, c->Aq);
  memcpy(wr, c->wr, sizeof(double) * (size_t)n);
  delete c;
  return n;
}

int host_essential_five_points_v2(const double* b1, const double* b2, double* Es) {
  OrderWave w;
  CoopShared* c = new CoopShared;
  memset(c, 0xff, sizeof(CoopShared));  // poison: nothing may be read before it is written
  for (int i = 0; i < 15; i++) { c->s1[i] = b1[i]; c->s2[i] = b2[i]; }
  const int n = essential_five_points_coop(w, *c);
  memcpy(Es, c->Es, sizeof(double) * 9 * (size_t)n);
  delete c;
  return n;
}

int host_ransac_relative_pose_v2(const double* b1, const double* b2, int n, double thr, int iterations, double probability, int use_lo, int lo_it,
                                 double* model, double* lo_model, int* inliers, int* iters_run) {
  OrderWave w;
  std::vector<int> inl(n > 0 ? n : 1), sub(n > 0 ? n : 1);
  std::vector<double> stop(n + 1);
  for (int c = 0; c <= n; c++) stop[c] = max_iterations_for(c, n > 0 ? n : 1, probability);
  WaveShared* s = new WaveShared;
  CoopShared* c = new CoopShared;
  PairWork P{b1, b2, n, nullptr, inl.data(), sub.data(), stop.data()};
  RansacParams prm{thr, 1.0 - cos(thr), iterations, probability, use_lo, lo_it, 64};
  RansacResult r;
  ransac_relative_pose_seq(w, *s, *c, P, prm, r);
  delete s;
  delete c;
  memcpy(model, r.model, sizeof(r.model));
  memcpy(lo_model, r.lo_model, sizeof(r.lo_model));
  for (int i = 0; i < r.best_score; i++) inliers[i] = inl[i];
  *iters_run = r.iterations_run;
  return r.best_score;
}

int host_robust_match_calibrated_v2(const double* b1, const double* b2, int n, double thr, int iterations, double probability, int use_lo,
                                    int lo_it, int refine_iterations, double* R, double* t, uint8_t* mask, double* ransac_models,
                                    int* ransac_info) {
  OrderWave w;
  std::vector<int> inl(n > 0 ? n : 1), sub(n > 0 ? n : 1);
  std::vector<double> stop(n + 1);
  for (int c = 0; c <= n; c++) stop[c] = max_iterations_for(c, n > 0 ? n : 1, probability);
  WaveShared* s = new WaveShared;
  CoopShared* c = new CoopShared;
  PairWork P{b1, b2, n, nullptr, inl.data(), sub.data(), stop.data()};
  RansacParams prm{thr, 1.0 - cos(thr), iterations, probability, use_lo, lo_it, 64};
  MatchResult r;
  robust_match_calibrated_seq(w, *s, *c, P, prm, refine_iterations, r);
  delete s;
  delete c;
  memset(mask, 0, (size_t)(n > 0 ? n : 0));
  for (int i = 0; i < r.n_inliers; i++) mask[sub[i]] = 1;
  memcpy(R, r.R, sizeof(r.R));
  memcpy(t, r.t, sizeof(r.t));
  memcpy(ransac_models, r.ransac.model, sizeof(r.ransac.model));
  memcpy(ransac_models + 12, r.ransac.lo_model, sizeof(r.ransac.lo_model));
  ransac_info[0] = r.ransac.best_score;
  ransac_info[1] = r.ransac.iterations_run;
  return r.n_inliers;
}

int host_essential_five_points(const double* b1, const double* b2, double* Es) { return essential_five_points(b1, b2, Es); }
int host_relative_pose_from_essential(const double* E, const double* b1, const double* b2, int n, double* RT) {
  return relative_pose_from_essential(E, b1, b2, nullptr, n, RT);
}
void host_pixel_bearings(int model, const double* cam, const double* px, int n, double* out) {
  for (int i = 0; i < n; i++) pixel_bearing(model, cam[0], cam[1], cam[2], px[2 * i], px[2 * i + 1], out + 3 * i);
}
void host_pixel_bearings_generic(int model, const double* par, const double* px, int n, double* out) {
  for (int i = 0; i < n; i++) pixel_bearing_generic(model, par, px[2 * i], px[2 * i + 1], out + 3 * i);
}
void host_inliers_bearings(const double* b1, const double* b2, int n, const double* R, const double* t, double thr, uint8_t* mask) {
  for (int i = 0; i < n; i++) mask[i] = (uint8_t)inlier_bearing(b1 + 3 * i, b2 + 3 * i, R, t, thr);
}
void host_refinement_picks(int n, int* picked) { refinement_picks(n, picked); }

int host_ransac_relative_pose(int width, const double* b1, const double* b2, int n, double thr, int iterations, double probability,
                              int use_lo, int lo_it, double* model, double* lo_model, int* inliers, int* iters_run) {
  switch (width) {
    case 1: return ransac_impl<1>(b1, b2, n, thr, iterations, probability, use_lo, lo_it, model, lo_model, inliers, iters_run);
    case 7: return ransac_impl<7>(b1, b2, n, thr, iterations, probability, use_lo, lo_it, model, lo_model, inliers, iters_run);
    default: return ransac_impl<64>(b1, b2, n, thr, iterations, probability, use_lo, lo_it, model, lo_model, inliers, iters_run);
  }
}

int host_relative_pose_refinement(double* RT, const double* b1, const double* b2, int n, int iterations, double* costs) {
  LoopWave<64> w;
  WaveShared* s = new WaveShared;
  std::vector<int> subset(n);
  for (int i = 0; i < n; i++) subset[i] = i;
  refinement_picks(n, s->picked);
  WaveRefineEval<LoopWave<64>> ev{w, *s, b1, b2, subset.data()};
  const int it = refine_relative_pose(RT, iterations, ev, costs);
  delete s;
  return it;
}

int host_robust_match_calibrated(const double* b1, const double* b2, int n, double thr, int iterations, double probability, int use_lo,
                                 int lo_it, int refine_iterations, double* R, double* t, uint8_t* mask, double* ransac_models,
                                 int* ransac_info) {
  LoopWave<64> w;
  std::vector<double> models((size_t)kWave * kMaxModels * 12);
  std::vector<int> inl(n > 0 ? n : 1), sub(n > 0 ? n : 1);
  WaveShared* s = new WaveShared;
  std::vector<double> stop(n + 1);
  for (int c = 0; c <= n; c++) stop[c] = max_iterations_for(c, n > 0 ? n : 1, probability);
  PairWork P{b1, b2, n, models.data(), inl.data(), sub.data(), stop.data()};
  RansacParams prm{thr, 1.0 - cos(thr), iterations, probability, use_lo, lo_it, g_batch0};
  MatchResult r;
  robust_match_calibrated_wave(w, *s, P, prm, refine_iterations, r);
  delete s;
  memset(mask, 0, (size_t)(n > 0 ? n : 0));
  for (int i = 0; i < r.n_inliers; i++) mask[sub[i]] = 1;
  memcpy(R, r.R, sizeof(r.R));
  memcpy(t, r.t, sizeof(r.t));
  memcpy(ransac_models, r.ransac.model, sizeof(r.ransac.model));
  memcpy(ransac_models + 12, r.ransac.lo_model, sizeof(r.ransac.lo_model));
  ransac_info[0] = r.ransac.best_score;
  ransac_info[1] = r.ransac.iterations_run;
  return r.n_inliers;
}
}
