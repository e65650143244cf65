// relpose_core_host.cpp -- TEST INFRASTRUCTURE: compiles the product's per-lane numerics (opensfm_amd/csrc/relpose_core.h)
// and the round-based orchestration of the LO-RANSAC (relpose_rounds.h) for the HOST with loops in place of kernels, so that
// tests/test_relpose_core_host.py can compare them bit for bit with the CPU oracle without a GPU.
// Nothing in the product links or loads this file.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <random>

#include "../../opensfm_amd/csrc/relpose_rounds.h"

using namespace osfm_rp;

static int g_max_width = kMaxSlots;  // cap of the speculative width (Rounds::max_width); results must not depend on it
extern "C" void host_set_max_width(int b) { g_max_width = b; }
// stage B1 by the group of lanes against stage B1 by one lane, over every five-point problem the rounds of this process solved
static long g_eig_problems = 0, g_eig_mismatches = 0;
extern "C" long host_eig_problems() { return g_eig_problems; }
extern "C" long host_eig_mismatches() { return g_eig_mismatches; }

struct LoopWave {  // "lanes" are loop iterations; single() runs once
  template <class F> void single(F f) { f(); }
  template <class F> void parallel_for(int n, F f) { for (int i = 0; i < n; i++) f(i); }
  template <class P> int count_if(int n, P p) { int c = 0; for (int i = 0; i < n; i++) c += p(i) ? 1 : 0; return c; }
  template <class P> int compact(int n, P p, int* out) { int c = 0; for (int i = 0; i < n; i++) if (p(i)) out[c++] = i; return c; }
  template <class P> int compact_changed(int n, P p, int* out, int* changed) {
    int c = 0;
    for (int i = 0; i < n; i++)
      if (p(i)) {
        if (out[c] != i) *changed = 1;
        out[c++] = i;
      }
    return c;
  }
  int atomic_add(int* p, int v) { const int o = *p; *p += v; return o; }
  // the GPU stages a window of the stream in LDS; here half a window, so that both paths of RngView::get are exercised
  RngView stage_rng(const RngTable& T, uint32_t* buf, int pos, bool want) {
    int n = 0;
    if (want)
      for (; n < kRngCache / 16 && pos + n < T.size; n++) buf[n] = T.tab[pos + n];
    return RngView{T, buf, pos, n};
  }
};

static const std::vector<uint32_t>& rng_table() {
  static std::vector<uint32_t> t;
  if (t.empty()) {
    std::mt19937 g(42);
    t.resize(1 << 20);
    for (auto& v : t) v = (uint32_t)g();
  }
  return t;
}

// The rounds of relpose_rounds.h over a batch of pairs, with loops in place of kernels.  The solver work spaces use the LDS layout
// of the GPU kernels (LaneArr with stride 64; the "lane" of a problem is its position in the work list modulo 64).
struct HostRounds {
  Rounds R;
  std::vector<PairState> st;
  std::vector<int> sidx, pos_before, pos_after, nmodels, lidx, lo_pos_after, lo_ok, inliers, list5, listN, counters;
  std::vector<double> models, lo_rt, stop, u1, u2, s5_at6, s5_basis, s5_E, lo_E;
  std::vector<double> s5_wr, s5_wr_check;
  std::vector<int> s5_nreal, s5_nreal_check, s5_valid;
  long eig_mismatches = 0;
  std::vector<int> s5_ok;
  std::vector<int64_t> stop_off;
  int rounds_run = 0;
  long solved5 = 0, solvedN = 0;
  HostRounds(const double* b1, const double* b2, const int64_t* off, int n_pairs, double thr, int iterations, double probability, int use_lo,
             int lo_it, int min_n) {
    const int64_t total = off[n_pairs];
    st.resize(n_pairs);
    sidx.resize((size_t)n_pairs * kMaxSlots * 5);
    pos_before.resize((size_t)n_pairs * kMaxSlots);
    pos_after.resize((size_t)n_pairs * kMaxSlots);
    nmodels.resize((size_t)n_pairs * kMaxSlots);
    models.resize((size_t)n_pairs * kMaxSlots * kMaxModels * 12);
    const int lo = lo_it > 0 ? lo_it : 1;
    lidx.resize((size_t)n_pairs * lo * kLoSampleMax);
    lo_pos_after.resize((size_t)n_pairs * lo);
    lo_ok.resize((size_t)n_pairs * lo);
    lo_rt.resize((size_t)n_pairs * lo * 12);
    inliers.resize((size_t)(total > 0 ? total : 1));
    list5.resize((size_t)n_pairs * kMaxSlots);
    const size_t cap5 = ((size_t)n_pairs * kMaxSlots + 63) / 64 * 64;
    s5_at6.resize(cap5 * 60);
    s5_basis.resize(cap5 * 36);
    s5_ok.resize(cap5);
    s5_E.resize(cap5 * kMaxModels * 9);
    lo_E.resize((size_t)n_pairs * lo * 9);
    s5_wr.assign(cap5 * kMaxModels, 0.0);
    s5_wr_check.assign(cap5 * kMaxModels, 0.0);
    s5_nreal.assign(cap5, 0);
    s5_nreal_check.assign(cap5, 0);
    s5_valid.assign(cap5 * kMaxModels, 0);
    listN.resize((size_t)n_pairs * lo);
    counters.assign(4, 0);
    stop.resize((size_t)total + n_pairs);
    stop_off.resize(n_pairs);
    for (int p = 0; p < n_pairs; p++) {
      const int n = (int)(off[p + 1] - off[p]);
      stop_off[p] = off[p] + p;
      for (int c = 0; c <= n; c++) stop[off[p] + p + c] = max_iterations_for(c, n > 0 ? n : 1, probability);
    }
    u1.resize((size_t)(total > 0 ? total : 1) * 3);
    u2.resize((size_t)(total > 0 ? total : 1) * 3);
    R = Rounds{b1, b2, u1.data(), u2.data(), off, n_pairs, stop.data(), stop_off.data(), RngTable{rng_table().data(), (int)rng_table().size()}, 1.0 - cos(thr), iterations, use_lo,
               lo_it, min_n, g_max_width, st.data(), sidx.data(), pos_before.data(), pos_after.data(), nmodels.data(), models.data(), lidx.data(),
               lo_pos_after.data(), lo_ok.data(), lo_rt.data(), inliers.data(), s5_at6.data(), s5_basis.data(), s5_ok.data(), s5_E.data(), lo_E.data(), list5.data(), listN.data(), counters.data(),
               s5_wr.data(), s5_nreal.data(), s5_valid.data()};
  }
  void run() {
    LoopWave w;
    WalkShared* sh = new WalkShared;
    std::vector<double> lds5((size_t)64 * 256);
    std::vector<int> ldsI((size_t)64 * 10);
    for (int p = 0; p < R.n_pairs; p++) pair_init(R, p);
    for (int64_t k = 0; k < R.offsets[R.n_pairs]; k++) pair_normalise(R, k);
    for (;;) {
      counters[0] = counters[1] = counters[2] = 0;
      for (int p = 0; p < R.n_pairs; p++) pair_walk(w, *sh, R, p);
      rounds_run++;
      if (getenv("OSFM_HOST_TRACE") && R.n_pairs == 1)
        fprintf(stderr, "round %d: phase %d it %d best %d pos %d n5 %d nN %d lo_l %d slot %d/%d model %d\n", rounds_run, st[0].phase, st[0].it, st[0].best_score,
                st[0].pos, counters[0], counters[1], st[0].lo_l, st[0].cur_slot, st[0].nslots, st[0].cur_model);
      if (counters[2] == 0) {
        delete sh;
        g_eig_mismatches += eig_mismatches;
        g_eig_problems += solved5;
        break;
      }
      solved5 += counters[0];
      solvedN += counters[1];
      typedef LaneArr<double, 64> D;
      typedef LaneArr<int, 64> I;
      for (int k = 0; k < counters[0]; k++) {  // stage A: LDS layout basis | M, colperm
        const int lane = k % 64;
        D base{lds5.data() + lane};
        solve5_stage_a(R, k, base, base + 36, I{ldsI.data() + lane});
      }
      // stage B1 twice: by the group of lanes the GPU uses (here its lane loops run in turn) and by one lane; the eigenvalues must
      // agree bit for bit (eig_mismatches is checked by the test)
      for (int k = 0; k < counters[0]; k++) {
        double shared_a[104];
        solve5_stage_b1_group(R, k, shared_a, 0);
      }
      for (int k = 0; k < counters[0]; k++) {
        s5_nreal_check[k] = s5_nreal[k];
        for (int q = 0; q < kMaxModels; q++) s5_wr_check[(size_t)k * kMaxModels + q] = s5_wr[(size_t)k * kMaxModels + q];
      }
      for (int k = 0; k < counters[0]; k++) {
        const int lane = k % 64;
        solve5_stage_b1(R, k, D{lds5.data() + lane});
        if (s5_nreal[k] != s5_nreal_check[k]) eig_mismatches++;
        for (int q = 0; q < s5_nreal[k] && q < s5_nreal_check[k]; q++)
          if (memcmp(&s5_wr[(size_t)k * kMaxModels + q], &s5_wr_check[(size_t)k * kMaxModels + q], 8) != 0) eig_mismatches++;
      }
      for (int q = 0; q < counters[0] * kMaxModels; q++) {  // stage B2: LDS layout S
        const int lane = q % 64;
        solve5_stage_b2(R, q, D{lds5.data() + lane});
      }
      for (int q = 0; q < counters[0] * kMaxModels; q++) pose5_item(R, q);
      for (int k = 0; k < counters[1]; k++) {  // N-point: registers on the GPU, plain arrays here
        double AtA[81], V[81], wv[9];
        solveN_problem<double*>(R, k, AtA, V, wv);
      }
      for (int k = 0; k < counters[1]; k++) poseN_item(R, k);
    }
  }
};

static int ransac_one(const double* b1, const double* b2, int n, double thr, int iterations, double probability, int use_lo, int lo_it,
                      double* model, double* lo_model, int* inliers, int* iters_run) {
  const int64_t off[2] = {0, n};
  HostRounds H(b1, b2, off, 1, thr, iterations, probability, use_lo, lo_it, 5);
  H.run();
  const PairState& S = H.st[0];
  memcpy(model, S.model, sizeof(S.model));
  memcpy(lo_model, S.lo_model, sizeof(S.lo_model));
  for (int i = 0; i < S.best_score; i++) inliers[i] = H.inliers[i];
  *iters_run = S.it;
  return S.best_score;
}

extern "C" {

// a batch of pairs through the rounds (RANSAC mode): scores[p], iterations[p], models[p][24] = model | lo_model, mask over all
// correspondences = inliers of the best score; returns the number of rounds
int host_rounds_ransac_batch(const double* b1, const double* b2, const int64_t* off, int n_pairs, double thr, int iterations, double probability,
                             int use_lo, int lo_it, int* scores, int* iters, double* models, uint8_t* mask) {
  HostRounds H(b1, b2, off, n_pairs, thr, iterations, probability, use_lo, lo_it, 5);
  H.run();
  memset(mask, 0, (size_t)off[n_pairs]);
  for (int p = 0; p < n_pairs; p++) {
    const PairState& S = H.st[p];
    scores[p] = S.best_score;
    iters[p] = S.it;
    memcpy(models + 24 * p, S.model, sizeof(S.model));
    memcpy(models + 24 * p + 12, S.lo_model, sizeof(S.lo_model));
    for (int i = 0; i < S.best_score; i++) mask[off[p] + H.inliers[off[p] + i]] = 1;
  }
  if (getenv("OSFM_HOST_STATS")) fprintf(stderr, "stats rounds %d solved5 %ld solvedN %ld\n", H.rounds_run, H.solved5, H.solvedN);
  return H.rounds_run;
}

// samples of the tabulated stream: `count` samples of `size` distinct indices below n, drawn one after the other
void host_draw_samples(int n, int size, int count, int* out) {
  const RngTable T{rng_table().data(), (int)rng_table().size()};
  int pos = 0, overflow = 0;
  for (int c = 0; c < count; c++) pos = draw_sample_tab(T, pos, size, n, out + (size_t)c * size, &overflow);
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
  const int saved = g_max_width;
  g_max_width = width < 1 ? 1 : (width > kMaxSlots ? kMaxSlots : width);
  const int r = ransac_one(b1, b2, n, thr, iterations, probability, use_lo, lo_it, model, lo_model, inliers, iters_run);
  g_max_width = saved;
  return r;
}

int host_relative_pose_refinement(double* RT, const double* b1, const double* b2, int n, int iterations, double* costs) {
  LoopWave w;
  RefineShared* s = new RefineShared;
  std::vector<int> subset(n);
  for (int i = 0; i < n; i++) subset[i] = i;
  refinement_picks(n, s->picked);
  WaveRefineEval<LoopWave> ev{w, *s, b1, b2, subset.data()};
  const int it = refine_relative_pose(RT, iterations, ev, costs);
  delete s;
  return it;
}

int host_robust_match_calibrated(const double* b1, const double* b2, int n, double thr, int iterations, double probability, int use_lo,
                                 int lo_it, int refine_iterations, double* R, double* t, uint8_t* mask, double* ransac_models,
                                 int* ransac_info) {
  const int64_t off[2] = {0, n};
  HostRounds H(b1, b2, off, 1, thr, iterations, probability, use_lo, lo_it, 8);
  H.run();
  const PairState& S = H.st[0];
  memset(mask, 0, (size_t)(n > 0 ? n : 0));
  memcpy(ransac_models, S.model, sizeof(S.model));
  memcpy(ransac_models + 12, S.lo_model, sizeof(S.lo_model));
  ransac_info[0] = S.best_score;
  ransac_info[1] = S.it;
  for (int i = 0; i < 9; i++) R[i] = 0.0;
  for (int i = 0; i < 3; i++) t[i] = 0.0;
  if (S.rejected) return 0;
  LoopWave w;
  RefineShared* rs = new RefineShared;
  std::vector<int> sub(n > 0 ? n : 1);
  const int cnt = robust_match_finish_wave(w, *rs, b1, b2, n, S.lo_model, thr, refine_iterations, sub.data(), R, t);
  delete rs;
  for (int i = 0; i < cnt; i++) mask[sub[i]] = 1;
  return cnt;
}
}
