// relpose_wave.h -- wavefront-level orchestration of calibrated robust matching for ONE image pair.
//
// reference: robust::Estimate<RansacScoring, RelativePose> (opensfm/src/robust/robust_estimator.h:37-119) and
// matching.robust_match_calibrated (opensfm/matching.py:871-903).
//
// The reference's RANSAC is sequential: iteration k + 1 draws from the generator state iteration k left behind, and
// the local optimisation draws too whenever a model ties or beats the best score.  One wavefront works on one pair:
//   1. lane 0 draws the samples of the next W iterations (W = wavefront width) ASSUMING no local optimisation runs;
//   2. every lane solves the five-point problem of one iteration (the expensive, divergent part) -> up to 10 poses;
//   3. the (iteration, model) list is then walked in the reference's order; each score is a wavefront-wide count
//      over the correspondences; the first time local optimisation fires the generator is rewound to where that
//      iteration left it, the rest of the speculative batch is thrown away and a new batch starts afterwards.
// The result is the reference's decision sequence exactly, with the five-point solves batched 64 wide.
//
// The code is written against a small "wave policy" W (parallel_for / count_if / compact / single) so that the very
// same orchestration runs on the GPU (relpose.hip: one lane per item, ballots) and in a host emulation
// (tests/native/relpose_core_host.cpp: plain loops) where it is compared bit for bit with the CPU oracle.
// Scalars outside the policy calls are wave-uniform: every lane computes them redundantly from uniform data.
#pragma once
#include "relpose_core.h"

namespace osfm_rp {

constexpr int kWave = 64;
constexpr int kMaxModels = 10;

struct RansacParams {
  double threshold_angle;  // radians (robust_matching_calib_threshold)
  double threshold_score;  // 1 - cos(threshold_angle) (RelativePose::ThresholdAdapter), computed by the HOST libm so that the
                           // device math library cannot move the inlier boundary
  int iterations;          // RobustEstimatorParams::iterations
  double probability;      // ::probability
  int use_lo;              // ::use_local_optimization
  int lo_iterations;       // ::local_optimization_iterations
  int batch0;              // speculative batch schedule: width of the first batch and of the batch after every rewind; it doubles
                           // (up to the wavefront width) while no local optimisation fires.  Results do not depend on it.
};
// (probability only enters through PairWork::stop_bound, tabulated on the host)

// per-pair scratch that every lane may read and write (LDS on the GPU)
struct WaveShared {
  uint32_t mt[624], mt_bak[624];
  int mt_idx, mt_idx_bak;
  int calls;                 // raw generator outputs consumed since the snapshot
  int calls_after[kWave];    // ... after the sample of batch slot k was drawn
  int sidx[kWave][5];
  int nmodels[kWave];
  int lidx[12];
  int lo_ok;
  double lo_rt[12];
  int picked[kRefineResiduals];
  double rbuf[kRefineResiduals + 1][7];  // residual + gradient of the refinement
};

struct PairWork {
  const double* b1;  // n x 3 bearings of the pair's matches (global memory)
  const double* b2;
  int n;
  double* models;  // kWave * kMaxModels * 12 doubles (global workspace)
  int* inliers;    // n ints: inlier list of the best score
  int* subset;     // n ints: inlier list of the refinement stages
  const double* stop_bound;  // n + 1 doubles: max_iterations_for(best inlier count) of ShouldStop, tabulated by the host
};

struct RansacResult {
  double model[12], lo_model[12];
  int best_score, iterations_run;
};

OSFM_HD uint32_t mt_next_counted(WaveShared& s) {
  Mt19937 g{s.mt, s.mt_idx};
  const uint32_t r = mt_next(g);
  s.mt_idx = g.idx;
  s.calls++;
  return r;
}
OSFM_HD void draw_sample_shared(WaveShared& s, int size, int n, int* idx) {
  // same as draw_sample(), on the shared generator, counting raw outputs
  for (int i = 0; i < size; i++) {
    int dup;
    do {
      const uint32_t range = (uint32_t)n;  // Lemire's multiply-shift, as mt_uniform()
      uint64_t product = (uint64_t)mt_next_counted(s) * (uint64_t)range;
      uint32_t low = (uint32_t)product;
      if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
          product = (uint64_t)mt_next_counted(s) * (uint64_t)range;
          low = (uint32_t)product;
        }
      }
      idx[i] = (int)(uint32_t)(product >> 32);
      dup = 0;
      for (int j = 0; j < i; j++) dup |= idx[j] == idx[i];
    } while (dup);
  }
}

template <class W>
OSFM_HD void ransac_relative_pose_wave(W& w, WaveShared& s, const PairWork& P, const RansacParams& prm, RansacResult& out) {
  const int n = P.n;
  const double thr = prm.threshold_score;
  for (int i = 0; i < 12; i++) out.model[i] = out.lo_model[i] = 0.0;
  out.best_score = 0;
  out.iterations_run = 0;
  if (n < 5) return;
  w.single([&]() {
    Mt19937 g{s.mt, 0};
    mt_seed(g, 42u);
    s.mt_idx = g.idx;
    s.calls = 0;
  });
  int best_score = 0, best_n = 0, it = 0, stop = 0;
  int width = prm.batch0 < 1 ? 1 : (prm.batch0 > W::width ? W::width : prm.batch0);
  while (it < prm.iterations && !stop) {
    const int B = (prm.iterations - it) < width ? (prm.iterations - it) : width;
    w.single([&]() {  // snapshot + speculative draws
      for (int i = 0; i < 624; i++) s.mt_bak[i] = s.mt[i];
      s.mt_idx_bak = s.mt_idx;
      s.calls = 0;
      for (int k = 0; k < B; k++) {
        draw_sample_shared(s, 5, n, s.sidx[k]);
        s.calls_after[k] = s.calls;
      }
    });
    w.parallel_for(B, [&](int k) {  // minimal solver, one iteration per lane
      double s1[15], s2[15], Es[90];
      for (int q = 0; q < 5; q++)
        for (int a = 0; a < 3; a++) {
          s1[3 * q + a] = P.b1[3 * s.sidx[k][q] + a];
          s2[3 * q + a] = P.b2[3 * s.sidx[k][q] + a];
        }
      const int nm = essential_five_points(s1, s2, Es);
      for (int j = 0; j < nm; j++) {
        double RT[12];
        for (int i = 0; i < 12; i++) RT[i] = 0.0;
        relative_pose_from_essential(Es + 9 * j, s1, s2, nullptr, 5, RT);
        double* dst = P.models + ((size_t)k * kMaxModels + j) * 12;
        for (int i = 0; i < 12; i++) dst[i] = RT[i];
      }
      s.nmodels[k] = nm;
    });
    int rolled = 0, k = 0;
    for (; k < B && !stop; k++) {
      const int nm = s.nmodels[k];
      for (int j = 0; j < nm && !stop; j++) {
        double RT[12];
        {
          const double* src = P.models + ((size_t)k * kMaxModels + j) * 12;
          for (int i = 0; i < 12; i++) RT[i] = src[i];
        }
        const int cnt = w.count_if(n, [&](int i) { return fabs(relpose_error(RT, P.b1 + 3 * i, P.b2 + 3 * i)) < thr; });
        if (cnt >= best_score) {  // std::max(score, best_score) returns its first argument on a tie: the newcomer replaces the best
          best_score = cnt;
          best_n = cnt;
          w.compact(n, [&](int i) { return fabs(relpose_error(RT, P.b1 + 3 * i, P.b2 + 3 * i)) < thr; }, P.inliers);
          for (int i = 0; i < 12; i++) out.model[i] = out.lo_model[i] = RT[i];
        }
        const int best_found = (cnt == best_score) && cnt >= 5;
        if (best_found && prm.use_lo) {
          if (!rolled) {  // rewind the generator to just after this iteration's sample
            rolled = 1;
            w.single([&]() {
              for (int i = 0; i < 624; i++) s.mt[i] = s.mt_bak[i];
              s.mt_idx = s.mt_idx_bak;
              const int replay = s.calls_after[k];
              s.calls = 0;
              for (int c = 0; c < replay; c++) (void)mt_next_counted(s);
            });
          }
          for (int l = 0; l < prm.lo_iterations; l++) {
            const int ninl = best_n;
            int lo_size = (int)(ninl * 0.5);
            if (lo_size > 12) lo_size = 12;
            if (lo_size < 5) lo_size = 5;
            w.single([&]() {
              int pick[12];
              draw_sample_shared(s, lo_size, ninl, pick);
              for (int q = 0; q < lo_size; q++) s.lidx[q] = P.inliers[pick[q]];
              double Elo[9];
              s.lo_ok = essential_n_points(P.b1, P.b2, s.lidx, lo_size, Elo);
              if (s.lo_ok) {
                double RTlo[12];
                for (int i = 0; i < 12; i++) RTlo[i] = 0.0;
                relative_pose_from_essential(Elo, P.b1, P.b2, s.lidx, lo_size, RTlo);
                for (int i = 0; i < 12; i++) s.lo_rt[i] = RTlo[i];
              }
            });
            if (!s.lo_ok) continue;
            double RTlo[12];
            for (int i = 0; i < 12; i++) RTlo[i] = s.lo_rt[i];
            const int c2 = w.count_if(n, [&](int i) { return fabs(relpose_error(RTlo, P.b1 + 3 * i, P.b2 + 3 * i)) < thr; });
            if (c2 >= best_score) {  // ties included, as above
              best_score = c2;
              best_n = c2;
              w.compact(n, [&](int i) { return fabs(relpose_error(RTlo, P.b1 + 3 * i, P.b2 + 3 * i)) < thr; }, P.inliers);
              for (int i = 0; i < 12; i++) out.lo_model[i] = RTlo[i];
            }
          }
        }
        stop = P.stop_bound[best_n] < (double)(it + k);
      }
      if (rolled) {  // the speculative samples after this iteration are void
        k++;
        break;
      }
    }
    it += k;
    if (rolled)
      width = prm.batch0 < 1 ? 1 : (prm.batch0 > W::width ? W::width : prm.batch0);
    else
      width = 2 * width > W::width ? W::width : 2 * width;
  }
  out.best_score = best_score;
  out.iterations_run = it;
}

// Evaluator of the refinement residuals over the wavefront (see refine_relative_pose in relpose_core.h)
template <class W>
struct WaveRefineEval {
  W& w;
  WaveShared& s;
  const double* b1;
  const double* b2;
  const int* subset;  // correspondence indices of the inlier subset
  OSFM_HD void eval(const double* par, int) {
    double p[6];
    for (int k = 0; k < 6; k++) p[k] = par[k];
    w.parallel_for(kRefineResiduals + 1, [&](int i) {
      double o[7];
      const int m = i < kRefineResiduals ? subset[s.picked[i]] : 0;
      refine_residual(i, p, b1 + 3 * m, b2 + 3 * m, o);
      for (int k = 0; k < 7; k++) s.rbuf[i][k] = o[k];
    });
  }
  OSFM_HD double res(int i) const { return s.rbuf[i][0]; }
  OSFM_HD double jac(int i, int k) const { return s.rbuf[i][1 + k]; }
};

struct MatchResult {
  double R[9], t[3];  // pose of the second camera in the first after the last refinement
  int n_inliers;      // written to PairWork::subset (ascending correspondence indices); 0 = rejected
  RansacResult ransac;
};

// robust_match_calibrated on bearings: LO-RANSAC, 3 x (inliers at 4, 2, 1 x threshold -> refinement), inliers.
// Part 1 (before the RANSAC): the `len(matches) < 8` gate; returns 0 when the pair is rejected outright.
OSFM_HD int robust_match_begin(const PairWork& P, MatchResult& out) {
  out.n_inliers = 0;
  for (int i = 0; i < 9; i++) out.R[i] = 0.0;
  for (int i = 0; i < 3; i++) out.t[i] = 0.0;
  if (P.n < 8) {  // matching.py:881-882
    for (int i = 0; i < 12; i++) out.ransac.model[i] = out.ransac.lo_model[i] = 0.0;
    out.ransac.best_score = out.ransac.iterations_run = 0;
    return 0;
  }
  return 1;
}
// Part 2 (after the RANSAC, which left its result in out.ransac)
template <class W>
OSFM_HD void robust_match_finish_wave(W& w, WaveShared& s, const PairWork& P, const RansacParams& prm, int refine_iterations, MatchResult& out) {
  double R[9], t[3];
  {  // multiview.relative_pose_ransac (multiview.py:494-516): R = R_lo^T, t = -R_lo^T t_lo
    const double* lo = out.ransac.lo_model;
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) R[3 * a + b] = lo[4 * b + a];
      t[a] = -(lo[a] * lo[3] + lo[4 + a] * lo[7] + lo[8 + a] * lo[11]);
    }
  }
  const double relax[3] = {4.0, 2.0, 1.0};
  for (int stage = 0; stage < 3; stage++) {
    const double th = relax[stage] * prm.threshold_angle;
    const int cnt = w.compact(P.n, [&](int i) { return inlier_bearing(P.b1 + 3 * i, P.b2 + 3 * i, R, t, th) != 0; }, P.subset);
    if (cnt < 8) return;
    double RT[12];  // multiview.relative_pose_optimize_nonlinear (multiview.py:541-553)
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) RT[4 * a + b] = R[3 * b + a];
      RT[4 * a + 3] = -(R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2]);
    }
    w.single([&]() { refinement_picks(cnt, s.picked); });
    WaveRefineEval<W> ev{w, s, P.b1, P.b2, P.subset};
    refine_relative_pose(RT, refine_iterations, ev, (double*)nullptr);
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) R[3 * a + b] = RT[4 * b + a];
      t[a] = -(RT[a] * RT[3] + RT[4 + a] * RT[7] + RT[8 + a] * RT[11]);
    }
  }
  out.n_inliers =
      w.compact(P.n, [&](int i) { return inlier_bearing(P.b1 + 3 * i, P.b2 + 3 * i, R, t, prm.threshold_angle) != 0; }, P.subset);
  for (int i = 0; i < 9; i++) out.R[i] = R[i];
  for (int i = 0; i < 3; i++) out.t[i] = t[i];
}
template <class W>
OSFM_HD void robust_match_calibrated_wave(W& w, WaveShared& s, const PairWork& P, const RansacParams& prm, int refine_iterations,
                                          MatchResult& out) {
  if (!robust_match_begin(P, out)) return;
  ransac_relative_pose_wave(w, s, P, prm, out.ransac);
  robust_match_finish_wave(w, s, P, prm, refine_iterations, out);
}

}  // namespace osfm_rp
