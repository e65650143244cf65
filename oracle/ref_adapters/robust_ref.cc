// robust_ref.cc -- TEST INFRASTRUCTURE: runs the REFERENCE's own LO-RANSAC loop on this box.
//
// Compiled (oracle/Makefile, target _ref/librobust_ref.so) against the reference's headers where they lie:
//   /root/reference/opensfm/src/robust/robust_estimator.h   Estimate<SCORING, MODEL>, ShouldStop
//   /root/reference/opensfm/src/robust/random_sampler.h     RandomSamplesGenerator<std::mt19937> (+ this toolchain's libstdc++
//                                                           std::uniform_int_distribution)
//   /root/reference/opensfm/src/robust/scorer.h             RansacScoring, ScoreInfo (std::max on ties!)
// No reference source is copied.  The reference's RelativePose model (relative_pose_model.h) needs Eigen, which this image does
// not have, so the MODEL plugged into the reference's template is an adapter over the oracle's own restatement of the five-point
// solver, the pose decomposition, the N-point solver and the error (relpose_oracle.c).  What this pins is therefore everything
// AROUND the model numerics: the sampler and its distribution as this toolchain compiles them, the order of draws, the handling
// of ties, the local-optimisation loop, the stopping rule -- i.e. the decision sequence the GPU kernel has to reproduce.
#include <array>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <utility>
#include <vector>

#include "robust_estimator.h"

extern "C" {
int oracle_essential_five_points(const double* b1, const double* b2, double* Es);
int oracle_relative_pose_from_essential(const double* E, const double* b1, const double* b2, int n, double* RT);
int oracle_essential_n_points_contiguous(const double* b1, const double* b2, int count, double* E);
double oracle_relpose_error(const double* RT, const double* x, const double* y);
}

namespace {
struct Err {
  double v;
  double norm() const { return std::sqrt(v * v); }  // Eigen::Matrix<double, 1, 1>::norm()
};
struct OracleRelativePose {
  using Type = std::array<double, 12>;
  using Data = std::pair<std::array<double, 3>, std::array<double, 3>>;
  static const int MINIMAL_SAMPLES = 5;
  static const int MAX_MODELS = 10;
  template <class IT>
  static void gather(IT begin, IT end, std::vector<double>& x, std::vector<double>& y) {
    for (IT it = begin; it != end; ++it)
      for (int a = 0; a < 3; a++) {
        x.push_back(it->first[a]);
        y.push_back(it->second[a]);
      }
  }
  template <class IT>
  static int Estimate(IT begin, IT end, Type* models) {
    std::vector<double> x, y;
    gather(begin, end, x, y);
    double Es[90];
    const int n = oracle_essential_five_points(x.data(), y.data(), Es);
    for (int i = 0; i < n; i++) {
      models[i].fill(0.0);
      oracle_relative_pose_from_essential(Es + 9 * i, x.data(), y.data(), 5, models[i].data());
    }
    return n;
  }
  template <class IT>
  static int EstimateNonMinimal(IT begin, IT end, Type* models) {
    std::vector<double> x, y;
    gather(begin, end, x, y);
    double E[9];
    const int count = (int)(x.size() / 3);
    if (!oracle_essential_n_points_contiguous(x.data(), y.data(), count, E)) return 0;
    models[0].fill(0.0);
    oracle_relative_pose_from_essential(E, x.data(), y.data(), count, models[0].data());
    return 1;
  }
  template <class IT>
  static std::vector<Err> EvaluateModel(const Type& model, IT begin, IT end) {
    std::vector<Err> errors;
    for (IT it = begin; it != end; ++it) errors.push_back(Err{oracle_relpose_error(model.data(), it->first.data(), it->second.data())});
    return errors;
  }
};
}  // namespace

extern "C" int ref_ransac_relative_pose(const double* b1, const double* b2, int n, double threshold_angle, int iterations, double probability,
                                        int use_lo, int lo_iterations, double* model, double* lo_model, int* inliers) {
  std::vector<OracleRelativePose::Data> samples(n);
  for (int i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) {
      samples[i].first[a] = b1[3 * i + a];
      samples[i].second[a] = b2[3 * i + a];
    }
  RobustEstimatorParams params;
  params.iterations = iterations;
  params.probability = probability;
  params.use_local_optimization = use_lo != 0;
  params.local_optimization_iterations = lo_iterations;
  RansacScoring scorer(1.0 - std::cos(threshold_angle));  // RelativePose::ThresholdAdapter (relative_pose_model.h:18-20)
  const auto best = Estimate<RansacScoring, OracleRelativePose>(samples, scorer, params);
  std::memcpy(model, best.model.data(), 12 * sizeof(double));
  std::memcpy(lo_model, best.lo_model.data(), 12 * sizeof(double));
  for (size_t i = 0; i < best.inliers_indices.size(); i++) inliers[i] = best.inliers_indices[i];
  return (int)best.score;
}

// the first `count` samples of size `size` out of `n` that RandomSamplesGenerator<std::mt19937>(42) hands out, as indices
namespace {
struct IndexModel {
  using Data = int;
};
}  // namespace
extern "C" void ref_random_samples(int n, int size, int count, int* out) {
  std::vector<int> samples(n);
  for (int i = 0; i < n; i++) samples[i] = i;
  RandomSamplesGenerator<std::mt19937> gen;
  for (int c = 0; c < count; c++) {
    const auto s = gen.GetRandomSamples<IndexModel>(samples, size);
    for (int k = 0; k < size; k++) out[c * size + k] = s[k];
  }
}
