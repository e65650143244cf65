// relpose_rounds.h -- calibrated (essential-matrix) LO-RANSAC for MANY image pairs at once, organised in rounds.
//
// reference: robust::Estimate<RansacScoring, RelativePose> (opensfm/src/robust/robust_estimator.h:37-119) as called by
// matching.robust_match_calibrated (opensfm/matching.py:871-903) through pyrobust.ransac_relative_pose.
//
// The reference's loop is sequential per pair: iteration i + 1 samples from the generator state iteration i left behind, and the
// local optimisation (LO) draws too whenever a model ties or beats the best score.  What is independent is (a) the pairs and
// (b) the minimal / non-minimal SOLVES once their samples are known.  So the work is cut into three kinds of kernels that all
// pairs go through together, round after round, until every pair has stopped:
//
//   walk   one wavefront per pair: scores the models that are ready, in the reference's order, with its decision rules
//          (std::max keeps the newcomer on ties, LO fires on every new / tied best with >= 5 inliers, ShouldStop).  When it
//          needs models it does not have -- the next iterations' five-point solutions, or the non-minimal models of the LO
//          chain -- it DRAWS their samples (the draws only need the generator position and the current inlier list, not the
//          solutions), appends the problems to a work list and yields.
//   solve5 one LANE per five-point problem of the list, whatever pair it belongs to: full wavefronts of identical,
//          divergence-tolerant sequential code.  Two stages with different work spaces: A (null space, cubic constraints,
//          Gauss-Jordan; 236 doubles per lane in LDS, one wavefront per CU) and B (Hessenberg-QR eigenvalues, eigenvectors;
//          100 doubles per lane, three wavefronts per CU) -- the long, latency-bound part runs on three SIMDs per CU.
//   solveN one lane per non-minimal (N-point) problem: 9 x 9 cyclic Jacobi entirely in registers (no LDS, four wavefronts per CU).
//   pose   one lane per essential matrix found by either solver: RelativePoseFromEssential (SVD, cheirality vote) -- registers
//          only, many wavefronts per SIMD to hide the division / square-root chains.
//
// Speculation: the samples of the next `width` iterations are drawn assuming no LO fires in between (LO moves the generator);
// the LO chain's samples are drawn assuming no LO model improves the best (an improvement changes the inlier list the later
// samples are drawn from).  A wrong guess only discards solutions; the generator position every decision continues from is
// recorded per sample, so the sequence of draws the reference would make is reproduced exactly.
//
// std::mt19937(42) is the same stream for every pair: its raw outputs are tabulated once (RngTable) and a generator "state" is
// an index into the table -- nothing to snapshot, rewind or replay.
//
// All functions are host + device: tests/native/relpose_core_host.cpp runs the very same round logic with loops in place of
// wavefronts and compares every result bit for bit with the CPU oracle.
#pragma once
#include "relpose_core.h"

namespace osfm_rp {

constexpr int kWave = 64;
constexpr int kMaxModels = 10;   // RelativePose::MAX_MODELS
constexpr int kMaxSlots = 16;    // speculative main iterations per round, at most
constexpr int kLoSampleMax = 12; // lo_sample_size_clamp

struct RngTable {
  const uint32_t* tab;  // raw outputs of std::mt19937(42), in order
  int size;
};
constexpr int kRngCache = 512;  // stream entries a walk stages next to itself before it draws (LDS on the GPU)
constexpr int kLoIterMax = 64;  // local_optimization_iterations the per-pair scratch is sized for (reference default: 10)

// A window of the stream: entries [cache_pos, cache_pos + cache_n) come from `cache`, everything else from the table
struct RngView {
  RngTable T;
  const uint32_t* cache;
  int cache_pos, cache_n;
  OSFM_HD uint32_t get(int i) const {
    const unsigned k = (unsigned)(i - cache_pos);
    return k < (unsigned)cache_n ? cache[k] : T.tab[i];
  }
};

// RandomSamplesGenerator::GenerateOneSample (robust/random_sampler.h:27-37) on the tabulated stream: `size` distinct indices in
// [0, n) with std::uniform_int_distribution as libstdc++ >= 11 draws it (Lemire's multiply-shift with rejection).  Returns the
// stream position after the sample; *overflow is set when the table is too short (the caller then reports it).
OSFM_HD int draw_sample_tab(const RngView& V, int pos, int size, int n, int* idx, int* overflow) {
  for (int i = 0; i < size; i++) {
    int dup;
    do {
      const uint32_t range = (uint32_t)n;
      if (pos >= V.T.size) {
        *overflow = 1;
        for (int q = i; q < size; q++) idx[q] = q < n ? q : 0;
        return pos;
      }
      uint64_t product = (uint64_t)V.get(pos++) * (uint64_t)range;
      uint32_t low = (uint32_t)product;
      if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
          if (pos >= V.T.size) {
            *overflow = 1;
            for (int q = i; q < size; q++) idx[q] = q < n ? q : 0;
            return pos;
          }
          product = (uint64_t)V.get(pos++) * (uint64_t)range;
          low = (uint32_t)product;
        }
      }
      idx[i] = (int)(uint32_t)(product >> 32);
      dup = 0;
      for (int j = 0; j < i; j++) dup |= idx[j] == idx[i];
    } while (dup);
  }
  return pos;
}
OSFM_HD int draw_sample_tab(const RngTable& T, int pos, int size, int n, int* idx, int* overflow) {
  const RngView V{T, nullptr, 0, 0};
  return draw_sample_tab(V, pos, size, n, idx, overflow);
}

// per-pair scratch of a walk (LDS on the GPU)
struct WalkShared {
  uint32_t rng[kRngCache];
  int pick[kLoIterMax * kLoSampleMax];
  int nlo, nmain, first_slot;  // what lane 0 drew, for the lanes that write it out
};

enum { kPhaseMain = 0, kPhaseLo = 1, kPhaseDone = 2 };

struct PairState {
  int phase;
  int pos;                  // generator position (index into the table) the next draw continues from
  int it;                   // index of the RANSAC iteration being walked
  int best_score, best_n;   // score of the best model and length of its inlier list
  int width;                // speculative main iterations of the next draw
  int nslots;               // main iterations whose models are ready (slots 0 .. nslots - 1 of this round)
  int cur_slot, cur_model;  // resume point of the walk
  int lo_l;                 // next LO iteration
  int lo_first;             // LO iterations lo_first .. lo_iterations - 1 were drawn this round
  int lo_size;              // sample size of the LO problems drawn this round
  int stop;
  int rejected;             // MATCH mode: fewer than 8 correspondences (matching.py:881-882)
  double model[12], lo_model[12];
};

// Everything the rounds work on; all pointers are global memory on the GPU.
struct Rounds {
  const double* b1;  // the bearings as handed over (relpose_error normalises them, as RelativePose::Evaluate does)
  const double* b2;
  double* u1;  // the same bearings divided by their norms (pair_normalise): what every model is scored on
  double* u2;
  const int64_t* offsets;  // n_pairs + 1: pair p owns correspondences offsets[p] .. offsets[p + 1] - 1
  int n_pairs;
  const double* stop_bound;  // ShouldStop's bound for every possible best inlier count, tabulated by the host (max_iterations_for):
  const int64_t* stop_off;   // pair p with n correspondences reads the n + 1 doubles at stop_bound + stop_off[p] (one table per
                             // distinct n, shared by the pairs of that size)
  RngTable rng;
  double thr_score;  // 1 - cos(threshold angle), from the host libm
  int iterations, use_lo, lo_iterations;
  int min_n;         // 5 (RANSAC mode) or 8 (MATCH mode: the len(matches) < 8 gate)
  int max_width;     // cap of the speculative width, 1 .. kMaxSlots (a tuning knob: results do not depend on it)
  PairState* st;
  int* sidx;          // [pair][slot][5]
  int* pos_before;    // [pair][slot]: generator position the slot's sample was drawn from -- the slot is valid iff the walk
                      //               arrives at it with the generator exactly there
  int* pos_after;     // [pair][slot]
  int* nmodels;       // [pair][slot]
  double* models;     // [pair][slot][10][12]
  int* lidx;          // [pair][lo][12]
  int* lo_pos_after;  // [pair][lo]
  int* lo_ok;         // [pair][lo]
  double* lo_rt;      // [pair][lo][12]
  int* inliers;       // per pair at offsets[p]: inlier list of the best score (ascending)
  // between the solver stages (indexed by the position k in the round's work list)
  double* s5_at6;     // [k / 64][60][64]
  double* s5_basis;   // [k / 64][36][64]
  int* s5_ok;         // [k]
  double* s5_E;       // [k][10][9]
  double* lo_E;       // [k][9]
  int* list5;         // work list of five-point problems: pair * kMaxSlots + slot
  int* listN;         // work list of N-point problems: pair * lo_iterations + l
  int* counters;      // [0] five-point problems, [1] N-point problems, [2] pairs still running, [3] table overflow
  // stage B in two parts (round 6): the real eigenvalues of the action matrix, then one solution per (problem, eigenvalue)
  double* s5_wr;      // [k][10] real eigenvalues in the order the iteration deflates them
  int* s5_nreal;      // [k] how many (0 when stage A failed)
  int* s5_valid;      // [k][10] 1 where s5_E[k][e] holds a solution
};

// ---- the solver stages.  k = position in the round's work list; problems are grouped in blocks of 64 (one per lane), and what one
//      stage hands to the next lies in global memory "element-major, lane-minor" inside a block (LaneArr stride 64): the loads
//      and stores of a wavefront are then contiguous. ----

// solve5 stage A: sample -> null space + rows 0..5 of the action matrix (relpose_core.h five_point_action_matrix)
template <class D, class I>
OSFM_HD void solve5_stage_a(const Rounds& R, int k, D basis, D M, I colperm) {
  const int item = R.list5[k];
  const int p = item / kMaxSlots;
  const int64_t o = R.offsets[p];
  const double *b1 = R.b1 + 3 * o, *b2 = R.b2 + 3 * o;
  const int* s = R.sidx + (size_t)item * 5;
  double s1[15], s2[15];
  for (int q = 0; q < 5; q++) {
    const int m = s[q];
    for (int a = 0; a < 3; a++) {
      s1[3 * q + a] = b1[3 * m + a];
      s2[3 * q + a] = b2[3 * m + a];
    }
  }
  const int ok = five_point_action_matrix(s1, s2, basis, M, colperm);
  R.s5_ok[k] = ok;
  if (!ok) return;
  const LaneArr<double, kWave> at6{R.s5_at6 + (size_t)(k / kWave) * 60 * kWave + k % kWave};
  const LaneArr<double, kWave> bas{R.s5_basis + (size_t)(k / kWave) * 36 * kWave + k % kWave};
  OSFM_UNROLL for (int i = 0; i < 60; i++) at6[i] = M[i];
  OSFM_UNROLL for (int i = 0; i < 36; i++) bas[i] = basis[i];
}

// solve5 stage B1: action matrix -> its real eigenvalues (Hessenberg + Francis QR), by ONE lane; S: 100 doubles of work space
template <class D>
OSFM_HD void solve5_stage_b1(const Rounds& R, int k, D S) {
  int n = 0;
  if (R.s5_ok[k]) {
    const LaneArr<const double, kWave> at6{R.s5_at6 + (size_t)(k / kWave) * 60 * kWave + k % kWave};
    OSFM_UNROLL for (int i = 0; i < 100; i++) S[i] = action_matrix_entry(at6, i);
    double* wr = R.s5_wr + (size_t)k * kMaxModels;
    n = real_eigenvalues10_put(S, [&](int q, double val) { wr[q] = val; });
  }
  R.s5_nreal[k] = n;
}
// ... the same by a group of kEigGroup lanes (real_eigenvalues10_group): `a` = 100 doubles the group shares, glane = 0 .. 15.
// This is what the GPU runs (rp_eig5_kernel); the host harness runs both and the results must agree bit for bit.
template <class PA>
OSFM_HD void solve5_stage_b1_group(const Rounds& R, int k, PA a, int glane) {
  int n = 0;
  if (R.s5_ok[k]) {
    const LaneArr<const double, kWave> at6{R.s5_at6 + (size_t)(k / kWave) * 60 * kWave + k % kWave};
    OSFM_GROUP_FOR(j, glane)
      for (int i = j; i < 100; i += kEigGroup) a[i] = action_matrix_entry(at6, i);
    OSFM_GROUP_SYNC();
    double* wr = R.s5_wr + (size_t)k * kMaxModels;
    n = real_eigenvalues10_group(a, glane, [&](int q, double val) {
      OSFM_GROUP_FOR(j, glane) if (j == 0) wr[q] = val;
    });
  }
  OSFM_GROUP_FOR(j, glane) if (j == 0) R.s5_nreal[k] = n;
}
// solve5 stage B2: lane q = (problem k, eigenvalue e) -> the essential matrix of that eigenvalue, if it has one; S: 100 doubles
template <class D>
OSFM_HD void solve5_stage_b2(const Rounds& R, int q, D S) {
  const int k = q / kMaxModels, e = q % kMaxModels;
  int valid = 0;
  if (R.s5_ok[k] && e < R.s5_nreal[k]) {
    const LaneArr<const double, kWave> at6{R.s5_at6 + (size_t)(k / kWave) * 60 * kWave + k % kWave};
    const LaneArr<const double, kWave> bas{R.s5_basis + (size_t)(k / kWave) * 36 * kWave + k % kWave};
    double Em[9];
    if (five_point_solution_at(at6, bas, S, R.s5_wr[q], Em)) {
      valid = 1;
      double* out = R.s5_E + (size_t)q * 9;
      for (int i = 0; i < 9; i++) out[i] = Em[i];
    }
  }
  R.s5_valid[q] = valid;
}

// pose stage, five-point side: lane q = (problem k, eigenvalue e) -> RelativePoseFromEssential on the sample.  The models of a slot are the
// solutions in eigenvalue order with the failed ones left out: model index = the number of valid solutions before e
OSFM_HD void pose5_item(const Rounds& R, int q) {
  const int k = q / kMaxModels, e = q % kMaxModels;
  const int item = R.list5[k];
  const int* valid = R.s5_valid + (size_t)k * kMaxModels;
  int j = 0, total = 0;
  for (int i = 0; i < kMaxModels; i++) {
    j += i < e ? valid[i] : 0;
    total += valid[i];
  }
  if (e == 0) R.nmodels[item] = total;
  if (!valid[e]) return;
  const int p = item / kMaxSlots;
  const int64_t o = R.offsets[p];
  const double *b1 = R.b1 + 3 * o, *b2 = R.b2 + 3 * o;
  const int* s = R.sidx + (size_t)item * 5;
  double E[9], RT[12];
  for (int i = 0; i < 9; i++) E[i] = R.s5_E[(size_t)q * 9 + i];
  for (int i = 0; i < 12; i++) RT[i] = 0.0;
  relative_pose_from_essential(E, b1, b2, s, 5, RT);  // the sample's bearings straight from the pair's arrays
  double* out = R.models + ((size_t)item * kMaxModels + j) * 12;
  for (int i = 0; i < 12; i++) out[i] = RT[i];
}

// ---- solveN: one non-minimal problem (pair, l) -> 0 or 1 essential matrix; AtA, V: 81 doubles each, w: 9 ----
template <class D>
OSFM_HD void solveN_problem(const Rounds& R, int k, D AtA, D V, D w) {
  const int item = R.listN[k];
  const int p = item / R.lo_iterations;
  const int64_t o = R.offsets[p];
  const double *b1 = R.b1 + 3 * o, *b2 = R.b2 + 3 * o;
  const int lo_size = R.st[p].lo_size;
  int idx[kLoSampleMax];
  for (int q = 0; q < kLoSampleMax; q++) idx[q] = q < lo_size ? R.lidx[(size_t)item * kLoSampleMax + q] : 0;
  double E[9];
  const int ok = essential_n_points_ws(b1, b2, idx, lo_size, E, AtA, V, w);
  if (ok)
    for (int i = 0; i < 9; i++) R.lo_E[(size_t)k * 9 + i] = E[i];
  R.lo_ok[item] = ok;
}
// pose stage, N-point side: lane k -> RelativePoseFromEssential on the LO sample
OSFM_HD void poseN_item(const Rounds& R, int k) {
  const int item = R.listN[k];
  if (!R.lo_ok[item]) return;
  const int p = item / R.lo_iterations;
  const int64_t o = R.offsets[p];
  const double *b1 = R.b1 + 3 * o, *b2 = R.b2 + 3 * o;
  const int lo_size = R.st[p].lo_size;
  double E[9], RT[12];
  for (int i = 0; i < 9; i++) E[i] = R.lo_E[(size_t)k * 9 + i];
  for (int i = 0; i < 12; i++) RT[i] = 0.0;
  relative_pose_from_essential(E, b1, b2, R.lidx + (size_t)item * kLoSampleMax, lo_size, RT);
  for (int i = 0; i < 12; i++) R.lo_rt[(size_t)item * 12 + i] = RT[i];
}

// ---- walk: one pair, one wavefront (policy W: single / parallel_for / count_if / compact / atomic_add / stage_rng, see
//      relpose.hip and the host test) ----
template <class W>
OSFM_HD void pair_walk(W& w, WalkShared& sh, const Rounds& R, int p) {
  PairState& S = R.st[p];
  if (S.phase == kPhaseDone) return;
  const int64_t o = R.offsets[p];
  const int n = (int)(R.offsets[p + 1] - o);
  const double *b1 = R.u1 + 3 * o, *b2 = R.u2 + 3 * o;  // unit bearings: the scores; the samples of the solvers use R.b1 / R.b2
  int* inliers = R.inliers + o;
  const double* stop_bound = R.stop_bound + R.stop_off[p];
  const double thr = R.thr_score;
  // the state is wave-uniform: every lane keeps its own copy in registers, lane 0 writes it back
  int phase = S.phase, pos = S.pos, it = S.it, best_score = S.best_score, best_n = S.best_n, width = S.width, nslots = S.nslots;
  int cur_slot = S.cur_slot, cur_model = S.cur_model, lo_l = S.lo_l, stop = S.stop;
  const int lo_first = S.lo_first;
  double model[12], lo_model[12];
  for (int i = 0; i < 12; i++) {
    model[i] = S.model[i];
    lo_model[i] = S.lo_model[i];
  }
  auto score = [&](const double* RT) { return w.count_if(n, [&](int i) { return fabs(relpose_error_unit(RT, b1 + 3 * i, b2 + 3 * i)) < thr; }); };
  // take: the inlier list of a new best; returns whether the list differs from the one it replaces (previous length prev_n)
  auto take = [&](const double* RT, int prev_n) {
    int changed = 0;
    const int c = w.compact_changed(n, [&](int i) { return fabs(relpose_error_unit(RT, b1 + 3 * i, b2 + 3 * i)) < thr; }, inliers, &changed);
    return changed || c != prev_n;
  };
  int request = 0;  // 1: five-point problems only, 2: the LO chain (+ the main iterations that follow it)
  while (true) {
    if (phase == kPhaseLo) {
      // the models of LO iterations lo_first .. are ready; they were drawn from the inlier list as it is now
      int improved = 0;
      int last_pos = pos;
      for (int l = (lo_l > lo_first ? lo_l : lo_first); l < R.lo_iterations; l++) {
        const size_t item = (size_t)p * R.lo_iterations + l;
        last_pos = R.lo_pos_after[item];
        lo_l = l + 1;
        if (!R.lo_ok[item]) continue;
        double RTlo[12];
        for (int i = 0; i < 12; i++) RTlo[i] = R.lo_rt[item * 12 + i];
        const int c2 = score(RTlo);
        if (c2 >= best_score) {  // std::max(lo_score, best_score): ties keep the newcomer
          const int prev_n = best_n;
          best_score = c2;
          best_n = c2;
          const int list_changed = take(RTlo, prev_n);
          for (int i = 0; i < 12; i++) lo_model[i] = RTlo[i];
          // the remaining LO samples were drawn from the previous inlier list: they are the reference's samples iff the list is
          // the same (a tie with the same inliers is the common case once the RANSAC has converged)
          if (list_changed) {
            improved = 1;
            break;
          }
        }
      }
      pos = last_pos;
      if (improved && lo_l < R.lo_iterations) {  // the remaining LO samples must be drawn from the new inlier list
        request = 2;
        break;
      }
      stop = stop_bound[best_n] < (double)it;  // ShouldStop at the end of the model that triggered the LO
      phase = kPhaseMain;
    }
    // main walk over the slots that are ready
    int trigger = 0;
    while (cur_slot < nslots && it < R.iterations) {  // (a stop raised by the LO chain still closes the iteration it is in)
      const size_t item = (size_t)p * kMaxSlots + cur_slot;
      if (cur_model == 0) {
        // entering an iteration: its sample is the one the reference draws here iff it was drawn from where the generator stands
        // now (speculation past an LO chain that did improve, or past a trigger, was drawn from somewhere else)
        if (R.pos_before[item] != pos) {
          nslots = cur_slot;
          break;
        }
        pos = R.pos_after[item];
      }  // (resuming an iteration after its LO chain -- cur_model > 0 -- the generator stands where the chain left it)
      const int nm = R.nmodels[item];
      while (cur_model < nm && !stop) {
        double RT[12];
        for (int i = 0; i < 12; i++) RT[i] = R.models[(item * kMaxModels + cur_model) * 12 + i];
        cur_model++;
        const int cnt = score(RT);
        if (cnt >= best_score) {
          const int prev_n = best_n;
          best_score = cnt;
          best_n = cnt;
          (void)take(RT, prev_n);
          for (int i = 0; i < 12; i++) model[i] = lo_model[i] = RT[i];
        }
        const int best_found = (cnt == best_score) && cnt >= 5;
        if (best_found && R.use_lo && R.lo_iterations > 0) {
          phase = kPhaseLo;
          lo_l = 0;
          trigger = 1;
          break;
        }
        stop = stop_bound[best_n] < (double)it;
      }
      if (trigger) break;
      cur_slot++;
      cur_model = 0;
      it++;
      if (stop) break;
    }
    if (trigger) {
      request = 2;
      // new bests come in bursts at the start and become rare later: speculate little right after an early one, more later
      width = it / 2 + 2;
      break;
    }
    if (stop || it >= R.iterations) {
      phase = kPhaseDone;
      break;
    }
    request = 1;  // every ready slot is used up (or void): the next samples continue from where the generator stands
    break;
  }
  // ---- draws: lane 0 walks the stream (staged next to it), all lanes write the problems out ----
  const RngView V = w.stage_rng(R.rng, sh.rng, pos, request != 0);
  int lo_size = S.lo_size, new_lo_first = S.lo_first;
  if (request == 2) {
    lo_size = (int)(best_n * 0.5);  // Matas: min(inliers / 2, 12), at least the minimal sample
    if (lo_size > kLoSampleMax) lo_size = kLoSampleMax;
    if (lo_size < 5) lo_size = 5;
    new_lo_first = lo_l;
  }
  w.single([&]() {
    int overflow = 0;
    int q = pos;
    sh.nlo = sh.nmain = 0;
    sh.first_slot = 0;
    if (request == 2) {
      for (int l = lo_l; l < R.lo_iterations; l++) {
        q = draw_sample_tab(V, q, lo_size, best_n, sh.pick + (l - lo_l) * kLoSampleMax, &overflow);
        R.lo_pos_after[(size_t)p * R.lo_iterations + l] = q;
      }
      sh.nlo = R.lo_iterations - lo_l;
      sh.first_slot = cur_slot + 1;  // the iteration in progress keeps its slot: its remaining models are still to be walked
    }
    if (request != 0) {
      // main iterations from where the generator stands (request 1) or will stand if no LO model improves the best (request 2)
      int B = width < 1 ? 1 : (width > R.max_width ? R.max_width : width);
      const int it_next = request == 2 ? it + 1 : it;
      if (B > R.iterations - it_next) B = R.iterations - it_next;
      if (B > kMaxSlots - sh.first_slot) B = kMaxSlots - sh.first_slot;
      if (B < 0) B = 0;
      for (int k = 0; k < B; k++) {
        const size_t item = (size_t)p * kMaxSlots + sh.first_slot + k;
        R.pos_before[item] = q;
        q = draw_sample_tab(V, q, 5, n, R.sidx + item * 5, &overflow);
        R.pos_after[item] = q;
      }
      sh.nmain = B;
      if (request == 1) width = 2 * B;
    }
    if (overflow) R.counters[3] = 1;
  });
  const int nlo = sh.nlo, nmain = sh.nmain, first_slot = sh.first_slot;
  if (request == 2) {
    w.parallel_for(nlo * kLoSampleMax, [&](int t) {
      const int l = lo_l + t / kLoSampleMax, k = t % kLoSampleMax;
      if (k < lo_size) R.lidx[((size_t)p * R.lo_iterations + l) * kLoSampleMax + k] = inliers[sh.pick[t]];
    });
  }
  if (request != 0) {
    nslots = first_slot + nmain;
    if (request == 1) {
      cur_slot = 0;
      cur_model = 0;
    }
  }
  w.single([&]() {
    if (nlo > 0) {
      const int base = w.atomic_add(&R.counters[1], nlo);
      for (int k = 0; k < nlo; k++) R.listN[base + k] = p * R.lo_iterations + lo_l + k;
    }
    if (nmain > 0) {
      const int base = w.atomic_add(&R.counters[0], nmain);
      for (int k = 0; k < nmain; k++) R.list5[base + k] = p * kMaxSlots + first_slot + k;
    }
    if (phase != kPhaseDone) w.atomic_add(&R.counters[2], 1);
    S.phase = phase;
    S.pos = pos;
    S.it = it;
    S.best_score = best_score;
    S.best_n = best_n;
    S.width = width;
    S.nslots = nslots;
    S.cur_slot = cur_slot;
    S.cur_model = cur_model;
    S.lo_l = lo_l;
    S.lo_first = new_lo_first;
    S.lo_size = lo_size;
    S.stop = stop;
    for (int i = 0; i < 12; i++) {
      S.model[i] = model[i];
      S.lo_model[i] = lo_model[i];
    }
  });
}

// correspondence k (global index): unit bearings for the scoring
OSFM_HD void pair_normalise(const Rounds& R, int64_t k) {
  normalise_bearing(R.b1 + 3 * k, R.u1 + 3 * k);
  normalise_bearing(R.b2 + 3 * k, R.u2 + 3 * k);
}

OSFM_HD void pair_init(const Rounds& R, int p) {
  PairState& S = R.st[p];
  const int n = (int)(R.offsets[p + 1] - R.offsets[p]);
  S.phase = n < R.min_n ? kPhaseDone : kPhaseMain;
  S.rejected = n < R.min_n;
  S.pos = 0;
  S.it = 0;
  S.best_score = S.best_n = 0;
  S.width = 1;  // the very first model always becomes the best: no point in speculating past it
  S.nslots = S.cur_slot = S.cur_model = 0;
  S.lo_l = S.lo_first = 0;
  S.lo_size = 5;
  S.stop = 0;
  for (int i = 0; i < 12; i++) S.model[i] = S.lo_model[i] = 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
// After the RANSAC: the rest of robust_match_calibrated (matching.py:886-903), one wavefront per pair.
// ---------------------------------------------------------------------------------------------------------------
struct RefineShared {
  int picked[kRefineResiduals];
  double rbuf[kRefineResiduals + 1][7];  // residual + gradient of the refinement
  double sums[64];                       // results of WaveRefineEval::reduce
  double keep[48];                       // the driver's wave-uniform arrays that outlive an evaluation (J^T J): LDS instead of 72 registers
};

// Evaluator of the refinement residuals over the wavefront (see refine_relative_pose in relpose_core.h)
template <class W>
struct WaveRefineEval {
  W& w;
  RefineShared& s;
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
  OSFM_HD double val(int i, int c) const { return s.rbuf[i][c]; }  // column 0: the residual, 1 + k: its derivative k
  // 48 doubles of the driver's own, the same value in every lane (what one lane stores every lane loads: wave-uniform)
  OSFM_HD void keep(int k, double v) { s.keep[k] = v; }
  OSFM_HD double kept(int k) const { return s.keep[k]; }
  // out[q] = sum over i = 0 .. 100 (in order, from 0.0) of term(q, i), q < nsums <= 64: one sum per lane.  The terms of eight
  // consecutive i are formed before they are added (in order): their LDS reads are then in flight together instead of one round trip
  // per addition -- the sum itself is the same chain of additions
  template <class F>
  OSFM_HD void reduce(int nsums, F term, double* out) {
    w.parallel_for(nsums, [&](int q) {
      double acc = 0.0;
      int i = 0;
      for (; i + 8 <= kRefineResiduals + 1; i += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = term(q, i + u);
#pragma unroll
        for (int u = 0; u < 8; u++) acc += t[u];
      }
      for (; i < kRefineResiduals + 1; i++) acc += term(q, i);
      s.sums[q] = acc;
    });
    for (int q = 0; q < nsums; q++) out[q] = s.sums[q];
  }
};

// 3 x (inliers at 4, 2, 1 x threshold -> TinySolver refinement on 100 rand()-picked inliers), final inliers.
// lo_model: ScoreInfo::lo_model of the RANSAC.  Returns the number of inliers written to subset (ascending), R / t = pose of the
// second camera in the first; 0 (and R = t = 0) where the reference returns an empty array.
template <class W>
OSFM_HD int robust_match_finish_wave(W& w, RefineShared& s, const double* b1, const double* b2, int n, const double* lo_model,
                                     double threshold_angle, int refine_iterations, int* subset, double* R_out, double* t_out) {
  double R[9], t[3];
  for (int i = 0; i < 9; i++) R_out[i] = 0.0;
  for (int i = 0; i < 3; i++) t_out[i] = 0.0;
  {  // multiview.relative_pose_ransac (multiview.py:494-516): R = R_lo^T, t = -R_lo^T t_lo
    const double* lo = lo_model;
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) R[3 * a + b] = lo[4 * b + a];
      t[a] = -(lo[a] * lo[3] + lo[4 + a] * lo[7] + lo[8 + a] * lo[11]);
    }
  }
  const double relax[3] = {4.0, 2.0, 1.0};
  for (int stage = 0; stage < 3; stage++) {
    const double th = relax[stage] * threshold_angle;
    const int cnt = w.compact(n, [&](int i) { return inlier_bearing(b1 + 3 * i, b2 + 3 * i, R, t, th) != 0; }, subset);
    if (cnt < 8) return 0;
    double RT[12];  // multiview.relative_pose_optimize_nonlinear (multiview.py:541-553)
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) RT[4 * a + b] = R[3 * b + a];
      RT[4 * a + 3] = -(R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2]);
    }
    w.single([&]() { refinement_picks(cnt, s.picked); });
    WaveRefineEval<W> ev{w, s, b1, b2, subset};
    refine_relative_pose(RT, refine_iterations, ev, (double*)nullptr);
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) R[3 * a + b] = RT[4 * b + a];
      t[a] = -(RT[a] * RT[3] + RT[4 + a] * RT[7] + RT[8 + a] * RT[11]);
    }
  }
  const int cnt = w.compact(n, [&](int i) { return inlier_bearing(b1 + 3 * i, b2 + 3 * i, R, t, threshold_angle) != 0; }, subset);
  for (int i = 0; i < 9; i++) R_out[i] = R[i];
  for (int i = 0; i < 3; i++) t_out[i] = t[i];
  return cnt;
}

}  // namespace osfm_rp
