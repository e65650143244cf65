// relpose_coop.h -- second organisation of the calibrated LO-RANSAC (opt-in: OSFM_RELPOSE_V2=1), same numbers bit for bit.
//
// Why: in relpose_wave.h every lane solves its own five-point problem, so the 10 x 20 / 10 x 10 matrices live in
// per-lane scratch (0.8 MB per wavefront, HBM-bound) and every local-optimisation trigger throws a 64-wide batch away.  Here ONE
// minimal problem is solved at a time by the whole wavefront on matrices that live in LDS: the element-wise steps (polynomial
// coefficients, Gauss-Jordan updates, the ten eigenvector eliminations) are spread over the lanes, the short sequential parts
// (5 x 9 null space, Hessenberg-QR) run on lane 0 against LDS, and the RANSAC loop is the reference's plain sequential loop
// (robust_estimator.h:37-119) -- no speculation, no generator rewind.
// Every element is computed by exactly the arithmetic of relpose_core.h (slot-wise accumulation orders included), so the results
// are the same doubles; tests/test_relpose_core_host.py checks that against the oracle with the items of every parallel step run
// forwards, backwards and shuffled (no step may depend on the order of its items).
#pragma once
#include "relpose_wave.h"

namespace osfm_rp {

struct CoopShared {
  double s1[15], s2[15];     // the sample's bearings
  double A[45], basis[36];   // epipolar system and its null space: E(i, j) = basis + (3 i + j) * 4
  double Q[3][10];           // the three 2 x 2 minors of the determinant (quadratic)
  double L[9][10];           // E E^T - tr/2 (quadratic)
  double M[200];             // ten cubic constraints
  double fcol[10];           // column snapshot of the elimination step
  double At[100], Aq[100];   // action matrix and its Hessenberg work copy
  double wr[10];
  double S[10][100];         // (At - lambda_e I), one per real eigenvalue
  double fS[10][10], ipv[10];
  int colperm[10][10], pr[10], pc[10], alive[10];
  double Ecand[10][9];
  int valid[10];
  double Es[90], RT[10][12];
  int nreal, count, ok;
};

// contribution order of mul_lin_lin / mul_quad_lin restricted to one output slot (same additions, same order, from 0.0)
OSFM_HD double lin_lin_slot(const double* a, const double* b, int k) {  // k in 0..9 (monomial slot 10 + k)
  double acc = 0.0;
  for (int i = 16; i < 20; i++)
    for (int j = 16; j < 20; j++)
      if (mul_slot(i, j) - 10 == k) acc += a[i - 16] * b[j - 16];
  return acc;
}
OSFM_HD double quad_lin_slot(const double* q, const double* b, int k) {  // k in 0..19
  double acc = 0.0;
  for (int i = 10; i < 20; i++)
    for (int j = 16; j < 20; j++)
      if (mul_slot(i, j) == k) acc += q[i - 10] * b[j - 16];
  return acc;
}

// Five-point solver on the sample in c.s1 / c.s2; returns the number of essential matrices left in c.Es (unit Frobenius norm).
template <class W>
OSFM_HD int essential_five_points_coop(W& w, CoopShared& c) {
  w.parallel_for(45, [&](int t) {
    const int i = t / 9, r = (t % 9) / 3, col = t % 3;
    c.A[t] = c.s2[3 * i + r] * c.s1[3 * i + col];
  });
  w.single([&]() { c.ok = nullspace_5x9(c.A, c.basis); });
  if (!c.ok) return 0;
#define OSFM_E(i, j) (c.basis + (3 * (i) + (j)) * 4)
  w.parallel_for(30, [&](int t) {  // minors of det E
    const int m = t / 10, k = t % 10;
    double qa, qb;
    if (m == 0) {
      qa = lin_lin_slot(OSFM_E(0, 1), OSFM_E(1, 2), k);
      qb = lin_lin_slot(OSFM_E(0, 2), OSFM_E(1, 1), k);
    } else if (m == 1) {
      qa = lin_lin_slot(OSFM_E(0, 2), OSFM_E(1, 0), k);
      qb = lin_lin_slot(OSFM_E(0, 0), OSFM_E(1, 2), k);
    } else {
      qa = lin_lin_slot(OSFM_E(0, 0), OSFM_E(1, 1), k);
      qb = lin_lin_slot(OSFM_E(0, 1), OSFM_E(1, 0), k);
    }
    c.Q[m][k] = qa - qb;
  });
  w.parallel_for(90, [&](int t) {  // L = E E^T
    const int p = t / 10, k = t % 10, i = p / 3, j = p % 3;
    c.L[p][k] = (lin_lin_slot(OSFM_E(i, 0), OSFM_E(j, 0), k) + lin_lin_slot(OSFM_E(i, 1), OSFM_E(j, 1), k)) +
                lin_lin_slot(OSFM_E(i, 2), OSFM_E(j, 2), k);
  });
  w.parallel_for(10, [&](int k) {  // L -= tr(L) / 2 on the diagonal
    const double tr = ((c.L[0][k] + c.L[4][k]) + c.L[8][k]) * 0.5;
    c.L[0][k] -= tr;
    c.L[4][k] -= tr;
    c.L[8][k] -= tr;
  });
  w.parallel_for(200, [&](int t) {  // the ten cubic constraints
    const int row = t / 20, k = t % 20;
    if (row == 0) {
      c.M[k] = (quad_lin_slot(c.Q[0], OSFM_E(2, 0), k) + quad_lin_slot(c.Q[1], OSFM_E(2, 1), k)) + quad_lin_slot(c.Q[2], OSFM_E(2, 2), k);
    } else {
      const int i = (row - 1) / 3, j = (row - 1) % 3;
      c.M[20 * row + k] = (quad_lin_slot(c.L[3 * i + 0], OSFM_E(0, j), k) + quad_lin_slot(c.L[3 * i + 1], OSFM_E(1, j), k)) +
                          quad_lin_slot(c.L[3 * i + 2], OSFM_E(2, j), k);
    }
  });
#undef OSFM_E
  // Gauss-Jordan on the cubic monomials, partial pivoting
  for (int k = 0; k < 10; k++) {
    int pr = k;
    for (int i = k + 1; i < 10; i++)
      if (fabs(c.M[i * 20 + k]) > fabs(c.M[pr * 20 + k])) pr = i;
    if (!(fabs(c.M[pr * 20 + k]) > 0)) return 0;
    w.parallel_for(20, [&](int j) {
      const double t = c.M[k * 20 + j];
      c.M[k * 20 + j] = c.M[pr * 20 + j];
      c.M[pr * 20 + j] = t;
    });
    const double ip = 1.0 / c.M[k * 20 + k];
    w.parallel_for(20, [&](int j) { c.M[k * 20 + j] *= ip; });
    w.parallel_for(10, [&](int i) { c.fcol[i] = c.M[i * 20 + k]; });
    w.parallel_for(200, [&](int t) {
      const int i = t / 20, j = t % 20;
      if (i == k) return;
      const double f = c.fcol[i];
      if (f == 0.0) return;
      c.M[i * 20 + j] -= f * c.M[k * 20 + j];
    });
  }
  w.parallel_for(100, [&](int t) {  // action matrix of "multiply by x"
    const int r = t / 10, j = t % 10;
    double v = 0.0;
    if (r < 6) {
      constexpr int src[6] = {0, 1, 2, 4, 5, 7};
      v = -c.M[src[r] * 20 + 10 + j];
    } else if ((r == 6 && j == 0) || (r == 7 && j == 1) || (r == 8 && j == 3) || (r == 9 && j == 6)) {
      v = 1.0;
    }
    c.At[t] = v;
    c.Aq[t] = v;
  });
  w.single([&]() { c.nreal = real_eigenvalues10(c.Aq, c.wr); });
  const int nreal = c.nreal;
  // eigenvectors: complete-pivot elimination of (At - lambda I), all eigenvalues at once
  w.parallel_for(100 * nreal, [&](int t) {
    const int e = t / 100, idx = t % 100;
    c.S[e][idx] = (idx / 10 == idx % 10) ? c.At[idx] - c.wr[e] : c.At[idx];
    if (idx < 10) c.colperm[e][idx] = idx;
    if (idx == 10) c.alive[e] = 1;
  });
  for (int k = 0; k < 9; k++) {
    w.parallel_for(nreal, [&](int e) {  // pivot search (first maximum in row-major order)
      if (!c.alive[e]) return;
      int pr = k, pc = k;
      double best = 0;
      for (int i = k; i < 10; i++)
        for (int j = k; j < 10; j++)
          if (fabs(c.S[e][i * 10 + j]) > best) {
            best = fabs(c.S[e][i * 10 + j]);
            pr = i;
            pc = j;
          }
      if (!(best > 0)) c.alive[e] = 0;
      c.pr[e] = pr;
      c.pc[e] = pc;
    });
    w.parallel_for(10 * nreal, [&](int t) {  // row swap
      const int e = t / 10, j = t % 10;
      if (!c.alive[e]) return;
      const double v = c.S[e][k * 10 + j];
      c.S[e][k * 10 + j] = c.S[e][c.pr[e] * 10 + j];
      c.S[e][c.pr[e] * 10 + j] = v;
    });
    w.parallel_for(10 * nreal, [&](int t) {  // column swap (+ the permutation record)
      const int e = t / 10, i = t % 10;
      if (!c.alive[e]) return;
      const double v = c.S[e][i * 10 + k];
      c.S[e][i * 10 + k] = c.S[e][i * 10 + c.pc[e]];
      c.S[e][i * 10 + c.pc[e]] = v;
      if (i == 0) {
        const int p = c.colperm[e][k];
        c.colperm[e][k] = c.colperm[e][c.pc[e]];
        c.colperm[e][c.pc[e]] = p;
      }
    });
    w.parallel_for(nreal, [&](int e) {
      if (c.alive[e]) c.ipv[e] = 1.0 / c.S[e][k * 10 + k];
    });
    w.parallel_for(10 * nreal, [&](int t) {
      const int e = t / 10, j = t % 10;
      if (c.alive[e]) c.S[e][k * 10 + j] *= c.ipv[e];
    });
    w.parallel_for(10 * nreal, [&](int t) {
      const int e = t / 10, i = t % 10;
      if (c.alive[e]) c.fS[e][i] = c.S[e][i * 10 + k];
    });
    w.parallel_for(100 * nreal, [&](int t) {
      const int e = t / 100, i = (t % 100) / 10, j = t % 10;
      if (!c.alive[e] || i == k) return;
      const double f = c.fS[e][i];
      if (f == 0.0) return;
      c.S[e][i * 10 + j] -= f * c.S[e][k * 10 + j];
    });
  }
  w.parallel_for(nreal, [&](int e) {  // monomial vector -> (x, y, z) -> E, normalised
    c.valid[e] = 0;
    if (!c.alive[e]) return;
    double v[10];
    for (int q = 0; q < 10; q++)  // v[colperm[q]] = ..., written as a gather so that v[] is indexed statically
      for (int k = 0; k < 10; k++)
        if (c.colperm[e][k] == q) v[q] = (k == 9) ? 1.0 : -c.S[e][k * 10 + 9];
    if (v[9] == 0.0) return;
    const double x = v[6] / v[9], y = v[7] / v[9], z = v[8] / v[9];
    double Em[9], nrm = 0.0;
    for (int i = 0; i < 9; i++) {
      Em[i] = x * c.basis[i * 4 + 0] + y * c.basis[i * 4 + 1] + z * c.basis[i * 4 + 2] + c.basis[i * 4 + 3];
      nrm += Em[i] * Em[i];
    }
    nrm = sqrt(nrm);
    if (!(nrm > 0) || !isfinite(nrm)) return;
    for (int i = 0; i < 9; i++) c.Ecand[e][i] = Em[i] / nrm;
    c.valid[e] = 1;
  });
  int count = 0;
  for (int e = 0; e < nreal; e++) count += c.valid[e];
  w.parallel_for(9 * nreal, [&](int t) {  // keep the valid ones, in eigenvalue order
    const int e = t / 9, i = t % 9;
    if (!c.valid[e]) return;
    int pos = 0;
    for (int q = 0; q < e; q++) pos += c.valid[q];
    c.Es[9 * pos + i] = c.Ecand[e][i];
  });
  return count;
}

// The reference's sequential LO-RANSAC (robust_estimator.h:37-119): same results as ransac_relative_pose_wave.
template <class W>
OSFM_HD void ransac_relative_pose_seq(W& w, WaveShared& s, CoopShared& c, const PairWork& P, const RansacParams& prm, RansacResult& out) {
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
  for (; it < prm.iterations && !stop; it++) {
    w.single([&]() { draw_sample_shared(s, 5, n, s.sidx[0]); });
    w.parallel_for(15, [&](int t) {
      c.s1[t] = P.b1[3 * s.sidx[0][t / 3] + t % 3];
      c.s2[t] = P.b2[3 * s.sidx[0][t / 3] + t % 3];
    });
    const int nm = essential_five_points_coop(w, c);
    w.parallel_for(nm, [&](int j) {
      double RT[12], x1[15], x2[15];
      for (int i = 0; i < 12; i++) RT[i] = 0.0;
      for (int i = 0; i < 15; i++) {
        x1[i] = c.s1[i];
        x2[i] = c.s2[i];
      }
      relative_pose_from_essential(c.Es + 9 * j, x1, x2, nullptr, 5, RT);
      for (int i = 0; i < 12; i++) c.RT[j][i] = RT[i];
    });
    for (int j = 0; j < nm && !stop; j++) {
      double RT[12];
      for (int i = 0; i < 12; i++) RT[i] = c.RT[j][i];
      const int cnt = w.count_if(n, [&](int i) { return fabs(relpose_error(RT, P.b1 + 3 * i, P.b2 + 3 * i)) < thr; });
      if (cnt >= best_score) {
        best_score = cnt;
        best_n = cnt;
        w.compact(n, [&](int i) { return fabs(relpose_error(RT, P.b1 + 3 * i, P.b2 + 3 * i)) < thr; }, P.inliers);
        for (int i = 0; i < 12; i++) out.model[i] = out.lo_model[i] = RT[i];
      }
      const int best_found = (cnt == best_score) && cnt >= 5;
      if (best_found && prm.use_lo) {
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
          if (c2 >= best_score) {
            best_score = c2;
            best_n = c2;
            w.compact(n, [&](int i) { return fabs(relpose_error(RTlo, P.b1 + 3 * i, P.b2 + 3 * i)) < thr; }, P.inliers);
            for (int i = 0; i < 12; i++) out.lo_model[i] = RTlo[i];
          }
        }
      }
      stop = P.stop_bound[best_n] < (double)it;
    }
  }
  out.best_score = best_score;
  out.iterations_run = it;
}

template <class W>
OSFM_HD void robust_match_calibrated_seq(W& w, WaveShared& s, CoopShared& c, const PairWork& P, const RansacParams& prm, int refine_iterations,
                                         MatchResult& out) {
  if (!robust_match_begin(P, out)) return;
  ransac_relative_pose_seq(w, s, c, P, prm, out.ransac);
  robust_match_finish_wave(w, s, P, prm, refine_iterations, out);
}

}  // namespace osfm_rp
