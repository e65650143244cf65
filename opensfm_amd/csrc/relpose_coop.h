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
  double JA[81], JV[81];     // N-point solve: A^T A and its Jacobi eigenvectors
  int ncolperm[9];           // column permutation of the 5 x 9 null-space elimination
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

// real_eigenvalues10 (relpose_core.h: elementary-similarity Hessenberg reduction + Francis double-shift QR) with every row / column
// sweep spread over the lanes.  The scalar control flow (pivot choice, deflation tests, shifts, the reflector of each step) is
// recomputed by every lane from the shared matrix -- it is uniform --, writes go through parallel steps only.  a: 10 x 10 in shared
// memory (destroyed); the real eigenvalues are left in c.wr in the order the sequential routine finds them.
template <class W>
OSFM_HD int real_eigenvalues10_coop(W& w, CoopShared& c, double* a) {
  constexpr int n = 10;
  for (int m = 1; m < n - 1; m++) {
    double x = 0.0;
    int i = m;
    for (int j = m; j < n; j++)
      if (fabs(a[j * n + m - 1]) > fabs(x)) {
        x = a[j * n + m - 1];
        i = j;
      }
    if (i != m) {
      w.parallel_for(n - (m - 1), [&](int t) {
        const int j = m - 1 + t;
        const double v = a[i * n + j];
        a[i * n + j] = a[m * n + j];
        a[m * n + j] = v;
      });
      w.parallel_for(n, [&](int j) {
        const double v = a[j * n + i];
        a[j * n + i] = a[j * n + m];
        a[j * n + m] = v;
      });
    }
    if (x != 0.0) {
      for (int i2 = m + 1; i2 < n; i2++) {
        double y = a[i2 * n + m - 1];
        if (y != 0.0) {
          y /= x;
          w.parallel_for(1 + n - m, [&](int t) {  // item 0 stores the multiplier, the others update row i2
            if (t == 0) {
              a[i2 * n + m - 1] = y;
              return;
            }
            const int j = m + t - 1;
            a[i2 * n + j] -= y * a[m * n + j];
          });
          w.parallel_for(n, [&](int j) { a[j * n + m] += y * a[j * n + i2]; });
        }
      }
    }
  }
  w.parallel_for(n * n, [&](int t) {
    const int i = t / n, j = t % n;
    if (i >= 2 && j < i - 1) a[t] = 0.0;
  });
  int nreal = 0, nn = n - 1, its;
  double anorm = 0.0, t = 0.0, p = 0, q = 0, r = 0, s, wv, x, y, z;
  for (int i = 0; i < n; i++)
    for (int j = (i > 0 ? i - 1 : 0); j < n; j++) anorm += fabs(a[i * n + j]);
  while (nn >= 0) {
    its = 0;
    int l;
    do {
      for (l = nn; l >= 1; l--) {
        s = fabs(a[(l - 1) * n + l - 1]) + fabs(a[l * n + l]);
        if (s == 0.0) s = anorm;
        if (fabs(a[l * n + l - 1]) + s == s) {
          w.single([&]() { a[l * n + l - 1] = 0.0; });
          break;
        }
      }
      x = a[nn * n + nn];
      if (l == nn) {
        const double root = x + t;
        w.single([&]() { c.wr[nreal] = root; });
        nreal++;
        nn--;
      } else {
        y = a[(nn - 1) * n + nn - 1];
        wv = a[nn * n + nn - 1] * a[(nn - 1) * n + nn];
        if (l == nn - 1) {
          p = 0.5 * (y - x);
          q = p * p + wv;
          z = sqrt(fabs(q));
          x += t;
          if (q >= 0.0) {
            z = p + (p >= 0.0 ? fabs(z) : -fabs(z));
            const double r0 = x + z, r1 = (z != 0.0) ? x - wv / z : x + z;
            w.single([&]() {
              c.wr[nreal] = r0;
              c.wr[nreal + 1] = r1;
            });
            nreal += 2;
          }
          nn -= 2;
        } else {
          if (its == 60) return nreal;
          if (its == 10 || its == 20) {
            t += x;
            w.parallel_for(nn + 1, [&](int i) { a[i * n + i] -= x; });
            s = fabs(a[nn * n + nn - 1]) + fabs(a[(nn - 1) * n + nn - 2]);
            y = x = 0.75 * s;
            wv = -0.4375 * s * s;
          }
          ++its;
          int m;
          for (m = nn - 2; m >= l; m--) {
            z = a[m * n + m];
            r = x - z;
            s = y - z;
            p = (r * s - wv) / a[(m + 1) * n + m] + a[m * n + m + 1];
            q = a[(m + 1) * n + m + 1] - z - r - s;
            r = a[(m + 2) * n + m + 1];
            s = fabs(p) + fabs(q) + fabs(r);
            p /= s;
            q /= s;
            r /= s;
            if (m == l) break;
            const double u = fabs(a[m * n + m - 1]) * (fabs(q) + fabs(r));
            const double v = fabs(p) * (fabs(a[(m - 1) * n + m - 1]) + fabs(z) + fabs(a[(m + 1) * n + m + 1]));
            if (u + v == v) break;
          }
          w.parallel_for(nn - (m + 2) + 1, [&](int t2) {
            const int i = m + 2 + t2;
            a[i * n + i - 2] = 0.0;
            if (i != m + 2) a[i * n + i - 3] = 0.0;
          });
          for (int k = m; k <= nn - 1; k++) {
            if (k != m) {
              p = a[k * n + k - 1];
              q = a[(k + 1) * n + k - 1];
              r = 0.0;
              if (k != nn - 1) r = a[(k + 2) * n + k - 1];
              if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) {
                p /= x;
                q /= x;
                r /= x;
              }
            }
            const double sg = sqrt(p * p + q * q + r * r);
            s = p >= 0.0 ? sg : -sg;
            if (s != 0.0) {
              // the sub-diagonal entry of the reflector column: negated (first step, when the block does not start at l) or -s x
              const int write_sub = (k == m) ? (l != m) : 1;
              const double sub = (k == m) ? -a[k * n + k - (l != m ? 1 : 0)] : -s * x;
              p += s;
              x = p / s;
              y = q / s;
              z = r / s;
              q /= p;
              r /= p;
              const int last = (k == nn - 1);
              w.parallel_for(1 + (nn - k + 1), [&](int t2) {  // item 0 stores the sub-diagonal entry, the others sweep rows k..k+2
                if (t2 == 0) {
                  if (write_sub) a[k * n + k - 1] = sub;
                  return;
                }
                const int j = k + t2 - 1;
                double pj = a[k * n + j] + q * a[(k + 1) * n + j];
                if (!last) {
                  pj += r * a[(k + 2) * n + j];
                  a[(k + 2) * n + j] -= pj * z;
                }
                a[(k + 1) * n + j] -= pj * y;
                a[k * n + j] -= pj * x;
              });
              const int mmin = nn < k + 3 ? nn : k + 3;
              w.parallel_for(mmin - l + 1, [&](int t2) {  // columns k..k+2
                const int i = l + t2;
                double pi = x * a[i * n + k] + y * a[i * n + k + 1];
                if (!last) {
                  pi += z * a[i * n + k + 2];
                  a[i * n + k + 2] -= pi * r;
                }
                a[i * n + k + 1] -= pi * q;
                a[i * n + k] -= pi;
              });
            }
          }
        }
      }
    } while (l < nn - 1);
  }
  return nreal;
}

// nullspace_5x9 (relpose_core.h) with the row / column operations spread over the lanes; the pivot search is the same ordered scan.
template <class W>
OSFM_HD int nullspace_5x9_coop(W& w, CoopShared& c) {
  constexpr int m = 5, n = 9;
  w.parallel_for(n, [&](int j) { c.ncolperm[j] = j; });
  for (int k = 0; k < m; k++) {
    int pr = k, pc = k;
    double best = 0;
    for (int i = k; i < m; i++)
      for (int j = k; j < n; j++)
        if (fabs(c.A[i * n + j]) > best) {
          best = fabs(c.A[i * n + j]);
          pr = i;
          pc = j;
        }
    if (!(best > 0)) return 0;
    w.parallel_for(n, [&](int j) {
      const double t = c.A[k * n + j];
      c.A[k * n + j] = c.A[pr * n + j];
      c.A[pr * n + j] = t;
    });
    w.parallel_for(m + 1, [&](int i) {
      if (i == m) {
        const int t = c.ncolperm[k];
        c.ncolperm[k] = c.ncolperm[pc];
        c.ncolperm[pc] = t;
        return;
      }
      const double t = c.A[i * n + k];
      c.A[i * n + k] = c.A[i * n + pc];
      c.A[i * n + pc] = t;
    });
    const double ip = 1.0 / c.A[k * n + k];
    w.parallel_for(n, [&](int j) { c.A[k * n + j] *= ip; });
    w.parallel_for(m, [&](int i) { c.fcol[i] = c.A[i * n + k]; });
    w.parallel_for(m * n, [&](int t) {
      const int i = t / n, j = t % n;
      if (i == k) return;
      const double f = c.fcol[i];
      if (f == 0.0) return;
      c.A[i * n + j] -= f * c.A[k * n + j];
    });
  }
  w.parallel_for(36, [&](int t) {  // basis[j][f]
    const int j = t / 4, f = t % 4;
    double v = 0.0;
    if (j == c.ncolperm[m + f]) v = 1.0;
    for (int k = 0; k < m; k++)
      if (c.ncolperm[k] == j) v = -c.A[k * n + m + f];
    c.basis[t] = v;
  });
  return 1;
}

// Five-point solver on the sample in c.s1 / c.s2; returns the number of essential matrices left in c.Es (unit Frobenius norm).
template <class W>
OSFM_HD int essential_five_points_coop(W& w, CoopShared& c) {
  w.parallel_for(45, [&](int t) {
    const int i = t / 9, r = (t % 9) / 3, col = t % 3;
    c.A[t] = c.s2[3 * i + r] * c.s1[3 * i + col];
  });
  if (!nullspace_5x9_coop(w, c)) return 0;
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
  const int nreal = real_eigenvalues10_coop(w, c, c.Aq);
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

// EssentialNPoints (relpose_core.h essential_n_points) with the 9 x 9 cyclic Jacobi spread over the lanes: the rotations stay in the
// reference order, each one updates its two columns, then its two rows and the eigenvector columns, one element per item.
// idx: `count` correspondence indices (shared memory); E (9) is returned in registers, identically on every lane.  Returns 0 or 1.
template <class W>
OSFM_HD int essential_n_points_coop(W& w, CoopShared& c, const double* b1, const double* b2, const int* idx, int count, double* E) {
  if (count < 9) return 0;
  constexpr int n = 9;
  w.parallel_for(81, [&](int t) {
    const int i = t / 9, j = t % 9;
    double acc = 0.0;
    for (int s = 0; s < count; s++) {
      const double *x1 = b1 + 3 * idx[s], *x2 = b2 + 3 * idx[s];
      acc += (x2[i / 3] * x1[i % 3]) * (x2[j / 3] * x1[j % 3]);
    }
    c.JA[t] = acc;
    c.JV[t] = (i == j) ? 1.0 : 0.0;
  });
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0.0;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) off += c.JA[p * n + q] * c.JA[p * n + q];
    if (!(off > 1e-300)) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = c.JA[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (c.JA[q * n + q] - c.JA[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        w.parallel_for(n, [&](int k) {  // columns p, q
          const double akp = c.JA[k * n + p], akq = c.JA[k * n + q];
          c.JA[k * n + p] = cs * akp - sn * akq;
          c.JA[k * n + q] = sn * akp + cs * akq;
        });
        w.parallel_for(2 * n, [&](int item) {  // rows p, q of A; columns p, q of V
          const int k = item % n;
          if (item < n) {
            const double apk = c.JA[p * n + k], aqk = c.JA[q * n + k];
            c.JA[p * n + k] = cs * apk - sn * aqk;
            c.JA[q * n + k] = sn * apk + cs * aqk;
          } else {
            const double vkp = c.JV[k * n + p], vkq = c.JV[k * n + q];
            c.JV[k * n + p] = cs * vkp - sn * vkq;
            c.JV[k * n + q] = sn * vkp + cs * vkq;
          }
        });
      }
  }
  double wv[9];
  for (int i = 0; i < n; i++) wv[i] = c.JA[i * n + i];
  int lo = 0, lo2 = -1;
  for (int i = 1; i < 9; i++)
    if (wv[i] < wv[lo]) lo = i;
  for (int i = 0; i < 9; i++)
    if (i != lo && (lo2 < 0 || wv[i] < wv[lo2])) lo2 = i;
  double w_lo = 0.0, w_lo2 = 0.0;  // wv[lo], wv[lo2] without dynamic register indexing
  for (int i = 0; i < 9; i++) {
    if (i == lo) w_lo = wv[i];
    if (i == lo2) w_lo2 = wv[i];
  }
  const double s_small = sqrt(fmax(w_lo, 0.0)), s_next = sqrt(fmax(w_lo2, 0.0));
  if (!(s_next / s_small > 4.0)) return 0;
  double Em[9];
  for (int i = 0; i < 9; i++) Em[i] = c.JV[9 * i + lo];
  double U[9], S[3], Vv[9];
  svd3(Em, U, S, Vv);
  const double d = 0.5 * (S[0] + S[1]);
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) E[3 * a + b] = d * (U[3 * a] * Vv[3 * b] + U[3 * a + 1] * Vv[3 * b + 1]);
  return 1;
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
          });
          double Elo[9], RTlo[12];
          if (!essential_n_points_coop(w, c, P.b1, P.b2, s.lidx, lo_size, Elo)) continue;
          for (int i = 0; i < 12; i++) RTlo[i] = 0.0;
          relative_pose_from_essential(Elo, P.b1, P.b2, s.lidx, lo_size, RTlo);  // every lane, same data: no broadcast needed
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
