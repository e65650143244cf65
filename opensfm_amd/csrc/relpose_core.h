// relpose_core.h -- per-thread numerics of calibrated (essential-matrix) robust matching, SURVEY.md 8a M-a9 / 8f-3.
//
// What the reference does on this path (opensfm/matching.py:871-903 robust_match_calibrated):
//   bearings -> pyrobust.ransac_relative_pose (LO-RANSAC, robust/robust_estimator.h:37-119, five-point minimal
//   solver geometry/essential.h:99-160, pose from E geometry/relative_pose.h:12-84, N-point LO model
//   geometry/essential.h:162-192) -> 3 x (compute_inliers_bearings, relative_pose_refinement
//   geometry/relative_pose.h:86-183) -> compute_inliers_bearings.
//
// Everything here is written for ONE GPU lane: plain doubles, fixed-size local arrays, + - * / sqrt only in the
// RANSAC part (so results do not depend on a math library), no recursion, every loop bounded.  The functions are
// host+device so that tests/test_relpose_core_host.py can compile this header with g++ and compare every stage
// bit for bit against the CPU oracle without a GPU; relpose.hip supplies the wavefront-level orchestration.
// Contraction is off (-ffp-contract=off in build.sh): an FMA would change the bits.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define OSFM_HD __host__ __device__ inline
#else
#define OSFM_HD inline
#endif

namespace osfm_rp {

constexpr double kEps = 2.220446049250313e-16;

// ---------------------------------------------------------------------------------------------------------------
// Polynomials of degree <= 3 in (x, y, z), monomial order of geometry/essential.h:30-61:
//   0..9  : xxx xxy xyy yyy xxz xyz yyz xzz yzz zzz      (cubic)
//   10..15: xx xy yy xz yz zz                             (quadratic)
//   16..19: x y z 1                                       (linear)
// A linear polynomial keeps slots 16..19 (4 numbers), a quadratic one 10..19 (10 numbers), a cubic one all 20.
// kMul[i - 10][j - 16] = slot of monomial_i * monomial_j for i in 10..19, j in 16..19.
// ---------------------------------------------------------------------------------------------------------------
OSFM_HD int mul_slot(int i, int j) {
  constexpr signed char kMul[10][4] = {{0, 1, 4, 10},   {1, 2, 5, 11},   {2, 3, 6, 12},   {4, 5, 7, 13},   {5, 6, 8, 14},
                                       {7, 8, 9, 15},   {10, 11, 13, 16}, {11, 12, 14, 17}, {13, 14, 15, 18}, {16, 17, 18, 19}};
  return kMul[i - 10][j - 16];
}
// q[10] (slots 10..19) = a[4] * b[4] (slots 16..19), accumulated in (i, j) order from zero
OSFM_HD void mul_lin_lin(const double* a, const double* b, double* q) {
  for (int k = 0; k < 10; k++) q[k] = 0.0;
  for (int i = 16; i < 20; i++)
    for (int j = 16; j < 20; j++) q[mul_slot(i, j) - 10] += a[i - 16] * b[j - 16];
}
// c[20] = q[10] (slots 10..19) * b[4] (slots 16..19)
OSFM_HD void mul_quad_lin(const double* q, const double* b, double* c) {
  for (int k = 0; k < 20; k++) c[k] = 0.0;
  for (int i = 10; i < 20; i++)
    for (int j = 16; j < 20; j++) c[mul_slot(i, j)] += q[i - 10] * b[j - 16];
}

// Null space of the 5 x 9 epipolar system by Gauss-Jordan with complete pivoting: basis[9][4].
OSFM_HD int nullspace_5x9(double* A, double* basis) {
  constexpr int m = 5, n = 9;
  int colperm[9];
  for (int j = 0; j < n; j++) colperm[j] = j;
  for (int k = 0; k < m; k++) {
    int pr = k, pc = k;
    double best = 0;
    for (int i = k; i < m; i++)
      for (int j = k; j < n; j++)
        if (fabs(A[i * n + j]) > best) {
          best = fabs(A[i * n + j]);
          pr = i;
          pc = j;
        }
    if (!(best > 0)) return 0;
    for (int j = 0; j < n; j++) {
      const double t = A[k * n + j];
      A[k * n + j] = A[pr * n + j];
      A[pr * n + j] = t;
    }
    for (int i = 0; i < m; i++) {
      const double t = A[i * n + k];
      A[i * n + k] = A[i * n + pc];
      A[i * n + pc] = t;
    }
    {
      const int t = colperm[k];
      colperm[k] = colperm[pc];
      colperm[pc] = t;
    }
    const double ip = 1.0 / A[k * n + k];
    for (int j = 0; j < n; j++) A[k * n + j] *= ip;
    for (int i = 0; i < m; i++) {
      if (i == k) continue;
      const double f = A[i * n + k];
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++) A[i * n + j] -= f * A[k * n + j];
    }
  }
  constexpr int nf = n - m;
  for (int f = 0; f < nf; f++) {
    for (int j = 0; j < n; j++) basis[j * nf + f] = 0.0;
    // colperm[] is indexed dynamically: written as a select chain so that it stays in registers
    for (int j = 0; j < n; j++)
      if (j == colperm[m + f]) basis[j * nf + f] = 1.0;
    for (int k = 0; k < m; k++) basis[colperm[k] * nf + f] = -A[k * n + m + f];
  }
  return 1;
}

// Real eigenvalues of a general 10 x 10 matrix (destroyed): elementary-similarity Hessenberg reduction and
// Francis double-shift QR on the real Schur form, with the classic 60-iteration cap (so the loop is bounded).
OSFM_HD int real_eigenvalues10(double* a, double* wr) {
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
      for (int j = m - 1; j < n; j++) {
        const double t = a[i * n + j];
        a[i * n + j] = a[m * n + j];
        a[m * n + j] = t;
      }
      for (int j = 0; j < n; j++) {
        const double t = a[j * n + i];
        a[j * n + i] = a[j * n + m];
        a[j * n + m] = t;
      }
    }
    if (x != 0.0) {
      for (i = m + 1; i < n; i++) {
        double y = a[i * n + m - 1];
        if (y != 0.0) {
          y /= x;
          a[i * n + m - 1] = y;
          for (int j = m; j < n; j++) a[i * n + j] -= y * a[m * n + j];
          for (int j = 0; j < n; j++) a[j * n + m] += y * a[j * n + i];
        }
      }
    }
  }
  for (int i = 2; i < n; i++)
    for (int j = 0; j < i - 1; j++) a[i * n + j] = 0.0;
  int nreal = 0, nn = n - 1, its;
  double anorm = 0.0, t = 0.0, p = 0, q = 0, r = 0, s, w, x, y, z;
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
          a[l * n + l - 1] = 0.0;
          break;
        }
      }
      x = a[nn * n + nn];
      if (l == nn) {
        wr[nreal++] = x + t;
        nn--;
      } else {
        y = a[(nn - 1) * n + nn - 1];
        w = a[nn * n + nn - 1] * a[(nn - 1) * n + nn];
        if (l == nn - 1) {
          p = 0.5 * (y - x);
          q = p * p + w;
          z = sqrt(fabs(q));
          x += t;
          if (q >= 0.0) {
            z = p + (p >= 0.0 ? fabs(z) : -fabs(z));
            wr[nreal] = wr[nreal + 1] = x + z;
            if (z != 0.0) wr[nreal + 1] = x - w / z;
            nreal += 2;
          }
          nn -= 2;
        } else {
          if (its == 60) return nreal;
          if (its == 10 || its == 20) {
            t += x;
            for (int i = 0; i <= nn; i++) a[i * n + i] -= x;
            s = fabs(a[nn * n + nn - 1]) + fabs(a[(nn - 1) * n + nn - 2]);
            y = x = 0.75 * s;
            w = -0.4375 * s * s;
          }
          ++its;
          int m;
          for (m = nn - 2; m >= l; m--) {
            z = a[m * n + m];
            r = x - z;
            s = y - z;
            p = (r * s - w) / a[(m + 1) * n + m] + a[m * n + m + 1];
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
          for (int i = m + 2; i <= nn; i++) {
            a[i * n + i - 2] = 0.0;
            if (i != m + 2) a[i * n + i - 3] = 0.0;
          }
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
              if (k == m) {
                if (l != m) a[k * n + k - 1] = -a[k * n + k - 1];
              } else {
                a[k * n + k - 1] = -s * x;
              }
              p += s;
              x = p / s;
              y = q / s;
              z = r / s;
              q /= p;
              r /= p;
              for (int j = k; j <= nn; j++) {
                p = a[k * n + j] + q * a[(k + 1) * n + j];
                if (k != nn - 1) {
                  p += r * a[(k + 2) * n + j];
                  a[(k + 2) * n + j] -= p * z;
                }
                a[(k + 1) * n + j] -= p * y;
                a[k * n + j] -= p * x;
              }
              const int mmin = nn < k + 3 ? nn : k + 3;
              for (int i = l; i <= mmin; i++) {
                p = x * a[i * n + k] + y * a[i * n + k + 1];
                if (k != nn - 1) {
                  p += z * a[i * n + k + 2];
                  a[i * n + k + 2] -= p * r;
                }
                a[i * n + k + 1] -= p * q;
                a[i * n + k] -= p;
              }
            }
          }
        }
      }
    } while (l < nn - 1);
  }
  return nreal;
}

// Five-point solver.  b1, b2: 5 x 3 bearings (x2^T E x1 = 0); Es: up to 10 row-major 3 x 3 matrices of unit
// Frobenius norm; returns how many.  (geometry/essential.h:99-160, geometry/src/essential.cc:54-143)
OSFM_HD int essential_five_points(const double* b1, const double* b2, double* Es) {
  double A[5 * 9], basis[9 * 4];
  for (int i = 0; i < 5; i++)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) A[i * 9 + 3 * r + c] = b2[3 * i + r] * b1[3 * i + c];
  if (!nullspace_5x9(A, basis)) return 0;
  // E(x, y, z) = x E0 + y E1 + z E2 + E3: entry (i, j) is the linear polynomial basis[(3 i + j) * 4 + 0..3]
#define OSFM_E(i, j) (basis + (3 * (i) + (j)) * 4)
  double M[10 * 20];
  {  // det E = 0
    double qa[10], qb[10], c[20];
    double* d = M;
    mul_lin_lin(OSFM_E(0, 1), OSFM_E(1, 2), qa);
    mul_lin_lin(OSFM_E(0, 2), OSFM_E(1, 1), qb);
    for (int k = 0; k < 10; k++) qa[k] -= qb[k];
    mul_quad_lin(qa, OSFM_E(2, 0), d);
    mul_lin_lin(OSFM_E(0, 2), OSFM_E(1, 0), qa);
    mul_lin_lin(OSFM_E(0, 0), OSFM_E(1, 2), qb);
    for (int k = 0; k < 10; k++) qa[k] -= qb[k];
    mul_quad_lin(qa, OSFM_E(2, 1), c);
    for (int k = 0; k < 20; k++) d[k] += c[k];
    mul_lin_lin(OSFM_E(0, 0), OSFM_E(1, 1), qa);
    mul_lin_lin(OSFM_E(0, 1), OSFM_E(1, 0), qb);
    for (int k = 0; k < 10; k++) qa[k] -= qb[k];
    mul_quad_lin(qa, OSFM_E(2, 2), c);
    for (int k = 0; k < 20; k++) d[k] += c[k];
  }
  {  // (E E^T - 1/2 tr(E E^T) I) E = 0
    double L[9][10];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double q1[10], q2[10];
        mul_lin_lin(OSFM_E(i, 0), OSFM_E(j, 0), L[3 * i + j]);
        mul_lin_lin(OSFM_E(i, 1), OSFM_E(j, 1), q1);
        mul_lin_lin(OSFM_E(i, 2), OSFM_E(j, 2), q2);
        for (int k = 0; k < 10; k++) L[3 * i + j][k] = (L[3 * i + j][k] + q1[k]) + q2[k];
      }
    double tr[10];
    for (int k = 0; k < 10; k++) tr[k] = ((L[0][k] + L[4][k]) + L[8][k]) * 0.5;
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 10; k++) L[4 * i][k] -= tr[k];
    int row = 1;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double c1[20], c2[20];
        double* le = M + 20 * row++;
        mul_quad_lin(L[3 * i + 0], OSFM_E(0, j), le);
        mul_quad_lin(L[3 * i + 1], OSFM_E(1, j), c1);
        mul_quad_lin(L[3 * i + 2], OSFM_E(2, j), c2);
        for (int k = 0; k < 20; k++) le[k] = (le[k] + c1[k]) + c2[k];
      }
  }
#undef OSFM_E
  // Gauss-Jordan on the cubic monomials (columns 0..9), partial pivoting
  for (int k = 0; k < 10; k++) {
    int pr = k;
    for (int i = k + 1; i < 10; i++)
      if (fabs(M[i * 20 + k]) > fabs(M[pr * 20 + k])) pr = i;
    if (!(fabs(M[pr * 20 + k]) > 0)) return 0;
    for (int j = 0; j < 20; j++) {
      const double t = M[k * 20 + j];
      M[k * 20 + j] = M[pr * 20 + j];
      M[pr * 20 + j] = t;
    }
    const double ip = 1.0 / M[k * 20 + k];
    for (int j = 0; j < 20; j++) M[k * 20 + j] *= ip;
    for (int i = 0; i < 10; i++) {
      if (i == k) continue;
      const double f = M[i * 20 + k];
      if (f == 0.0) continue;
      for (int j = 0; j < 20; j++) M[i * 20 + j] -= f * M[k * 20 + j];
    }
  }
  // action matrix of "multiply by x" on [xx xy yy xz yz zz x y z 1]
  double At[100], Aq[100];
  for (int i = 0; i < 100; i++) At[i] = 0.0;
  {
    constexpr int src[6] = {0, 1, 2, 4, 5, 7};
    for (int r = 0; r < 6; r++)
      for (int j = 0; j < 10; j++) At[r * 10 + j] = -M[src[r] * 20 + 10 + j];
  }
  At[6 * 10 + 0] = 1.0;
  At[7 * 10 + 1] = 1.0;
  At[8 * 10 + 3] = 1.0;
  At[9 * 10 + 6] = 1.0;
  for (int i = 0; i < 100; i++) Aq[i] = At[i];
  double wr[10];
  const int nreal = real_eigenvalues10(Aq, wr);
  int count = 0;
  double* S = Aq;  // the Hessenberg copy is dead: reuse its storage for (At - lambda I)
  for (int e = 0; e < nreal && count < 10; e++) {
    double v[10];
    for (int i = 0; i < 100; i++) S[i] = At[i];
    for (int i = 0; i < 10; i++) S[i * 10 + i] -= wr[e];
    int colperm[10];
    for (int j = 0; j < 10; j++) colperm[j] = j;
    int ok = 1;
    for (int k = 0; k < 9 && ok; k++) {
      int pr = k, pc = k;
      double best = 0;
      for (int i = k; i < 10; i++)
        for (int j = k; j < 10; j++)
          if (fabs(S[i * 10 + j]) > best) {
            best = fabs(S[i * 10 + j]);
            pr = i;
            pc = j;
          }
      if (!(best > 0)) {
        ok = 0;
        break;
      }
      for (int j = 0; j < 10; j++) {
        const double t = S[k * 10 + j];
        S[k * 10 + j] = S[pr * 10 + j];
        S[pr * 10 + j] = t;
      }
      for (int i = 0; i < 10; i++) {
        const double t = S[i * 10 + k];
        S[i * 10 + k] = S[i * 10 + pc];
        S[i * 10 + pc] = t;
      }
      {
        const int t = colperm[k];
        colperm[k] = colperm[pc];
        colperm[pc] = t;
      }
      const double ip = 1.0 / S[k * 10 + k];
      for (int j = 0; j < 10; j++) S[k * 10 + j] *= ip;
      for (int i = 0; i < 10; i++) {
        if (i == k) continue;
        const double f = S[i * 10 + k];
        if (f == 0.0) continue;
        for (int j = 0; j < 10; j++) S[i * 10 + j] -= f * S[k * 10 + j];
      }
    }
    if (!ok) continue;
    v[colperm[9]] = 1.0;
    for (int k = 0; k < 9; k++) v[colperm[k]] = -S[k * 10 + 9];
    if (v[9] == 0.0) continue;
    const double x = v[6] / v[9], y = v[7] / v[9], z = v[8] / v[9];
    double Em[9], nrm = 0.0;
    for (int i = 0; i < 9; i++) {
      Em[i] = x * basis[i * 4 + 0] + y * basis[i * 4 + 1] + z * basis[i * 4 + 2] + basis[i * 4 + 3];
      nrm += Em[i] * Em[i];
    }
    nrm = sqrt(nrm);
    if (!(nrm > 0) || !isfinite(nrm)) continue;
    for (int i = 0; i < 9; i++) Es[9 * count + i] = Em[i] / nrm;
    count++;
  }
  return count;
}

// ---------------------------------------------------------------------------------------------------------------
// Relative pose from an essential matrix (geometry/relative_pose.h:12-84), one-sided Jacobi SVD of a 3 x 3.
// ---------------------------------------------------------------------------------------------------------------
OSFM_HD void svd3(const double* A, double* U, double* S, double* V) {
  double G[9];
  for (int i = 0; i < 9; i++) G[i] = A[i];
  for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 3; k++) {
          alpha += G[3 * k + p] * G[3 * k + p];
          beta += G[3 * k + q] * G[3 * k + q];
          gamma += G[3 * k + p] * G[3 * k + q];
        }
        if (gamma == 0.0) continue;
        off = fmax(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; k++) {
          const double gp = G[3 * k + p], gq = G[3 * k + q];
          G[3 * k + p] = c * gp - s * gq;
          G[3 * k + q] = s * gp + c * gq;
          const double vp = V[3 * k + p], vq = V[3 * k + q];
          V[3 * k + p] = c * vp - s * vq;
          V[3 * k + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-16) break;
  }
  int order[3] = {0, 1, 2};
  double nrm[3];
  for (int j = 0; j < 3; j++) nrm[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
  for (int a = 0; a < 2; a++)
    for (int b = a + 1; b < 3; b++)
      if (nrm[order[b]] > nrm[order[a]]) {
        const int t = order[a];
        order[a] = order[b];
        order[b] = t;
      }
  double Vs[9];
  for (int j = 0; j < 3; j++) {
    const int o = order[j];
    S[j] = nrm[o];
    for (int k = 0; k < 3; k++) {
      Vs[3 * k + j] = V[3 * k + o];
      U[3 * k + j] = nrm[o] > 0 ? G[3 * k + o] / nrm[o] : 0.0;
    }
  }
  for (int i = 0; i < 9; i++) V[i] = Vs[i];
  if (!(S[2] > 1e-12 * S[0])) {  // complete U to an orthonormal basis when the third singular value vanishes
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
}
OSFM_HD double det3(const double* M) {
  return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
// geometry::TriangulateTwoBearingsMidpointSolve (geometry/triangulation.h:84-108); first centre is the origin
OSFM_HD int triangulate_midpoint2(const double* c1, const double* r0, const double* r1, double* X) {
  const double t[3] = {c1[0] - 0.0, c1[1] - 0.0, c1[2] - 0.0};
  const double b0 = t[0] * r0[0] + t[1] * r0[1] + t[2] * r0[2], b1 = t[0] * r1[0] + t[1] * r1[1] + t[2] * r1[2];
  const double a00 = r0[0] * r0[0] + r0[1] * r0[1] + r0[2] * r0[2];
  const double a10 = r0[0] * r1[0] + r0[1] * r1[1] + r0[2] * r1[2];
  const double a01 = -a10, a11 = -(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  const double det = a00 * a11 - a01 * a10;
  if (-1e-10 < det && det < 1e-10) return 0;
  const double l0 = (a11 * b0 - a01 * b1) / det, l1 = (-a10 * b0 + a00 * b1) / det;
  for (int i = 0; i < 3; i++) X[i] = 0.5 * ((0.0 + l0 * r0[i]) + (c1[i] + l1 * r1[i]));
  return 1;
}
// E row-major; b1/b2: n bearing pairs addressed through idx (idx == nullptr: 0..n-1); RT: 3 x 4 row-major [R | t]
OSFM_HD int relative_pose_from_essential(const double* E, const double* b1, const double* b2, const int* idx, int n, double* RT) {
  double U[9], S[3], V[9], Vt[9];
  svd3(E, U, S, V);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Vt[3 * i + j] = V[3 * j + i];
  if (det3(U) < 0)
    for (int k = 0; k < 3; k++) U[3 * k + 2] = -U[3 * k + 2];
  if (det3(Vt) < 0)
    for (int k = 0; k < 3; k++) Vt[6 + k] = -Vt[6 + k];
  double best = 0.0;
  int found = 0;
  for (int i = 0; i < 2; i++) {
    double t[3] = {U[2], U[5], U[8]};
    if (i == 1)
      for (int k = 0; k < 3; k++) t[k] = -t[k];
    const double tn = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    for (int k = 0; k < 3; k++) t[k] /= tn;
    for (int j = 0; j < 2; j++) {
      // W = [0 -1 0; 1 0 0; 0 0 1] (j == 0) or its transpose
      double Wm[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
      if (j == 1) {
        Wm[1] = 1;
        Wm[3] = -1;
      }
      double UW[9], R[9];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) UW[3 * a + b] = U[3 * a] * Wm[b] + U[3 * a + 1] * Wm[3 + b] + U[3 * a + 2] * Wm[6 + b];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) R[3 * a + b] = UW[3 * a] * Vt[b] + UW[3 * a + 1] * Vt[3 + b] + UW[3 * a + 2] * Vt[6 + b];
      double c1[3];
      for (int a = 0; a < 3; a++) c1[a] = -(R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2]);
      double score = 0.0;
      for (int s = 0; s < n; s++) {
        const int m = idx ? idx[s] : s;
        const double *x = b1 + 3 * m, *y = b2 + 3 * m;
        double ry[3], X[3];
        for (int a = 0; a < 3; a++) ry[a] = R[a] * y[0] + R[3 + a] * y[1] + R[6 + a] * y[2];
        if (!triangulate_midpoint2(c1, x, ry, X)) continue;
        const double nx = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
        double Y[3];
        for (int a = 0; a < 3; a++) Y[a] = R[3 * a] * X[0] + R[3 * a + 1] * X[1] + R[3 * a + 2] * X[2] + t[a];
        const double ny = sqrt(Y[0] * Y[0] + Y[1] * Y[1] + Y[2] * Y[2]);
        score += 0.5 * ((X[0] * x[0] + X[1] * x[1] + X[2] * x[2]) / nx + (Y[0] * y[0] + Y[1] * y[1] + Y[2] * y[2]) / ny);
      }
      if (score > best) {
        best = score;
        found = 1;
        for (int a = 0; a < 3; a++) {
          for (int b = 0; b < 3; b++) RT[4 * a + b] = R[3 * a + b];
          RT[4 * a + 3] = t[a];
        }
      }
    }
  }
  return found;
}

// ---------------------------------------------------------------------------------------------------------------
// Sampler: std::mt19937(42) + libstdc++'s uniform_int_distribution (robust/random_sampler.h); the state lives wherever the caller
// puts it (LDS in the kernel).
// ---------------------------------------------------------------------------------------------------------------
struct Mt19937 {
  uint32_t* mt;  // 624 words
  int idx;
};
OSFM_HD void mt_seed(Mt19937& g, uint32_t seed) {
  g.mt[0] = seed;
  for (int i = 1; i < 624; i++) g.mt[i] = 1812433253u * (g.mt[i - 1] ^ (g.mt[i - 1] >> 30)) + (uint32_t)i;
  g.idx = 624;
}
OSFM_HD uint32_t mt_next(Mt19937& g) {
  if (g.idx >= 624) {
    for (int i = 0; i < 624; i++) {
      const uint32_t y = (g.mt[i] & 0x80000000u) | (g.mt[(i + 1) % 624] & 0x7fffffffu);
      g.mt[i] = g.mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g.idx = 0;
  }
  uint32_t y = g.mt[g.idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
// std::uniform_int_distribution<unsigned long>(0, range_max) on a 32-bit generator as libstdc++ >= 11 compiles it: Lemire's
// multiply-shift with rejection (the oracle pins this against the reference's random_sampler.h compiled on the build box)
OSFM_HD uint32_t mt_uniform(Mt19937& g, uint32_t range_max) {  // [0, range_max]
  const uint32_t range = range_max + 1u;
  uint64_t product = (uint64_t)mt_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)mt_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}
OSFM_HD void draw_sample(Mt19937& g, int size, int n, int* idx) {  // distinct indices (n >= size is the caller's duty)
  for (int i = 0; i < size; i++) {
    int dup;
    do {
      idx[i] = (int)mt_uniform(g, (uint32_t)(n - 1));
      dup = 0;
      for (int j = 0; j < i; j++) dup |= idx[j] == idx[i];
    } while (dup);
  }
}

// Cyclic Jacobi eigen-decomposition of a symmetric 9 x 9 (destroyed): eigenvalues w, eigenvectors V (columns).
OSFM_HD void jacobi_eig9(double* A, double* w, double* V) {
  constexpr int n = 9;
  for (int i = 0; i < n * n; i++) V[i] = (i % (n + 1) == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0.0;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) off += A[p * n + q] * A[p * n + q];
    if (!(off > 1e-300)) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}
// EssentialNPoints (geometry/essential.h:162-192) with foundation::SolveAX0 (foundation/numeric.h:20-43): 0 or 1 model
OSFM_HD int essential_n_points(const double* b1, const double* b2, const int* idx, int count, double* E) {
  if (count < 9) return 0;
  double AtA[81], w[9], V[81];
  for (int i = 0; i < 81; i++) AtA[i] = 0.0;
  for (int s = 0; s < count; s++) {
    const double *x1 = b1 + 3 * idx[s], *x2 = b2 + 3 * idx[s];
    double row[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) row[3 * r + c] = x2[r] * x1[c];
    for (int i = 0; i < 9; i++)
      for (int j = 0; j < 9; j++) AtA[9 * i + j] += row[i] * row[j];
  }
  jacobi_eig9(AtA, w, V);
  int lo = 0, lo2 = -1;
  for (int i = 1; i < 9; i++)
    if (w[i] < w[lo]) lo = i;
  for (int i = 0; i < 9; i++)
    if (i != lo && (lo2 < 0 || w[i] < w[lo2])) lo2 = i;
  const double s_small = sqrt(fmax(w[lo], 0.0)), s_next = sqrt(fmax(w[lo2], 0.0));
  if (!(s_next / s_small > 4.0)) return 0;
  double Em[9];
  for (int i = 0; i < 9; i++) Em[i] = V[9 * i + lo];
  double U[9], S[3], Vv[9];
  svd3(Em, U, S, Vv);
  const double d = 0.5 * (S[0] + S[1]);
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) E[3 * a + b] = d * (U[3 * a] * Vv[3 * b] + U[3 * a + 1] * Vv[3 * b + 1]);
  return 1;
}
// RelativePose::Evaluate (robust/relative_pose_model.h): 1 - mean cosine between the bearings and the midpoint
OSFM_HD double relpose_error(const double* RT, const double* x0, const double* y0) {
  double x[3], y[3], R[9], t[3];
  const double nx = sqrt(x0[0] * x0[0] + x0[1] * x0[1] + x0[2] * x0[2]), ny = sqrt(y0[0] * y0[0] + y0[1] * y0[1] + y0[2] * y0[2]);
  for (int a = 0; a < 3; a++) {
    x[a] = x0[a] / nx;
    y[a] = y0[a] / ny;
    t[a] = RT[4 * a + 3];
    for (int b = 0; b < 3; b++) R[3 * a + b] = RT[4 * a + b];
  }
  double c1[3], ry[3], X[3], Y[3];
  for (int a = 0; a < 3; a++) {
    c1[a] = -(R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2]);
    ry[a] = R[a] * y[0] + R[3 + a] * y[1] + R[6 + a] * y[2];
  }
  if (!triangulate_midpoint2(c1, x, ry, X)) return 1.0;
  for (int a = 0; a < 3; a++) Y[a] = R[3 * a] * X[0] + R[3 * a + 1] * X[1] + R[3 * a + 2] * X[2] + t[a];
  const double nX = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]), nY = sqrt(Y[0] * Y[0] + Y[1] * Y[1] + Y[2] * Y[2]);
  return 1.0 - 0.5 * ((X[0] * x[0] + X[1] * x[1] + X[2] * x[2]) / nX + (Y[0] * y[0] + Y[1] * y[1] + Y[2] * y[2]) / nY);
}
// ShouldStop (robust/robust_estimator.h:20-35) for MINIMAL_SAMPLES = 5: the bound depends only on (best inlier count, n,
// probability) and goes through std::pow / std::log, so the HOST tabulates it with its libm for every possible count (n + 1
// doubles per pair) and the kernel only looks it up -- the device math library cannot move a stopping decision.
inline double max_iterations_for(int best_n, int n, double probability) {  // host only
  const double ratio = (double)best_n / (double)n;
  double p1 = 1.0 - pow(ratio, 5.0);
  if (p1 > 1.0 - kEps) p1 = 1.0 - kEps;
  return log(1.0 - probability) / log(p1);
}

// ---------------------------------------------------------------------------------------------------------------
// Bearings and the inlier test of robust_match_calibrated (camera_instances.h:154-160, matching.py:805-844)
// ---------------------------------------------------------------------------------------------------------------
// Camera models (OSFM_CAMERA_* of include/osfm_mi355.h) as ProjectGeneric<PROJ, DISTO, AFF> (camera_instances.h:127-160):
//   proj  0 perspective, 1 fisheye, 2 dual (leading parameter "transition"), 3 spherical
//   kind  0 Disto2, 1 Disto24, 2 Disto2468, 3 DistoBrown, 4 Disto62, 5 Disto624, -1 none
//   na    1 UniformScale [focal], 4 Affine [focal aspect_ratio cx cy], 0 none
// parameters in the native order [PROJ][DISTO][AFF]; models 0 / 1 are [k1 k2 focal].
OSFM_HD void camera_layout(int model, int* proj, int* kind, int* nd, int* na) {
  switch (model) {
    case 0: *proj = 0; *kind = 1; *nd = 2; *na = 1; break;
    case 1: *proj = 1; *kind = 1; *nd = 2; *na = 1; break;
    case 2: *proj = 0; *kind = 3; *nd = 5; *na = 4; break;
    case 3: *proj = 1; *kind = 2; *nd = 4; *na = 4; break;
    case 4: *proj = 1; *kind = 4; *nd = 8; *na = 4; break;
    case 5: *proj = 1; *kind = 5; *nd = 12; *na = 4; break;
    case 6: *proj = 2; *kind = 1; *nd = 2; *na = 1; break;
    case 7: *proj = 0; *kind = 1; *nd = 2; *na = 4; break;
    case 8: *proj = 0; *kind = 0; *nd = 1; *na = 4; break;
    default: *proj = 3; *kind = -1; *nd = 0; *na = 0; break;  // 9: spherical
  }
}
// radial polynomial of the 1-D distortions and the "derivative" the reference's Newton iteration divides by
// (camera_distortions_functions.h:93-101, 191-199, 307-321: Disto2 / Disto24 use 1 + 2 k1 r2 (+ 4 k2 r4), not d(r D)/dr)
OSFM_HD double radial_1d(int kind, const double* k, double r2) {
  if (kind == 0) return 1.0 + r2 * k[0];
  if (kind == 1) return 1.0 + r2 * (k[0] + k[1] * r2);
  return 1.0 + r2 * (k[0] + r2 * (k[1] + r2 * (k[2] + r2 * k[3])));
}
OSFM_HD double radial_1d_derivative(int kind, const double* k, double r2) {
  if (kind == 0) return 1.0 + r2 * 2.0 * k[0];
  if (kind == 1) return 1.0 + r2 * 2.0 * (k[0] + 2.0 * k[1] * r2);
  return 1.0 + r2 * (3.0 * k[0] + r2 * (5.0 * k[1] + r2 * (7.0 * k[2] + r2 * 9.0 * k[3])));
}
// forward value and the four derivative entries [dX/dx, dX/dy, dY/dx, dY/dy] of the 2-D distortions, as
// ForwardDerivatives<T, false> writes them (camera_distortions_functions.h:348-385 Disto62, :527-566 Disto624, :726-753 Brown)
OSFM_HD void distort_2d(int kind, const double* k, double x, double y, double* out, double* jac) {
  const double x2 = x * x, y2 = y * y, r2 = x2 + y2;
  if (kind == 3) {
    const double k1 = k[0], k2 = k[1], k3 = k[2], p1 = k[3], p2 = k[4];
    const double x4 = x2 * x2, y4 = y2 * y2, r4 = r2 * r2, r6 = r4 * r2;
    const double rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
    out[0] = x * rad + (2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x));
    out[1] = y * rad + (2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y));
    jac[0] = 5.0 * k2 * x4 + 3.0 * k1 * x2 + 6.0 * k3 * x2 * r4 + 6.0 * k2 * x2 * y2 + k3 * r6 + k2 * y4 + k1 * y2 + 1.0 + 2.0 * p1 * y +
             6.0 * p2 * x;
    jac[1] = x * (2.0 * k1 * y + 4.0 * k2 * y * r2 + 6.0 * k3 * y * r4) + 2.0 * p1 * x + 2.0 * p2 * y;
    jac[3] = 5.0 * k2 * y4 + 3.0 * k1 * y2 + 6.0 * k3 * y2 * r4 + 6.0 * k2 * x2 * y2 + k3 * r6 + k2 * x4 + k1 * x2 + 1.0 + 2.0 * p2 * x +
             6.0 * p1 * y;
    jac[2] = y * (2.0 * k1 * x + 4.0 * k2 * x * r2 + 6.0 * k3 * x * r4) + 2.0 * p2 * y + 2.0 * p1 * x;
    return;
  }
  const double k1 = k[0], k2 = k[1], k3 = k[2], k4 = k[3], k5 = k[4], k6 = k[5], p1 = k[6], p2 = k[7];
  const double r2_2 = r2 * r2, r2_3 = r2_2 * r2, r2_4 = r2_3 * r2, r2_5 = r2_4 * r2;
  const double rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * (k4 + r2 * (k5 + r2 * k6)))));
  double tx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x), ty = 2.0 * p2 * x * y + p1 * (r2 + 2.0 * y * y);
  const double dx_dxt = 2.0 * y * p1 + 6.0 * p2 * x, dx_dyt = 2.0 * x * p1 + 2.0 * p2 * y, dy_dxt = dx_dyt,
               dy_dyt = 2.0 * x * p2 + 6.0 * p1 * y;
  const double dr_dx = 2.0 * x, dr_dy = 2.0 * y;
  const double dp_dr = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r2_2 + 4.0 * k4 * r2_3 + 5.0 * k5 * r2_4 + 6.0 * k6 * r2_5;
  if (kind == 5) {
    const double s0 = k[8], s1 = k[9], s2 = k[10], s3 = k[11];
    out[0] = x * rad + tx + (s0 * r2 + s1 * r2 * r2);
    out[1] = y * rad + ty + (s2 * r2 + s3 * r2 * r2);
    const double dx_dx_tp = s0 * 2.0 * x + s1 * 4.0 * x * r2, dx_dy_tp = s0 * 2.0 * y + s1 * 4.0 * y * r2;
    const double dy_dx_tp = s2 * 2.0 * x + s3 * 4.0 * x * r2, dy_dy_tp = s2 * 2.0 * y + s3 * 4.0 * y * r2;
    jac[0] = rad + x * dp_dr * dr_dx + dx_dxt + dx_dx_tp;
    jac[1] = x * dp_dr * dr_dy + dx_dyt + dx_dy_tp;
    jac[2] = y * dp_dr * dr_dx + dy_dxt + dy_dx_tp;
    jac[3] = rad + y * dp_dr * dr_dy + dy_dyt + dy_dy_tp;
  } else {
    out[0] = x * rad + tx;
    out[1] = y * rad + ty;
    jac[0] = rad + x * dp_dr * dr_dx + dx_dxt;
    jac[1] = x * dp_dr * dr_dy + dx_dyt;
    jac[2] = y * dp_dr * dr_dx + dy_dxt;
    jac[3] = rad + y * dp_dr * dr_dy + dy_dyt;
  }
}
// DISTO::Backward: Newton-Raphson, 10 iterations, stop when the decrement is below 1e-6 BEFORE applying it
// (foundation/newton_raphson.h:76-90).  1-D kinds iterate on the radius; 2-D kinds on the point with
// decr = ((M^T M)^-1 M^T) f where M is the reference's Mat2 filled through .data(), i.e. the TRANSPOSE of the Jacobian
// (column-major storage; it only matters for the thin-prism terms of Disto624, whose Jacobian is not symmetric).
OSFM_HD void undistort(int kind, const double* k, double xd, double yd, double* xu, double* yu) {
  *xu = xd;
  *yu = yd;
  if (kind < 0) return;
  const double rd = sqrt(xd * xd + yd * yd);
  if (rd < kEps) return;
  if (kind <= 2) {
    double r = rd;
    for (int it = 0; it < 10; it++) {
      const double r2 = r * r;
      const double fv = r * radial_1d(kind, k, r2) - rd;
      const double dv = radial_1d_derivative(kind, k, r2);
      const double decr = dv == 0.0 ? 0.0 : fv / dv;
      if (fabs(decr) < 1e-6) break;
      r -= decr;
    }
    const double dist = radial_1d(kind, k, r * r);
    *xu = xd / dist;
    *yu = yd / dist;
    return;
  }
  double cx = xd, cy = yd;
  for (int it = 0; it < 10; it++) {
    double o[2], j[4];
    distort_2d(kind, k, cx, cy, o, j);
    const double f0 = o[0] - xd, f1 = o[1] - yd;
    const double m00 = j[0], m10 = j[1], m01 = j[2], m11 = j[3];  // M(row, col), filled column by column
    const double a00 = m00 * m00 + m10 * m10, a01 = m00 * m01 + m10 * m11, a10 = m01 * m00 + m11 * m10, a11 = m01 * m01 + m11 * m11;
    const double invdet = 1.0 / (a00 * a11 - a10 * a01);
    const double i00 = a11 * invdet, i01 = -a01 * invdet, i10 = -a10 * invdet, i11 = a00 * invdet;
    // P = A^-1 M^T, decr = P f
    const double p00 = i00 * m00 + i01 * m01, p01 = i00 * m10 + i01 * m11, p10 = i10 * m00 + i11 * m01, p11 = i10 * m10 + i11 * m11;
    const double d0 = p00 * f0 + p01 * f1, d1 = p10 * f0 + p11 * f1;
    if (sqrt(d0 * d0 + d1 * d1) < 1e-6) break;
    cx -= d0;
    cy -= d1;
  }
  *xu = cx;
  *yu = cy;
}
// Camera::Bearing -> ProjectGeneric::Backward = PROJ::Backward(DISTO::Backward(AFF::Backward(pixel)))
OSFM_HD void pixel_bearing_generic(int model, const double* par, double px, double py, double* b) {
  int proj, kind, nd, na;
  camera_layout(model, &proj, &kind, &nd, &na);
  const double* kd = par + (proj == 2 ? 1 : 0);
  const double* ka = kd + nd;
  double xd = px, yd = py;
  if (na == 1) {  // UniformScale::Backward
    xd = px / ka[0];
    yd = py / ka[0];
  } else if (na == 4) {  // Affine::Backward
    xd = (px - ka[2]) / ka[0];
    yd = (py - ka[3]) / (ka[1] * ka[0]);
  }
  double xu, yu;
  undistort(kind, kd, xd, yd, &xu, &yu);
  if (proj == 1) {  // FisheyeProjection::Backward: the undistorted radius is the angle from the optical axis
    const double theta = sqrt(xu * xu + yu * yu);
    const double s = theta > 1e-8 ? sin(theta) / theta : 1.0;
    b[0] = xu * s;
    b[1] = yu * s;
    b[2] = cos(theta);
  } else if (proj == 3) {  // SphericalProjection::Backward
    const double lon = xu * 2 * M_PI, lat = -yu * 2 * M_PI;
    b[0] = cos(lat) * sin(lon);
    b[1] = -sin(lat);
    b[2] = cos(lat) * cos(lon);
  } else {
    if (proj == 2) {  // DualProjection::Backward: theta from r by 5 Newton steps (the first with a doubled derivative)
      const double t = par[0], r = sqrt(xu * xu + yu * yu);
      double theta = 0.0;
      for (int it = 0; it < 5; it++) {
        const double fv = t * tan(theta) + (1.0 - t) * theta - r;
        const double secant = 1.0 / cos(theta);
        const double dv = (it == 0 ? 2.0 : 1.0) * (t * secant * secant - t + 1);
        const double decr = dv == 0.0 ? 0.0 : fv / dv;
        if (fabs(decr) < 1e-6) break;
        theta -= decr;
      }
      const double s = tan(theta) / (t * tan(theta) + (1.0 - t) * theta);  // NaN at the exact image centre, as the reference
      xu = s * xu;
      yu = s * yu;
    }
    const double inv = 1.0 / sqrt(xu * xu + yu * yu + 1.0);
    b[0] = xu * inv;
    b[1] = yu * inv;
    b[2] = inv;
  }
}
OSFM_HD void pixel_bearing(int model, double k1, double k2, double f, double px, double py, double* b) {
  const double par[3] = {k1, k2, f};
  pixel_bearing_generic(model, par, px, py, b);
}
// R (row-major), t: second camera expressed in the first (matching.py:813-817)
OSFM_HD int inlier_bearing(const double* x, const double* y, const double* R, const double* t, double threshold) {
  double ry[3], X[3];
  for (int a = 0; a < 3; a++) ry[a] = R[3 * a] * y[0] + R[3 * a + 1] * y[1] + R[3 * a + 2] * y[2];
  if (!triangulate_midpoint2(t, x, ry, X)) return 0;
  const double n1 = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
  const double d[3] = {X[0] - t[0], X[1] - t[1], X[2] - t[2]};
  double q[3];
  for (int a = 0; a < 3; a++) q[a] = R[a] * d[0] + R[3 + a] * d[1] + R[6 + a] * d[2];
  const double n2 = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  double e1 = 0, e2 = 0;
  for (int a = 0; a < 3; a++) {
    const double u = X[a] / n1 - x[a], v = q[a] / n2 - y[a];
    e1 += u * u;
    e2 += v * v;
  }
  return (sqrt(e1) < threshold) && (sqrt(e2) < threshold);
}

// ---------------------------------------------------------------------------------------------------------------
// Refinement (geometry/relative_pose.h:86-183): glibc rand() picks, duals over the 6 parameters, TinySolver LM.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRefineResiduals = 100;  // MAX_ERRORS; one more residual ties the baseline length to 1

// The i-th value of rand() after srand(42) only needs a 34-word window of glibc's additive-feedback generator:
// fills picked[100] = int(float(rand()) / RAND_MAX * n), clamped to n - 1.
OSFM_HD void refinement_picks(int n, int* picked) {
  int32_t r[34];
  {  // srand(42): 31 LCG words, 3 wrapped, 310 discarded outputs
    r[0] = 42;
    for (int i = 1; i < 31; i++) {
      const int64_t hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
      int64_t word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      r[i] = (int32_t)word;
    }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
  }
  // ring of 34: value k lives in r[k % 34]; value k = value[k-31] + value[k-3]
  for (int k = 34; k < 344 + kRefineResiduals; k++) {
    const int32_t v = (int32_t)((uint32_t)r[(k - 31) % 34] + (uint32_t)r[(k - 3) % 34]);
    r[k % 34] = v;
    if (k >= 344) {
      const int out = (int)(((uint32_t)v) >> 1);
      int idx = (int)(((float)out / (float)2147483647) * (float)n);
      if (idx >= n) idx = n - 1;
      picked[k - 344] = idx;
    }
  }
}

struct D6 {
  double v, d[6];
};
OSFM_HD D6 dc(double c) {
  D6 r;
  r.v = c;
  for (int i = 0; i < 6; i++) r.d[i] = 0;
  return r;
}
OSFM_HD D6 dvar(double c, int k) {
  D6 r = dc(c);
  r.d[k] = 1.0;
  return r;
}
OSFM_HD D6 dadd(D6 a, const D6& b) {
  for (int i = 0; i < 6; i++) a.d[i] += b.d[i];
  a.v += b.v;
  return a;
}
OSFM_HD D6 dsub(D6 a, const D6& b) {
  for (int i = 0; i < 6; i++) a.d[i] -= b.d[i];
  a.v -= b.v;
  return a;
}
OSFM_HD D6 dneg(D6 a) {
  for (int i = 0; i < 6; i++) a.d[i] = -a.d[i];
  a.v = -a.v;
  return a;
}
OSFM_HD D6 dmul(const D6& a, const D6& b) {
  D6 r;
  r.v = a.v * b.v;
  for (int i = 0; i < 6; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
OSFM_HD D6 dmulc(D6 a, double c) {
  for (int i = 0; i < 6; i++) a.d[i] *= c;
  a.v *= c;
  return a;
}
OSFM_HD D6 ddiv(const D6& a, const D6& b) {
  D6 r;
  const double ib = 1.0 / b.v;
  r.v = a.v * ib;
  for (int i = 0; i < 6; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
OSFM_HD D6 dsqrt(const D6& a) {
  D6 r;
  r.v = sqrt(a.v);
  const double h = 0.5 / r.v;
  for (int i = 0; i < 6; i++) r.d[i] = a.d[i] * h;
  return r;
}
OSFM_HD D6 dsin(const D6& a) {
  D6 r;
  r.v = sin(a.v);
  const double c = cos(a.v);
  for (int i = 0; i < 6; i++) r.d[i] = a.d[i] * c;
  return r;
}
OSFM_HD D6 dcos(const D6& a) {
  D6 r;
  r.v = cos(a.v);
  const double s = -sin(a.v);
  for (int i = 0; i < 6; i++) r.d[i] = a.d[i] * s;
  return r;
}
OSFM_HD D6 ddot3(const D6* a, const D6* b) { return dadd(dadd(dmul(a[0], b[0]), dmul(a[1], b[1])), dmul(a[2], b[2])); }
// ceres::AngleAxisRotatePoint
OSFM_HD void aa_rotate(const D6* aa, const D6* pt, D6* out) {
  const D6 theta2 = ddot3(aa, aa);
  if (theta2.v > kEps) {
    const D6 theta = dsqrt(theta2), ct = dcos(theta), st = dsin(theta), ti = ddiv(dc(1.0), theta);
    const D6 w[3] = {dmul(aa[0], ti), dmul(aa[1], ti), dmul(aa[2], ti)};
    const D6 wxp[3] = {dsub(dmul(w[1], pt[2]), dmul(w[2], pt[1])), dsub(dmul(w[2], pt[0]), dmul(w[0], pt[2])),
                       dsub(dmul(w[0], pt[1]), dmul(w[1], pt[0]))};
    const D6 tmp = dmul(ddot3(w, pt), dsub(dc(1.0), ct));
    for (int i = 0; i < 3; i++) out[i] = dadd(dadd(dmul(pt[i], ct), dmul(wxp[i], st)), dmul(w[i], tmp));
  } else {
    const D6 wxp[3] = {dsub(dmul(aa[1], pt[2]), dmul(aa[2], pt[1])), dsub(dmul(aa[2], pt[0]), dmul(aa[0], pt[2])),
                       dsub(dmul(aa[0], pt[1]), dmul(aa[1], pt[0]))};
    for (int i = 0; i < 3; i++) out[i] = dadd(pt[i], wxp[i]);
  }
}
// One residual of RelativePoseCost: i < 100 -> bearing pair (xs, ys); i == 100 -> 1 - |c| (xs, ys unused).
// par = [angle-axis of R, centre c = -R^T t].  out[0] = residual, out[1..6] = its gradient.
OSFM_HD void refine_residual(int i, const double* par, const double* xs, const double* ys, double* out) {
  D6 rot[3], tr[3], rot_t[3];
  for (int k = 0; k < 3; k++) {
    rot[k] = dvar(par[k], k);
    tr[k] = dvar(par[3 + k], 3 + k);
    rot_t[k] = dneg(rot[k]);
  }
  D6 r;
  if (i < kRefineResiduals) {
    const D6 x[3] = {dc(xs[0]), dc(xs[1]), dc(xs[2])}, y[3] = {dc(ys[0]), dc(ys[1]), dc(ys[2])};
    D6 ry[3];
    aa_rotate(rot_t, y, ry);
    const D6 b0 = ddot3(tr, x), b1v = ddot3(tr, ry);
    const D6 a00 = ddot3(x, x), a10 = ddot3(x, ry), a01 = dneg(a10), a11 = dneg(ddot3(ry, ry));
    const D6 det = dsub(dmul(a00, a11), dmul(a01, a10));
    if (-1e-10 < det.v && det.v < 1e-10) {
      r = dc(1.0);
    } else {
      const D6 l0 = ddiv(dsub(dmul(a11, b0), dmul(a01, b1v)), det), l1 = ddiv(dsub(dmul(a00, b1v), dmul(a10, b0)), det);
      D6 pt[3], yc[3], py[3];
      for (int a = 0; a < 3; a++) pt[a] = dmulc(dadd(dmul(l0, x[a]), dadd(tr[a], dmul(l1, ry[a]))), 0.5);
      const D6 npt = dsqrt(ddot3(pt, pt));
      for (int a = 0; a < 3; a++) yc[a] = dsub(pt[a], tr[a]);
      aa_rotate(rot, yc, py);
      const D6 npy = dsqrt(ddot3(py, py));
      const D6 s = dadd(ddiv(ddot3(pt, x), npt), ddiv(ddot3(py, y), npy));
      r = dsub(dc(1.0), dmulc(s, 0.5));
    }
  } else {
    r = dsub(dc(1.0), dsqrt(ddot3(tr, tr)));
  }
  out[0] = r.v;
  for (int k = 0; k < 6; k++) out[1 + k] = r.d[k];
}
// ceres RotationMatrixToAngleAxis (through the quaternion) / AngleAxisToRotationMatrix, R row-major
OSFM_HD void rotmat_to_aa(const double* R, double* aa) {
  double q[4];
  const double trace = R[0] + R[4] + R[8];
  if (trace >= 0.0) {
    double t = sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t;
    q[2] = (R[2] - R[6]) * t;
    q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j + 1] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k + 1] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > 0.0) {
    const double s = sqrt(s2), c = q[0];
    const double two_theta = 2.0 * (c < 0.0 ? atan2(-s, -c) : atan2(s, c));
    const double k = two_theta / s;
    for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * k;
  } else {
    for (int a = 0; a < 3; a++) aa[a] = q[a + 1] * 2.0;
  }
}
OSFM_HD void aa_to_rotmat(const double* aa, double* R) {
  const double th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2 > kEps) {
    const double th = sqrt(th2), wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th, c = cos(th), s = sin(th);
    R[0] = c + wx * wx * (1 - c);
    R[1] = wx * wy * (1 - c) - wz * s;
    R[2] = wy * s + wx * wz * (1 - c);
    R[3] = wz * s + wx * wy * (1 - c);
    R[4] = c + wy * wy * (1 - c);
    R[5] = -wx * s + wy * wz * (1 - c);
    R[6] = -wy * s + wx * wz * (1 - c);
    R[7] = wx * s + wy * wz * (1 - c);
    R[8] = c + wz * wz * (1 - c);
  } else {
    R[0] = 1;
    R[1] = -aa[2];
    R[2] = aa[1];
    R[3] = aa[2];
    R[4] = 1;
    R[5] = -aa[0];
    R[6] = -aa[1];
    R[7] = aa[0];
    R[8] = 1;
  }
}
OSFM_HD int ldlt_solve6(const double* A, const double* b, double* x) {
  double L[36], D[6];
  for (int i = 0; i < 36; i++) L[i] = 0.0;
  for (int j = 0; j < 6; j++) {
    double d = A[7 * j];
    for (int k = 0; k < j; k++) d -= L[6 * j + k] * L[6 * j + k] * D[k];
    if (d == 0.0 || !isfinite(d)) return 0;
    D[j] = d;
    L[7 * j] = 1.0;
    for (int i = j + 1; i < 6; i++) {
      double v = A[6 * i + j];
      for (int k = 0; k < j; k++) v -= L[6 * i + k] * L[6 * j + k] * D[k];
      L[6 * i + j] = v / d;
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) {
    double v = b[i];
    for (int k = 0; k < i; k++) v -= L[6 * i + k] * y[k];
    y[i] = v;
  }
  for (int i = 0; i < 6; i++) y[i] /= D[i];
  for (int i = 5; i >= 0; i--) {
    double v = y[i];
    for (int k = i + 1; k < 6; k++) v -= L[6 * k + i] * x[k];
    x[i] = v;
  }
  return 1;
}

// The Levenberg-Marquardt driver of ceres::TinySolver, written against an evaluator so that the kernel can spread
// the 101 residuals over the lanes of a wavefront while a host test evaluates them in a loop:
//   eval(par, want_jacobian) must leave  res[101]  and (when asked)  jac[101][6]  where the driver can read them
//   through the two accessors; all reductions below run over i = 0..100 in order (the oracle's order).
// RT (3 x 4 row-major, x2 ~ R x1 + t) is refined in place; returns the TinySolver iteration count.
template <class Eval>
OSFM_HD int refine_relative_pose(double* RT, int iterations, Eval& ev, double* costs) {
  constexpr int NR = kRefineResiduals + 1;
  double R[9], x[6];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) R[3 * a + b] = RT[4 * a + b];
  rotmat_to_aa(R, x);
  for (int a = 0; a < 3; a++) x[3 + a] = -(R[a] * RT[3] + R[3 + a] * RT[7] + R[6 + a] * RT[11]);
  double scal[6], jtj[36], g[6], cost = 0;
  int have_scale = 0, it = 0;
  auto update = [&]() {
    ev.eval(x, 1);
    if (!have_scale) {
      for (int k = 0; k < 6; k++) {
        double s = 0;
        for (int i = 0; i < NR; i++) s += ev.jac(i, k) * ev.jac(i, k);
        scal[k] = 1.0 / (1.0 + sqrt(s));
      }
      have_scale = 1;
    }
    for (int a = 0; a < 6; a++) {
      g[a] = 0;
      for (int i = 0; i < NR; i++) g[a] += (ev.jac(i, a) * scal[a]) * (-ev.res(i));
      for (int b = 0; b < 6; b++) {
        double s = 0;
        for (int i = 0; i < NR; i++) s += (ev.jac(i, a) * scal[a]) * (ev.jac(i, b) * scal[b]);
        jtj[6 * a + b] = s;
      }
    }
    cost = 0;
    for (int i = 0; i < NR; i++) cost += (-ev.res(i)) * (-ev.res(i));
    cost *= 0.5;
  };
  update();
  if (costs) costs[0] = cost;
  double gmax = 0;
  for (int k = 0; k < 6; k++) gmax = fmax(gmax, fabs(g[k]));
  double u = 1.0 / 1e4, v = 2.0;
  if (!(gmax < 1e-10) && !(cost < kEps)) {
    for (it = 1; it < iterations; it++) {
      double reg[36], step[6], dx[6], xn[6];
      for (int k = 0; k < 36; k++) reg[k] = jtj[k];
      for (int k = 0; k < 6; k++) reg[7 * k] += u * fmin(fmax(jtj[7 * k], 1e-6), 1e32);
      if (!ldlt_solve6(reg, g, step)) {
        u *= v;
        v *= 2;
        continue;
      }
      double dxn = 0, xnorm = 0;
      for (int k = 0; k < 6; k++) {
        dx[k] = scal[k] * step[k];
        dxn += dx[k] * dx[k];
        xnorm += x[k] * x[k];
      }
      if (sqrt(dxn) < 1e-8 * (sqrt(xnorm) + 1e-8)) break;
      for (int k = 0; k < 6; k++) xn[k] = x[k] + dx[k];
      ev.eval(xn, 0);
      double fn2 = 0;
      for (int i = 0; i < NR; i++) fn2 += ev.res(i) * ev.res(i);
      const double cost_change = 2.0 * cost - fn2;
      double mc = 0;
      for (int a = 0; a < 6; a++) {
        double s = 2.0 * g[a];
        for (int b = 0; b < 6; b++) s -= jtj[6 * a + b] * step[b];
        mc += step[a] * s;
      }
      const double rho = cost_change / mc;
      if (rho > 0) {
        for (int k = 0; k < 6; k++) x[k] = xn[k];
        update();
        gmax = 0;
        for (int k = 0; k < 6; k++) gmax = fmax(gmax, fabs(g[k]));
        if (gmax < 1e-10 || cost < kEps) {
          it++;
          break;
        }
        const double tmp = 2.0 * rho - 1.0;
        u = u * fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp);
        v = 2.0;
        continue;
      }
      u *= v;
      v *= 2.0;
    }
  }
  if (costs) costs[1] = cost;
  aa_to_rotmat(x, R);
  for (int a = 0; a < 3; a++) {
    for (int b = 0; b < 3; b++) RT[4 * a + b] = R[3 * a + b];
    RT[4 * a + 3] = -(R[3 * a] * x[3] + R[3 * a + 1] * x[4] + R[3 * a + 2] * x[5]);
  }
  return it;
}

}  // namespace osfm_rp
