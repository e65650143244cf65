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
// Loops with a fixed trip count are unrolled on the device: a lane runs these routines alone on its SIMD (one wavefront per CU in
// the solver kernels), so the only latency hiding there is comes from issuing the LDS reads of a whole row before the arithmetic.
// Variable-length loops are written over their full range with a predicate for the same reason (same operations per element).
#define OSFM_UNROLL _Pragma("unroll")
#define OSFM_NOUNROLL _Pragma("unroll 1")  // long-bodied outer loops: unrolling them only multiplies the live registers
#else
#define OSFM_HD inline
#define OSFM_UNROLL
#define OSFM_NOUNROLL
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

// Per-lane arrays whose element i lives at p[i * STRIDE].  STRIDE = 1 is an ordinary array (host code, register / stack arrays);
// the lane-per-problem solver kernels keep their matrices in LDS with STRIDE = 64 (element-major, lane-minor: lane l owns
// p[l], p[64 + l], ...), so that ANY per-lane dynamic index hits the lane's own bank -- pivoting and deflation loops index their
// matrices differently in every lane and would otherwise live in scratch memory.  The numerics below take "anything indexable".
template <class T, int STRIDE>
struct LaneArr {
  T* p;
  OSFM_HD T& operator[](int i) const { return p[i * STRIDE]; }
  OSFM_HD LaneArr operator+(int k) const { return LaneArr{p + k * STRIDE}; }
};

// Null space of the 5 x 9 epipolar system by Gauss-Jordan with complete pivoting: basis[9][4].  colperm: 9 ints of work space.
template <class PA, class PB, class PI>
OSFM_HD int nullspace_5x9(PA A, PB basis, PI colperm) {
  constexpr int m = 5, n = 9;
  OSFM_UNROLL for (int j = 0; j < n; j++) colperm[j] = j;
  for (int k = 0; k < m; k++) {
    int pr = k, pc = k;
    double best = 0;
    OSFM_UNROLL for (int i = 0; i < m; i++)
      OSFM_UNROLL for (int j = 0; j < n; j++) {
        const double a = fabs(A[i * n + j]);
        if (i >= k && j >= k && a > best) {
          best = a;
          pr = i;
          pc = j;
        }
      }
    if (!(best > 0)) return 0;
    OSFM_UNROLL for (int j = 0; j < n; j++) {
      const double t = A[k * n + j];
      A[k * n + j] = A[pr * n + j];
      A[pr * n + j] = t;
    }
    OSFM_UNROLL for (int i = 0; i < m; i++) {
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
    double rowk[n];
    OSFM_UNROLL for (int j = 0; j < n; j++) {
      rowk[j] = A[k * n + j] * ip;
      A[k * n + j] = rowk[j];
    }
    for (int i = 0; i < m; i++) {
      if (i == k) continue;
      const double f = A[i * n + k];
      if (f == 0.0) continue;
      OSFM_UNROLL for (int j = 0; j < n; j++) A[i * n + j] -= f * rowk[j];
    }
  }
  constexpr int nf = n - m;
  for (int f = 0; f < nf; f++) {
    OSFM_UNROLL for (int j = 0; j < n; j++) basis[j * nf + f] = 0.0;
    // colperm[] is indexed dynamically: written as a select chain so that it stays in registers
    const int cf = colperm[m + f];
    OSFM_UNROLL for (int j = 0; j < n; j++)
      if (j == cf) basis[j * nf + f] = 1.0;
    OSFM_UNROLL for (int k = 0; k < m; k++) basis[colperm[k] * nf + f] = -A[k * n + m + f];
  }
  return 1;
}
OSFM_HD int nullspace_5x9(double* A, double* basis) {
  int colperm[9];
  return nullspace_5x9(A, basis, colperm);
}

// Real eigenvalues of a general 10 x 10 matrix (destroyed): elementary-similarity Hessenberg reduction and
// Francis double-shift QR on the real Schur form, with the classic 60-iteration cap (so the loop is bounded).
// put(k, value) receives the k-th real eigenvalue, k = 0, 1, ... in the order the iteration deflates them
template <class PA, class Put>
OSFM_HD int real_eigenvalues10_put(PA a, Put put) {
  constexpr int n = 10;
  for (int m = 1; m < n - 1; m++) {
    double x = 0.0;
    int i = m;
    {
      double col[n];
      OSFM_UNROLL for (int j = 0; j < n; j++) col[j] = a[j * n + m - 1];
      OSFM_UNROLL for (int j = 0; j < n; j++)
        if (j >= m && fabs(col[j]) > fabs(x)) {
          x = col[j];
          i = j;
        }
    }
    if (i != m) {
      {
        double ri[n], rm[n];
        OSFM_UNROLL for (int j = 0; j < n; j++) {
          ri[j] = a[i * n + j];
          rm[j] = a[m * n + j];
        }
        OSFM_UNROLL for (int j = 0; j < n; j++)
          if (j >= m - 1) {
            a[i * n + j] = rm[j];
            a[m * n + j] = ri[j];
          }
      }
      {
        double ci[n], cm[n];
        OSFM_UNROLL for (int j = 0; j < n; j++) {
          ci[j] = a[j * n + i];
          cm[j] = a[j * n + m];
        }
        OSFM_UNROLL for (int j = 0; j < n; j++) {
          a[j * n + i] = cm[j];
          a[j * n + m] = ci[j];
        }
      }
    }
    if (x != 0.0) {
      for (i = m + 1; i < n; i++) {
        double y = a[i * n + m - 1];
        if (y != 0.0) {
          y /= x;
          a[i * n + m - 1] = y;
          {
            double ri[n], rm[n];
            OSFM_UNROLL for (int j = 0; j < n; j++) {
              ri[j] = a[i * n + j];
              rm[j] = a[m * n + j];
            }
            OSFM_UNROLL for (int j = 0; j < n; j++)
              if (j >= m) a[i * n + j] = ri[j] - y * rm[j];
          }
          {
            double ci[n], cm[n];
            OSFM_UNROLL for (int j = 0; j < n; j++) {
              ci[j] = a[j * n + i];
              cm[j] = a[j * n + m];
            }
            OSFM_UNROLL for (int j = 0; j < n; j++) a[j * n + m] = cm[j] + y * ci[j];
          }
        }
      }
    }
  }
  OSFM_UNROLL for (int i = 2; i < n; i++)
    OSFM_UNROLL for (int j = 0; j < i - 1; j++) a[i * n + j] = 0.0;
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
        put(nreal, x + t);
        nreal++;
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
            put(nreal, x + z);
            put(nreal + 1, z != 0.0 ? x - w / z : x + z);
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
              {  // rows k, k + 1 (, k + 2), columns k .. nn
                const bool three = (k != nn - 1);
                double r0[n], r1[n], r2[n];
                OSFM_UNROLL for (int j = 0; j < n; j++) {
                  r0[j] = a[k * n + j];
                  r1[j] = a[(k + 1) * n + j];
                  r2[j] = three ? a[(k + 2) * n + j] : 0.0;
                }
                OSFM_UNROLL for (int j = 0; j < n; j++)
                  if (j >= k && j <= nn) {
                    double pj = r0[j] + q * r1[j];
                    if (three) {
                      pj += r * r2[j];
                      a[(k + 2) * n + j] = r2[j] - pj * z;
                    }
                    a[(k + 1) * n + j] = r1[j] - pj * y;
                    a[k * n + j] = r0[j] - pj * x;
                  }
              }
              {  // columns k, k + 1 (, k + 2), rows l .. min(nn, k + 3)
                const int mmin = nn < k + 3 ? nn : k + 3;
                const bool three = (k != nn - 1);
                double c0[n], c1[n], c2[n];
                OSFM_UNROLL for (int i = 0; i < n; i++) {
                  c0[i] = a[i * n + k];
                  c1[i] = a[i * n + k + 1];
                  c2[i] = three ? a[i * n + k + 2] : 0.0;
                }
                OSFM_UNROLL for (int i = 0; i < n; i++)
                  if (i >= l && i <= mmin) {
                    double pi = x * c0[i] + y * c1[i];
                    if (three) {
                      pi += z * c2[i];
                      a[i * n + k + 2] = c2[i] - pi * r;
                    }
                    a[i * n + k + 1] = c1[i] - pi * q;
                    a[i * n + k] = c0[i] - pi;
                  }
              }
            }
          }
        }
      }
    } while (l < nn - 1);
  }
  return nreal;
}

template <class PA, class PW>
OSFM_HD int real_eigenvalues10(PA a, PW wr) {
  return real_eigenvalues10_put(a, [&](int k, double v) { wr[k] = v; });
}

// Five-point solver in two stages, so that a kernel can give each the work space it needs (relpose.hip):
//   stage A  five_point_action_matrix: epipolar null space -> ten cubic constraints -> Gauss-Jordan -> the six non-trivial rows of the
//            10 x 10 action matrix of "multiply by x" (work space: basis 36, M 200 doubles, colperm 9 ints, per-lane dynamic indices)
//   stage B  five_point_solutions: its real eigenvalues (Hessenberg-QR) and, per eigenvalue, the eigenvector by complete-pivot
//            elimination -> essential matrices (work space: ONE 10 x 10 matrix; everything else is kept in registers through
//            select chains / unrolled loops, so that three wavefronts fit the LDS of a CU instead of one)
// D = "array of double", I = "array of int" (LaneArr or plain pointers).
// q[10] = a * b for linear polynomials read through any indexable type, same accumulation order as mul_lin_lin
template <class PA, class PB>
OSFM_HD void mul_lin_lin_t(PA a, PB b, double* q) {
  OSFM_UNROLL for (int k = 0; k < 10; k++) q[k] = 0.0;
  OSFM_UNROLL for (int i = 16; i < 20; i++)
    OSFM_UNROLL for (int j = 16; j < 20; j++) q[mul_slot(i, j) - 10] += a[i - 16] * b[j - 16];
}
template <class PB>
OSFM_HD void mul_quad_lin_t(const double* q, PB b, double* c) {
  OSFM_UNROLL for (int k = 0; k < 20; k++) c[k] = 0.0;
  OSFM_UNROLL for (int i = 10; i < 20; i++)
    OSFM_UNROLL for (int j = 16; j < 20; j++) c[mul_slot(i, j)] += q[i - 10] * b[j - 16];
}
// L(i, j) = row i of E times row j of E (quadratic): (E_i0 E_j0 + E_i1 E_j1) + E_i2 E_j2
template <class D>
OSFM_HD void five_point_L(D basis, int i, int j, double* L) {
  double q1[10], q2[10];
  mul_lin_lin_t(basis + (3 * i + 0) * 4, basis + (3 * j + 0) * 4, L);
  mul_lin_lin_t(basis + (3 * i + 1) * 4, basis + (3 * j + 1) * 4, q1);
  mul_lin_lin_t(basis + (3 * i + 2) * 4, basis + (3 * j + 2) * 4, q2);
  OSFM_UNROLL for (int k = 0; k < 10; k++) L[k] = (L[k] + q1[k]) + q2[k];
}

// Stage A.  b1, b2: 5 x 3 bearings (x2^T E x1 = 0).  On success (return 1) basis[36] holds the null space and M[0 .. 59] rows 0..5 of
// the action matrix (rows 6..9 are the constant rows e0, e1, e3, e6).  The 5 x 9 epipolar system lives in M's first 45 entries first.
template <class D, class I>
OSFM_HD int five_point_action_matrix(const double* b1, const double* b2, D basis, D M, I colperm) {
  D A = M;
  for (int i = 0; i < 5; i++)
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) A[i * 9 + 3 * r + c] = b2[3 * i + r] * b1[3 * i + c];
  if (!nullspace_5x9(A, basis, colperm)) return 0;
  // E(x, y, z) = x E0 + y E1 + z E2 + E3: entry (i, j) is the linear polynomial basis[(3 i + j) * 4 + 0..3]
#define OSFM_E(i, j) (basis + (3 * (i) + (j)) * 4)
  {  // det E = 0: row 0 = (c_a + c_b) + c_c, the three cofactor expansions along the third row
    double qa[10], qb[10], c[20];
    mul_lin_lin_t(OSFM_E(0, 1), OSFM_E(1, 2), qa);
    mul_lin_lin_t(OSFM_E(0, 2), OSFM_E(1, 1), qb);
    OSFM_UNROLL for (int k = 0; k < 10; k++) qa[k] -= qb[k];
    mul_quad_lin_t(qa, OSFM_E(2, 0), c);
    OSFM_UNROLL for (int k = 0; k < 20; k++) M[k] = c[k];
    mul_lin_lin_t(OSFM_E(0, 2), OSFM_E(1, 0), qa);
    mul_lin_lin_t(OSFM_E(0, 0), OSFM_E(1, 2), qb);
    OSFM_UNROLL for (int k = 0; k < 10; k++) qa[k] -= qb[k];
    mul_quad_lin_t(qa, OSFM_E(2, 1), c);
    OSFM_UNROLL for (int k = 0; k < 20; k++) M[k] += c[k];
    mul_lin_lin_t(OSFM_E(0, 0), OSFM_E(1, 1), qa);
    mul_lin_lin_t(OSFM_E(0, 1), OSFM_E(1, 0), qb);
    OSFM_UNROLL for (int k = 0; k < 10; k++) qa[k] -= qb[k];
    mul_quad_lin_t(qa, OSFM_E(2, 2), c);
    OSFM_UNROLL for (int k = 0; k < 20; k++) M[k] += c[k];
  }
  {  // (E E^T - 1/2 tr(E E^T) I) E = 0: rows 1..9.  Only the three L(i, .) of one i are alive at a time (the diagonal ones are
     // evaluated twice, once for the trace: the same operations give the same doubles)
    double tr[10];
    {
      double l0[10], l1[10];
      five_point_L(basis, 0, 0, l0);
      five_point_L(basis, 1, 1, l1);
      OSFM_UNROLL for (int k = 0; k < 10; k++) tr[k] = l0[k] + l1[k];
      five_point_L(basis, 2, 2, l0);
      OSFM_UNROLL for (int k = 0; k < 10; k++) tr[k] = (tr[k] + l0[k]) * 0.5;
    }
    for (int i = 0; i < 3; i++) {
      double Li[3][10];
      OSFM_UNROLL for (int q = 0; q < 3; q++) five_point_L(basis, i, q, Li[q]);
      OSFM_UNROLL for (int q = 0; q < 3; q++)
        if (q == i) {
          OSFM_UNROLL for (int k = 0; k < 10; k++) Li[q][k] -= tr[k];
        }
      for (int j = 0; j < 3; j++) {
        D le = M + 20 * (1 + 3 * i + j);
        double c[20];
        mul_quad_lin_t(Li[0], OSFM_E(0, j), c);
        OSFM_UNROLL for (int k = 0; k < 20; k++) le[k] = c[k];
        mul_quad_lin_t(Li[1], OSFM_E(1, j), c);
        OSFM_UNROLL for (int k = 0; k < 20; k++) le[k] += c[k];
        mul_quad_lin_t(Li[2], OSFM_E(2, j), c);
        OSFM_UNROLL for (int k = 0; k < 20; k++) le[k] += c[k];
      }
    }
  }
#undef OSFM_E
  // Gauss-Jordan on the cubic monomials (columns 0..9), partial pivoting
  for (int k = 0; k < 10; k++) {
    int pr = k;
    {
      double col[10];
      OSFM_UNROLL for (int i = 0; i < 10; i++) col[i] = fabs(M[i * 20 + k]);
      double bestv = 0.0;  // |M[pr][k]| of the running choice; the first candidate is row k itself
      OSFM_UNROLL for (int i = 0; i < 10; i++) {
        if (i == k) bestv = col[i];
        if (i > k && col[i] > bestv) {
          bestv = col[i];
          pr = i;
        }
      }
      if (!(bestv > 0)) return 0;
    }
    double rowk[20];
    {  // swap rows k and pr, scale the new row k
      double rowp[20];
      OSFM_UNROLL for (int j = 0; j < 20; j++) {
        rowk[j] = M[k * 20 + j];
        rowp[j] = M[pr * 20 + j];
      }
      OSFM_UNROLL for (int j = 0; j < 20; j++) M[pr * 20 + j] = rowk[j];
      double piv = 0.0;
      OSFM_UNROLL for (int j = 0; j < 10; j++)
        if (j == k) piv = rowp[j];
      const double ipv = 1.0 / piv;
      OSFM_UNROLL for (int j = 0; j < 20; j++) {
        rowk[j] = rowp[j] * ipv;
        M[k * 20 + j] = rowk[j];
      }
    }
    for (int i = 0; i < 10; i++) {
      if (i == k) continue;
      const double f = M[i * 20 + k];
      if (f == 0.0) continue;
      OSFM_UNROLL for (int j = 0; j < 20; j++) M[i * 20 + j] -= f * rowk[j];
    }
  }
  // action matrix of "multiply by x" on [xx xy yy xz yz zz x y z 1]: rows 0..5 = minus the right half of rows {0, 1, 2, 4, 5, 7} of
  // the eliminated M.  They take the start of M's storage, row by row, in an order in which every row of M is read before its
  // storage is overwritten (row r reads M row src[r] >= r, columns 10..19).
  {
    constexpr int src[6] = {0, 1, 2, 4, 5, 7};
    OSFM_UNROLL for (int r = 0; r < 6; r++) {
      double row[10];
      OSFM_UNROLL for (int j = 0; j < 10; j++) row[j] = -M[src[r] * 20 + 10 + j];
      OSFM_UNROLL for (int j = 0; j < 10; j++) M[r * 10 + j] = row[j];
    }
  }
  return 1;
}

// entry i (row-major, 0..99) of the action matrix given its rows 0..5
template <class PA>
OSFM_HD double action_matrix_entry(PA At6, int i) {
  return i < 60 ? At6[i] : ((i == 60 || i == 71 || i == 83 || i == 96) ? 1.0 : 0.0);
}

// Stage B.  At6: rows 0..5 of the action matrix (60 entries, read-only), basis: the null space (36, read-only), S: 100 doubles of
// work space.  Every real solution is handed to emit(E) -- a row-major 3 x 3 of unit Frobenius norm -- in eigenvalue order;
// returns how many (<= 10).
// The solution that belongs to one real eigenvalue lam of the action matrix: the eigenvector by Gauss-Jordan with complete pivoting on
// A - lam I (S: 100 doubles of work space), its entries (x, y, z, 1), E = x E0 + y E1 + z E2 + E3 normalised.  False when the
// elimination breaks down or the vector has no finite dehomogenisation.  (Round 6: a function of its own -- the GPU runs one lane per
// (problem, eigenvalue), solve5_stage_b2; five_point_solutions below calls it in eigenvalue order.)
template <class PA, class PB, class PS>
OSFM_HD bool five_point_solution_at(PA At6, PB basis, PS S, double lam, double* Em) {
  {
    OSFM_UNROLL for (int i = 0; i < 100; i++) S[i] = (i % 11 == 0) ? action_matrix_entry(At6, i) - lam : action_matrix_entry(At6, i);
    int cp[10];  // column permutation, in registers: k is a constant in the unrolled elimination, pc goes through a select chain
    OSFM_UNROLL for (int j = 0; j < 10; j++) cp[j] = j;
    int ok = 1;
    OSFM_UNROLL for (int k = 0; k < 9; k++) {
      if (ok) {
        int pr = k, pc = k;
        double best = 0;
        OSFM_UNROLL for (int i = k; i < 10; i++) {
          double row[10];
          OSFM_UNROLL for (int j = k; j < 10; j++) row[j] = fabs(S[i * 10 + j]);
          OSFM_UNROLL for (int j = k; j < 10; j++)
            if (row[j] > best) {
              best = row[j];
              pr = i;
              pc = j;
            }
        }
        if (!(best > 0)) {
          ok = 0;
        } else {
          double rowk[10];
          {  // row swap k <-> pr
            double rowp[10];
            OSFM_UNROLL for (int j = 0; j < 10; j++) {
              rowk[j] = S[k * 10 + j];
              rowp[j] = S[pr * 10 + j];
            }
            OSFM_UNROLL for (int j = 0; j < 10; j++) {
              S[pr * 10 + j] = rowk[j];
              S[k * 10 + j] = rowp[j];
            }
          }
          {  // column swap k <-> pc
            double ck[10], cq[10];
            OSFM_UNROLL for (int i = 0; i < 10; i++) {
              ck[i] = S[i * 10 + k];
              cq[i] = S[i * 10 + pc];
            }
            OSFM_UNROLL for (int i = 0; i < 10; i++) {
              S[i * 10 + k] = cq[i];
              S[i * 10 + pc] = ck[i];
            }
          }
          {
            const int t = cp[k];
            int u = t;
            OSFM_UNROLL for (int q = 0; q < 10; q++)
              if (q == pc) u = cp[q];
            OSFM_UNROLL for (int q = 0; q < 10; q++)
              if (q == pc) cp[q] = t;
            cp[k] = u;
          }
          const double ip = 1.0 / S[k * 10 + k];
          OSFM_UNROLL for (int j = 0; j < 10; j++) {
            rowk[j] = S[k * 10 + j] * ip;
            S[k * 10 + j] = rowk[j];
          }
          for (int i = 0; i < 10; i++) {
            if (i == k) continue;
            const double f = S[i * 10 + k];
            if (f == 0.0) continue;
            OSFM_UNROLL for (int j = 0; j < 10; j++) S[i * 10 + j] -= f * rowk[j];
          }
        }
      }
    }
    if (!ok) return false;
    // v[cp[9]] = 1, v[cp[k]] = -S[k][9]; only the entries 6..9 (x, y, z, 1) of the monomial vector are needed
    double v6 = 0.0, v7 = 0.0, v8 = 0.0, v9 = 0.0;
    OSFM_UNROLL for (int k = 0; k < 10; k++) {
      const double val = (k == 9) ? 1.0 : -S[k * 10 + 9];
      if (cp[k] == 6) v6 = val;
      if (cp[k] == 7) v7 = val;
      if (cp[k] == 8) v8 = val;
      if (cp[k] == 9) v9 = val;
    }
    if (v9 == 0.0) return false;
    const double x = v6 / v9, y = v7 / v9, z = v8 / v9;
    double nrm = 0.0;
    OSFM_UNROLL for (int i = 0; i < 9; i++) {
      Em[i] = x * basis[i * 4 + 0] + y * basis[i * 4 + 1] + z * basis[i * 4 + 2] + basis[i * 4 + 3];
      nrm += Em[i] * Em[i];
    }
    nrm = sqrt(nrm);
    if (!(nrm > 0) || !isfinite(nrm)) return false;
    OSFM_UNROLL for (int i = 0; i < 9; i++) Em[i] = Em[i] / nrm;
    return true;
  }
}
template <class PA, class PB, class PS, class Emit>
OSFM_HD int five_point_solutions(PA At6, PB basis, PS S, Emit emit) {
  OSFM_UNROLL for (int i = 0; i < 100; i++) S[i] = action_matrix_entry(At6, i);
  double wr[10];
  OSFM_UNROLL for (int q = 0; q < 10; q++) wr[q] = 0.0;
  const int nreal = real_eigenvalues10_put(S, [&](int k, double val) {
    OSFM_UNROLL for (int q = 0; q < 10; q++)
      if (q == k) wr[q] = val;
  });
  int count = 0;
  for (int e = 0; e < nreal && count < 10; e++) {
    double lam = 0.0;
    OSFM_UNROLL for (int q = 0; q < 10; q++)
      if (q == e) lam = wr[q];
    double Em[9];
    if (!five_point_solution_at(At6, basis, S, lam, Em)) continue;
    emit(Em);
    count++;
  }
  return count;
}
// both stages with the work space on the stack: Es receives up to 10 row-major 3 x 3 matrices; returns how many
OSFM_HD int essential_five_points(const double* b1, const double* b2, double* Es) {
  double basis[36], M[200], S[100];
  int colperm[9];
  if (!five_point_action_matrix(b1, b2, basis, M, colperm)) return 0;
  int n = 0;
  return five_point_solutions(M, basis, S, [&](const double* Em) {
    for (int i = 0; i < 9; i++) Es[9 * n + i] = Em[i];
    n++;
  });
}

// ---------------------------------------------------------------------------------------------------------------
// The same eigenvalues by a GROUP of lanes (round 6): the matrix lies in memory the group shares (LDS on the GPU), lane j of the group
// owns column j in the row operations and row j in the column operations, and every lane runs the scalar bookkeeping (pivot
// searches, shifts, the Householder vectors) redundantly from the same operands -- so the control flow is uniform inside a
// group and every element goes through the same expressions, in the same order, as in real_eigenvalues10_put: the same bits.
// What one lane did in ~30 k dependent LDS operations takes ~1/6 of that per lane here; the kernel's critical path is the
// scalar chain (divisions, one square root per reflection).
// OSFM_GROUP_FOR(j, glane) { body } runs the body for the calling lane on the device and for j = 0 .. 15 in turn on the host
// (tests/native/relpose_core_host.cpp): a body only touches what its own j owns between two OSFM_GROUP_SYNCs, so both
// orders of execution give the same memory.  OSFM_GROUP_SYNC keeps the compiler from moving shared accesses across it (the
// lanes of a wavefront run in lockstep and its LDS operations complete in order: no hardware barrier is needed).
// ---------------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define OSFM_GROUP_FOR(j, glane) for (int j = (glane), osfm_once_ = 1; osfm_once_; osfm_once_ = 0)
#define OSFM_GROUP_SYNC()                               \
  do {                                                  \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
#else
#define OSFM_GROUP_FOR(j, glane) for (int j = 0; j < 16; j++)
#define OSFM_GROUP_SYNC() do { } while (0)
#endif
constexpr int kEigGroup = 16;  // lanes per 10 x 10 eigenproblem (ten of them own a column / a row)

template <class PA, class Put>
OSFM_HD int real_eigenvalues10_group(PA a, int glane, Put put) {
  constexpr int n = 10;
  (void)glane;
  for (int m = 1; m < n - 1; m++) {
    double x = 0.0;
    int i = m;
    for (int j = m; j < n; j++) {
      const double c = a[j * n + m - 1];
      if (fabs(c) > fabs(x)) {
        x = c;
        i = j;
      }
    }
    if (i != m) {
      OSFM_GROUP_FOR(j, glane) if (j < n && j >= m - 1) {
        const double ri = a[i * n + j], rm = a[m * n + j];
        a[i * n + j] = rm;
        a[m * n + j] = ri;
      }
      OSFM_GROUP_SYNC();
      OSFM_GROUP_FOR(j, glane) if (j < n) {
        const double ci = a[j * n + i], cm = a[j * n + m];
        a[j * n + i] = cm;
        a[j * n + m] = ci;
      }
      OSFM_GROUP_SYNC();
    }
    if (x != 0.0) {
      for (i = m + 1; i < n; i++) {
        double y = a[i * n + m - 1];
        if (y != 0.0) {
          y /= x;
          OSFM_GROUP_SYNC();
          OSFM_GROUP_FOR(j, glane) {
            if (j == m - 1) a[i * n + m - 1] = y;
            if (j < n && j >= m) a[i * n + j] = a[i * n + j] - y * a[m * n + j];
          }
          OSFM_GROUP_SYNC();
          OSFM_GROUP_FOR(j, glane) if (j < n) a[j * n + m] = a[j * n + m] + y * a[j * n + i];
          OSFM_GROUP_SYNC();
        }
      }
    }
  }
  OSFM_GROUP_FOR(j, glane) if (j < n)
    for (int i = j + 2; i < n; i++) a[i * n + j] = 0.0;
  OSFM_GROUP_SYNC();
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
          OSFM_GROUP_SYNC();
          OSFM_GROUP_FOR(j, glane) if (j == 0) a[l * n + l - 1] = 0.0;
          OSFM_GROUP_SYNC();
          break;
        }
      }
      x = a[nn * n + nn];
      if (l == nn) {
        put(nreal, x + t);
        nreal++;
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
            put(nreal, x + z);
            put(nreal + 1, z != 0.0 ? x - w / z : x + z);
            nreal += 2;
          }
          nn -= 2;
        } else {
          if (its == 60) return nreal;
          if (its == 10 || its == 20) {
            t += x;
            OSFM_GROUP_SYNC();
            OSFM_GROUP_FOR(j, glane) if (j <= nn) a[j * n + j] -= x;
            OSFM_GROUP_SYNC();
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
          OSFM_GROUP_SYNC();
          OSFM_GROUP_FOR(i, glane) if (i >= m + 2 && i <= nn) {
            a[i * n + i - 2] = 0.0;
            if (i != m + 2) a[i * n + i - 3] = 0.0;
          }
          OSFM_GROUP_SYNC();
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
              OSFM_GROUP_SYNC();
              if (k == m) {
                if (l != m) OSFM_GROUP_FOR(j, glane) if (j == 0) a[k * n + k - 1] = -a[k * n + k - 1];
              } else {
                OSFM_GROUP_FOR(j, glane) if (j == 0) a[k * n + k - 1] = -s * x;
              }
              p += s;
              x = p / s;
              y = q / s;
              z = r / s;
              q /= p;
              r /= p;
              const bool three = (k != nn - 1);
              OSFM_GROUP_FOR(j, glane) if (j >= k && j <= nn) {  // rows k, k + 1 (, k + 2): lane j owns column j
                const double r0 = a[k * n + j], r1 = a[(k + 1) * n + j], r2 = three ? a[(k + 2) * n + j] : 0.0;
                double pj = r0 + q * r1;
                if (three) {
                  pj += r * r2;
                  a[(k + 2) * n + j] = r2 - pj * z;
                }
                a[(k + 1) * n + j] = r1 - pj * y;
                a[k * n + j] = r0 - pj * x;
              }
              OSFM_GROUP_SYNC();
              const int mmin = nn < k + 3 ? nn : k + 3;
              OSFM_GROUP_FOR(i, glane) if (i >= l && i <= mmin) {  // columns k, k + 1 (, k + 2): lane i owns row i
                const double c0 = a[i * n + k], c1 = a[i * n + k + 1], c2 = three ? a[i * n + k + 2] : 0.0;
                double pi = x * c0 + y * c1;
                if (three) {
                  pi += z * c2;
                  a[i * n + k + 2] = c2 - pi * r;
                }
                a[i * n + k + 1] = c1 - pi * q;
                a[i * n + k] = c0 - pi;
              }
              OSFM_GROUP_SYNC();
            }
          }
        }
      }
    } while (l < nn - 1);
  }
  return nreal;
}

// ---------------------------------------------------------------------------------------------------------------
// Relative pose from an essential matrix (geometry/relative_pose.h:12-84), one-sided Jacobi SVD of a 3 x 3.
// ---------------------------------------------------------------------------------------------------------------
OSFM_HD void svd3(const double* A, double* U, double* S, double* V) {
  double G[9];
  for (int i = 0; i < 9; i++) G[i] = A[i];
  for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  OSFM_NOUNROLL for (int sweep = 0; sweep < 60; sweep++) {
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
  // columns sorted by decreasing norm (the same three compare-and-swap steps as a selection sort on an index array), with the
  // run-time column indices resolved through selects so that nothing is indexed dynamically
  double nrm[3];
  OSFM_UNROLL for (int j = 0; j < 3; j++) nrm[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
  int o0 = 0, o1 = 1, o2 = 2;
  auto pick = [](const double* a, int stride, int i) { return i == 0 ? a[0] : (i == 1 ? a[stride] : a[2 * stride]); };
  if (pick(nrm, 1, o1) > pick(nrm, 1, o0)) {
    const int t = o0;
    o0 = o1;
    o1 = t;
  }
  if (pick(nrm, 1, o2) > pick(nrm, 1, o0)) {
    const int t = o0;
    o0 = o2;
    o2 = t;
  }
  if (pick(nrm, 1, o2) > pick(nrm, 1, o1)) {
    const int t = o1;
    o1 = o2;
    o2 = t;
  }
  double Vs[9];
  OSFM_UNROLL for (int j = 0; j < 3; j++) {
    const int o = j == 0 ? o0 : (j == 1 ? o1 : o2);
    const double nj = pick(nrm, 1, o);
    S[j] = nj;
    OSFM_UNROLL for (int k = 0; k < 3; k++) {
      Vs[3 * k + j] = pick(V + 3 * k, 1, o);
      U[3 * k + j] = nj > 0 ? pick(G + 3 * k, 1, o) / nj : 0.0;
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
  OSFM_NOUNROLL for (int i = 0; i < 2; i++) {
    double t[3] = {U[2], U[5], U[8]};
    if (i == 1)
      for (int k = 0; k < 3; k++) t[k] = -t[k];
    const double tn = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    for (int k = 0; k < 3; k++) t[k] /= tn;
    OSFM_NOUNROLL for (int j = 0; j < 2; j++) {
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
      OSFM_NOUNROLL for (int s = 0; s < n; s++) {
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
template <class PA, class PW, class PV>
OSFM_HD void jacobi_eig9(PA A, PW w, PV V) {
  constexpr int n = 9;
  OSFM_UNROLL for (int i = 0; i < n * n; i++) V[i] = (i % (n + 1) == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; sweep++) {
    double off = 0.0;
    OSFM_UNROLL for (int p = 0; p < n - 1; p++)
      OSFM_UNROLL for (int q = p + 1; q < n; q++) {
        const double apq = A[p * n + q];
        off += apq * apq;
      }
    if (!(off > 1e-300)) break;
    // the 36 rotations of a sweep are unrolled: p and q are then constants, and a matrix kept in a local array stays in registers
    OSFM_UNROLL for (int p = 0; p < n - 1; p++)
      OSFM_UNROLL for (int q = p + 1; q < n; q++) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        {  // columns p, q of A
          double cp[n], cq[n];
          OSFM_UNROLL for (int k = 0; k < n; k++) {
            cp[k] = A[k * n + p];
            cq[k] = A[k * n + q];
          }
          OSFM_UNROLL for (int k = 0; k < n; k++) {
            A[k * n + p] = c * cp[k] - s * cq[k];
            A[k * n + q] = s * cp[k] + c * cq[k];
          }
        }
        {  // rows p, q of A; columns p, q of V
          double rp[n], rq[n], vp[n], vq[n];
          OSFM_UNROLL for (int k = 0; k < n; k++) {
            rp[k] = A[p * n + k];
            rq[k] = A[q * n + k];
            vp[k] = V[k * n + p];
            vq[k] = V[k * n + q];
          }
          OSFM_UNROLL for (int k = 0; k < n; k++) {
            A[p * n + k] = c * rp[k] - s * rq[k];
            A[q * n + k] = s * rp[k] + c * rq[k];
            V[k * n + p] = c * vp[k] - s * vq[k];
            V[k * n + q] = s * vp[k] + c * vq[k];
          }
        }
      }
  }
  OSFM_UNROLL for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}
// EssentialNPoints (geometry/essential.h:162-192) with foundation::SolveAX0 (foundation/numeric.h:20-43): 0 or 1 model
// AtA, V: 81 doubles each, w: 9 -- work space, one per problem (LaneArr in the solver kernel, stack arrays below)
template <class D>
OSFM_HD int essential_n_points_ws(const double* b1, const double* b2, const int* idx, int count, double* E, D AtA, D V, D w) {
  if (count < 9) return 0;
  // A^T A = sum over the samples of row^T row, every entry summed over s = 0, 1, ... from 0.0.  Three rows of the result at a time
  // stay in registers (27 accumulators) instead of 81 read-modify-writes of the work space per sample.
  OSFM_UNROLL for (int i0 = 0; i0 < 9; i0 += 3) {
    double acc[3][9];
    OSFM_UNROLL for (int a = 0; a < 3; a++)
      OSFM_UNROLL for (int j = 0; j < 9; j++) acc[a][j] = 0.0;
    for (int s = 0; s < count; s++) {
      const double *x1 = b1 + 3 * idx[s], *x2 = b2 + 3 * idx[s];
      double row[9];
      OSFM_UNROLL for (int r = 0; r < 3; r++)
        OSFM_UNROLL for (int c = 0; c < 3; c++) row[3 * r + c] = x2[r] * x1[c];
      OSFM_UNROLL for (int a = 0; a < 3; a++)
        OSFM_UNROLL for (int j = 0; j < 9; j++) acc[a][j] += row[i0 + a] * row[j];
    }
    OSFM_UNROLL for (int a = 0; a < 3; a++)
      OSFM_UNROLL for (int j = 0; j < 9; j++) AtA[9 * (i0 + a) + j] = acc[a][j];
  }
  jacobi_eig9(AtA, w, V);
  // smallest and second smallest eigenvalue (first occurrence wins), without indexing w / V by a run-time value
  int lo = 0, lo2 = -1;
  double w_lo = w[0], w_lo2 = 0.0;
  OSFM_UNROLL for (int i = 1; i < 9; i++) {
    const double wi = w[i];
    if (wi < w_lo) {
      w_lo = wi;
      lo = i;
    }
  }
  OSFM_UNROLL for (int i = 0; i < 9; i++) {
    const double wi = w[i];
    if (i != lo && (lo2 < 0 || wi < w_lo2)) {
      w_lo2 = wi;
      lo2 = i;
    }
  }
  const double s_small = sqrt(fmax(w_lo, 0.0)), s_next = sqrt(fmax(w_lo2, 0.0));
  if (!(s_next / s_small > 4.0)) return 0;
  double Em[9];
  OSFM_UNROLL for (int i = 0; i < 9; i++) {
    double e = 0.0;
    OSFM_UNROLL for (int c = 0; c < 9; c++)
      if (c == lo) e = V[9 * i + c];
    Em[i] = e;
  }
  double U[9], S[3], Vv[9];
  svd3(Em, U, S, Vv);
  const double d = 0.5 * (S[0] + S[1]);
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) E[3 * a + b] = d * (U[3 * a] * Vv[3 * b] + U[3 * a + 1] * Vv[3 * b + 1]);
  return 1;
}
OSFM_HD int essential_n_points(const double* b1, const double* b2, const int* idx, int count, double* E) {
  double AtA[81], w[9], V[81];
  return essential_n_points_ws<double*>(b1, b2, idx, count, E, AtA, V, w);
}
// RelativePose::Evaluate (robust/relative_pose_model.h): 1 - mean cosine between the bearings and the midpoint
// x, y: the two bearings already divided by their norms (the first step of Evaluate)
OSFM_HD double relpose_error_unit(const double* RT, const double* x, const double* y) {
  double R[9], t[3];
  for (int a = 0; a < 3; a++) {
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
// d.first.normalized() / d.second.normalized() of Evaluate: done once per correspondence, every model then scores the unit bearings
OSFM_HD void normalise_bearing(const double* v, double* u) {
  const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  for (int a = 0; a < 3; a++) u[a] = v[a] / nv;
}
OSFM_HD double relpose_error(const double* RT, const double* x0, const double* y0) {
  double x[3], y[3];
  normalise_bearing(x0, x);
  normalise_bearing(y0, y);
  return relpose_error_unit(RT, x, y);
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

// A value with NDUAL of its six derivatives (forward mode).  Every derivative component goes through its own copy of the same
// expression, so a residual evaluated in two halves of three derivatives (refine_residual: kRefineDualWidth) has the bits of one
// evaluation with all six -- and half the live registers: the refinement kernel fits two wavefronts on a SIMD (round 6; with six
// derivatives it held 512 registers and spilled).  The value part is computed once per half.
template <int NDUAL>
struct Dn {
  double v, d[NDUAL];
};
constexpr int kRefineDualWidth = 3;
typedef Dn<kRefineDualWidth> D6;  // (the name predates the halves)
OSFM_HD D6 dc(double c) {
  D6 r;
  r.v = c;
  for (int i = 0; i < kRefineDualWidth; i++) r.d[i] = 0;
  return r;
}
OSFM_HD D6 dvar(double c, int k) {  // k outside 0 .. kRefineDualWidth - 1: a parameter of the other half, a constant here
  D6 r = dc(c);
  for (int i = 0; i < kRefineDualWidth; i++)
    if (i == k) r.d[i] = 1.0;
  return r;
}
OSFM_HD D6 dadd(D6 a, const D6& b) {
  for (int i = 0; i < kRefineDualWidth; i++) a.d[i] += b.d[i];
  a.v += b.v;
  return a;
}
OSFM_HD D6 dsub(D6 a, const D6& b) {
  for (int i = 0; i < kRefineDualWidth; i++) a.d[i] -= b.d[i];
  a.v -= b.v;
  return a;
}
OSFM_HD D6 dneg(D6 a) {
  for (int i = 0; i < kRefineDualWidth; i++) a.d[i] = -a.d[i];
  a.v = -a.v;
  return a;
}
OSFM_HD D6 dmul(const D6& a, const D6& b) {
  D6 r;
  r.v = a.v * b.v;
  for (int i = 0; i < kRefineDualWidth; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
OSFM_HD D6 dmulc(D6 a, double c) {
  for (int i = 0; i < kRefineDualWidth; i++) a.d[i] *= c;
  a.v *= c;
  return a;
}
OSFM_HD D6 ddiv(const D6& a, const D6& b) {
  D6 r;
  const double ib = 1.0 / b.v;
  r.v = a.v * ib;
  for (int i = 0; i < kRefineDualWidth; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
OSFM_HD D6 dsqrt(const D6& a) {
  D6 r;
  r.v = sqrt(a.v);
  const double h = 0.5 / r.v;
  for (int i = 0; i < kRefineDualWidth; i++) r.d[i] = a.d[i] * h;
  return r;
}
OSFM_HD D6 dsin(const D6& a) {
  D6 r;
  r.v = sin(a.v);
  const double c = cos(a.v);
  for (int i = 0; i < kRefineDualWidth; i++) r.d[i] = a.d[i] * c;
  return r;
}
OSFM_HD D6 dcos(const D6& a) {
  D6 r;
  r.v = cos(a.v);
  const double s = -sin(a.v);
  for (int i = 0; i < kRefineDualWidth; i++) r.d[i] = a.d[i] * s;
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
OSFM_HD void refine_residual_half(int i, const double* par, const double* xs, const double* ys, int k0, double* val, double* grad) {
  D6 rot[3], tr[3], rot_t[3];
  for (int k = 0; k < 3; k++) {
    rot[k] = dvar(par[k], k - k0);
    tr[k] = dvar(par[3 + k], 3 + k - k0);
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
  *val = r.v;
  for (int k = 0; k < kRefineDualWidth; k++) grad[k] = r.d[k];
}
OSFM_HD void refine_residual(int i, const double* par, const double* xs, const double* ys, double* out) {
  static_assert(6 % kRefineDualWidth == 0, "the six derivatives in equal parts");
  OSFM_NOUNROLL for (int k0 = 0; k0 < 6; k0 += kRefineDualWidth) refine_residual_half(i, par, xs, ys, k0, out, out + 1 + k0);
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
  double R[9], x[6];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) R[3 * a + b] = RT[4 * a + b];
  rotmat_to_aa(R, x);
  for (int a = 0; a < 3; a++) x[3 + a] = -(R[a] * RT[3] + R[3 + a] * RT[7] + R[6 + a] * RT[11]);
  double scal[6], g[6], cost = 0;  // J^T J lives with the evaluator (ev.keep / ev.kept, entries 0 .. 35): the same in every lane, and 72 registers otherwise
  int have_scale = 0, it = 0;
  // every sum below runs over the residuals i = 0 .. NR - 1 in order, from 0.0 (the oracle's order); ev.reduce() may give each
  // sum to a different lane
  // fresh = false: the evaluator's buffers already hold the residuals AND Jacobians at x -- the trial evaluation of an accepted step
  // (every evaluation forms the duals: eval's second argument is a wish, not a switch), the same operands through the same
  // operations, so evaluating again would reproduce them bit for bit (round 6: a third of the refinement's evaluations)
  auto update = [&](bool fresh) {
    if (fresh) ev.eval(x, 1);
    if (!have_scale) {
      double ss[6];
      ev.reduce(6, [&](int k, int i) { return ev.jac(i, k) * ev.jac(i, k); }, ss);
      for (int k = 0; k < 6; k++) scal[k] = 1.0 / (1.0 + sqrt(ss[k]));
      have_scale = 1;
    }
    double sums[43];
    // sum q: gradient entry q < 6 (J_q scal_q) (-r); J^T J entry (a, b) = ((q - 6) / 6, (q - 6) % 6) for q < 42; q = 42: (-r) (-r).  Written as
    // one product of two scaled columns of [r | J] (column 0 = r with factor -1: the negation, exactly) so that the lanes that own
    // different kinds of sums run the same instructions
    ev.reduce(43, [&](int q, int i) {
      const int ca = q < 6 ? 1 + q : (q < 42 ? 1 + (q - 6) / 6 : 0), cb = q < 6 ? 0 : (q < 42 ? 1 + (q - 6) % 6 : 0);
      double sa = -1.0, sb = -1.0;
      for (int k = 0; k < 6; k++) {
        if (ca == 1 + k) sa = scal[k];
        if (cb == 1 + k) sb = scal[k];
      }
      return (ev.val(i, ca) * sa) * (ev.val(i, cb) * sb);
    }, sums);
    for (int a = 0; a < 6; a++) g[a] = sums[a];
    for (int k = 0; k < 36; k++) ev.keep(k, sums[6 + k]);
    cost = sums[42] * 0.5;
  };
  update(true);
  if (costs) costs[0] = cost;
  double gmax = 0;
  for (int k = 0; k < 6; k++) gmax = fmax(gmax, fabs(g[k]));
  double u = 1.0 / 1e4, v = 2.0;
  if (!(gmax < 1e-10) && !(cost < kEps)) {
    for (it = 1; it < iterations; it++) {
      double reg[36], step[6], dx[6], xn[6];
      for (int k = 0; k < 36; k++) reg[k] = ev.kept(k);
      for (int k = 0; k < 6; k++) reg[7 * k] += u * fmin(fmax(ev.kept(7 * k), 1e-6), 1e32);
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
      ev.eval(xn, 1);
      double fn2 = 0;
      ev.reduce(1, [&](int, int i) { return ev.res(i) * ev.res(i); }, &fn2);
      const double cost_change = 2.0 * cost - fn2;
      double mc = 0;
      for (int a = 0; a < 6; a++) {
        double s = 2.0 * g[a];
        for (int b = 0; b < 6; b++) s -= ev.kept(6 * a + b) * step[b];
        mc += step[a] * s;
      }
      const double rho = cost_change / mc;
      if (rho > 0) {
        for (int k = 0; k < 6; k++) x[k] = xn[k];
        update(false);
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
