// ba.hip -- global bundle adjustment on one MI355X (gfx950), fp64.
//
// Replaces bundle::BundleAdjuster::Run (opensfm/src/bundle/src/bundle_adjuster.cc:595-1121) as
// driven by sfm::BAHelpers::Bundle (opensfm/src/sfm/src/ba_helpers.cc:581-763): robustified
// Levenberg-Marquardt over reprojection residuals (projection_errors.h:59-208) with the Ceres
// trust-region rules the reference inherits (defaults, bundle_adjuster.cc:1104-1113).
//
// Where Ceres' SPARSE_SCHUR factorises the reduced camera system, this path never forms it:
//   * one thread per observation builds residual + analytic Jacobian blocks (SoA, coalesced);
//   * the 3x3 point blocks are eliminated on the fly (thread per point, observations stored
//     point-major so a track is a contiguous segment);
//   * the reduced system S = Jc^T (I - Jp Hpp^-1 Jp^T) Jc + D is applied implicitly inside a
//     block-Jacobi preconditioned CG (two streaming passes over the observations per mat-vec:
//     point-major gather + shot-major segmented reduce; no atomics, bit-reproducible);
//   * all CG scalars stay on the device; the host only polls the residual every few iterations.
// Every kernel here is HBM-bandwidth bound (gather/scatter + streaming), see DESIGN.md.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "ba_math.h"
#include "osfm_internal.h"

using namespace osfm_ba;

namespace {

constexpr int kCoopObs = 256;  // observations per workgroup of the cooperative per-track kernels (mat-vec, point gradient)

// Streaming stores (round 6): the Jacobian rows are written once by the evaluation and read much later by other kernels, 680 MB that no cache
// holds -- written past the caches (`nt`) they cost 3 % of an LM iteration less (3.08 -> 3.00 ms at configs[4], profiles/r06_ba_variants2.json).
// OSFM_BA_NT_LEVEL (build knob): 0 none, 1 the evaluation's rows (default), 2 every streaming store (the E blocks, w, the border's w), 3 also the
// streaming LOADS of the rows -- levels 2 and 3 measured level with 1 (profiles/r06_ba_variants3.json: w is read back by the very next kernel)
#ifndef OSFM_BA_NT_LEVEL
#define OSFM_BA_NT_LEVEL 1
#endif
typedef double osfm_v2d __attribute__((ext_vector_type(2)));
template <int LEVEL>
__device__ __forceinline__ void st_stream(double *p, double v) {
  if (OSFM_BA_NT_LEVEL >= LEVEL) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int LEVEL>
__device__ __forceinline__ void st_stream2(double *p, double a, double b) {  // p: 16-byte aligned
  osfm_v2d v;
  v.x = a;
  v.y = b;
  if (OSFM_BA_NT_LEVEL >= LEVEL) __builtin_nontemporal_store(v, reinterpret_cast<osfm_v2d *>(p));
  else *reinterpret_cast<osfm_v2d *>(p) = v;
}
__device__ __forceinline__ double ld_stream(const double *p) {
  if (OSFM_BA_NT_LEVEL >= 3) return __builtin_nontemporal_load(p);
  return *p;
}

// the six entries of Jk (d residual / d [k1 k2 focal]) from the point (u, v) of the undistorted image plane.  ONE function for the evaluation and
// for the point-major kernels, which rebuild Jk from the (u, v) their rows keep (round 6): the same operations in the same order, the same bits.
__device__ __forceinline__ void jk_entries(double inv_sigma, double k1, double k2, double f, double u, double v, double *Jk) {
  const double r2 = u * u + v * v;
  const double d = 1.0 + r2 * (k1 + k2 * r2);
  Jk[0] = inv_sigma * f * r2 * u;
  Jk[1] = inv_sigma * f * r2 * r2 * u;
  Jk[2] = inv_sigma * d * u;
  Jk[3] = inv_sigma * f * r2 * v;
  Jk[4] = inv_sigma * f * r2 * r2 * v;
  Jk[5] = inv_sigma * d * v;
}

// uv (optional, with JAC): the point (u, v) Jk was formed from -- (0, 0) for the constant cameras of another model, whose Jk is zero
template <bool JAC>
__device__ __forceinline__ void project_obs(int model, const double *X, const double *pose, const double *R, const double *dR,
                                            const double *cam, double ox, double oy, double inv_sigma, double *res,
                                            double *Jp, double *Jc, double *Jk, double *uv = nullptr) {
  const double p[3] = {X[0] - pose[3], X[1] - pose[4], X[2] - pose[5]};
  double Xc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) Xc[i] = R[3 * i] * p[0] + R[3 * i + 1] * p[1] + R[3 * i + 2] * p[2];
  if (model >= 2) {  // constant camera of another 2-D model: cam points at its 16 native parameters
    double out[2], J[6], M[6];
    project_generic<JAC>(model, cam, Xc, out, J);
    res[0] = inv_sigma * (out[0] - ox);
    res[1] = inv_sigma * (out[1] - oy);
    if (!JAC) return;
    for (int j = 0; j < 6; j++) M[j] = inv_sigma * J[j];
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) {
        const double mr = M[3 * i] * R[j] + M[3 * i + 1] * R[3 + j] + M[3 * i + 2] * R[6 + j];
        Jp[3 * i + j] = mr;
        Jc[6 * i + 3 + j] = -mr;
      }
    for (int k = 0; k < 3; k++) {
      double q[3];
      for (int i = 0; i < 3; i++) q[i] = dR[9 * k + 3 * i] * p[0] + dR[9 * k + 3 * i + 1] * p[1] + dR[9 * k + 3 * i + 2] * p[2];
      for (int i = 0; i < 2; i++) Jc[6 * i + k] = -(M[3 * i] * q[0] + M[3 * i + 1] * q[1] + M[3 * i + 2] * q[2]);
    }
    for (int i = 0; i < 6; i++) Jk[i] = 0.0;
    if (uv) uv[0] = uv[1] = 0.0;
    return;
  }
  const double k1 = cam[0], k2 = cam[1], f = cam[2];
  double u, v, jp[6];
  project_stage<JAC>(model, Xc, u, v, jp);
  const double r2 = u * u + v * v;
  const double d = 1.0 + r2 * (k1 + k2 * r2);
  res[0] = inv_sigma * (f * d * u - ox);
  res[1] = inv_sigma * (f * d * v - oy);
  if (!JAC) return;
  const double x2 = u * u, y2 = v * v, x4 = x2 * x2, y4 = y2 * y2;
  const double jd00 = 5.0 * k2 * x4 + 3.0 * k1 * x2 + 6.0 * k2 * x2 * y2 + k2 * y4 + k1 * y2 + 1.0;
  const double jd01 = u * (2.0 * k1 * v + 4.0 * k2 * v * r2);
  const double jd10 = v * (2.0 * k1 * u + 4.0 * k2 * u * r2);
  const double jd11 = 5.0 * k2 * y4 + 3.0 * k1 * y2 + 6.0 * k2 * y2 * x2 + k2 * x4 + k1 * x2 + 1.0;
  double M[6];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    M[j] = inv_sigma * f * (jd00 * jp[j] + jd01 * jp[3 + j]);
    M[3 + j] = inv_sigma * f * (jd10 * jp[j] + jd11 * jp[3 + j]);
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double mr = M[3 * i] * R[j] + M[3 * i + 1] * R[3 + j] + M[3 * i + 2] * R[6 + j];
      Jp[3 * i + j] = mr;
      Jc[6 * i + 3 + j] = -mr;
    }
#pragma unroll
  for (int k = 0; k < 3; k++) {
    double q[3];
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = dR[9 * k + 3 * i] * p[0] + dR[9 * k + 3 * i + 1] * p[1] + dR[9 * k + 3 * i + 2] * p[2];
#pragma unroll
    for (int i = 0; i < 2; i++) Jc[6 * i + k] = -(M[3 * i] * q[0] + M[3 * i + 1] * q[1] + M[3 * i + 2] * q[2]);
  }
  jk_entries(inv_sigma, k1, k2, f, u, v, Jk);
  if (uv) {
    uv[0] = u;
    uv[1] = v;
  }
}

__device__ __forceinline__ void loss_eval(int loss, double a, double s, double &rho, double &rho1) {
  const double b = a * a;
  switch (loss) {
    case OSFM_LOSS_SOFTLONE: {
      const double sum = 1.0 + s / b, tmp = sqrt(sum);
      rho = 2.0 * b * (tmp - 1.0);
      rho1 = 1.0 / tmp;
    } break;
    case OSFM_LOSS_HUBER:
      if (s > b) {
        const double r = sqrt(s);
        rho = 2.0 * a * r - b;
        rho1 = a / r;
      } else {
        rho = s;
        rho1 = 1.0;
      }
      break;
    case OSFM_LOSS_CAUCHY: {
      const double sum = 1.0 + s / b;
      rho = b * log(sum);
      rho1 = 1.0 / sum;
    } break;
    default:
      rho = s;
      rho1 = 1.0;
  }
}

// block-wide sum of NV values per thread; result valid on thread 0
template <int NV>
__device__ __forceinline__ void block_sum(double *v, double *lds /* [4*NV] */) {
#pragma unroll
  for (int k = 0; k < NV; k++)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v[k] += __shfl_xor(v[k], m);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  if (nw == 1) return;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; k++) lds[w * NV + k] = v[k];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < NV; k++) {
      double s = lds[k];
      for (int w2 = 1; w2 < nw; w2++) s += lds[w2 * NV + k];
      v[k] = s;
    }
}

// Sum of NV values per thread over a workgroup of W wavefronts, wavefront by wavefront in a fixed order; the result is valid on thread 0.
// (the per-shot kernels: one wavefront per shot when there are thousands of shots, W = 4 when there are few -- a 48-shot local bundle
// adjustment gave 48 wavefronts a thirteen-step loop of ~500 fp64 instructions each and left the other 1 000 SIMDs idle)
template <int NV, int W>
__device__ __forceinline__ void shot_sum(double (&v)[NV], double *lds /* [W * NV] when W > 1 */) {
#pragma unroll
  for (int i = 0; i < NV; i++)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v[i] += __shfl_xor(v[i], m);
  if (W == 1) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; i++) lds[w * NV + i] = v[i];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int i = 0; i < NV; i++) {
      double t = lds[i];
#pragma unroll
      for (int q = 1; q < W; q++) t += lds[q * NV + i];
      v[i] = t;
    }
}
constexpr int kShotWavesSmall = 4;   // wavefronts per shot of the per-shot kernels when the problem has at most kShotWavesBelow shots
constexpr int kShotWavesBelow = 1024;

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  The per-shot kernels gather rows that the neighbouring shots
// want too (the E blocks / w entries of the points they share), so every XCD gets a CONTIGUOUS range of shots: band assembly 1.76 ->
// 1.11 ms at configs[4] from this mapping alone.
__device__ __forceinline__ long xcd_contiguous(long b, long n) {  // bijection for any n
  const long q = n >> 3, r = n & 7;
  const long xcd = b & 7, within = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
}

// ------------------------------------------------------------------------------------------
// device problem image
// ------------------------------------------------------------------------------------------
// generic mode (osfm_bundle_solve; kernels in ba_generic.inc): what the device image holds beyond the [k1 k2 focal] configuration.
// In that mode a "shot" of the arrays below is a rig INSTANCE (the 6-parameter unknown of the band), a VIEW is one of the caller's
// shots = (instance, rig camera, camera); the views of an instance are consecutive.
struct GenDev {
  int NRr;  // residual rows per observation: 2, or 3 when a spherical camera is present
  int KW;   // border slots per observation row: camera parameters then the rig camera pose of the widest view
  int NB;   // border unknowns: nred = 6 S + NB
  int NV, NRC;
  int oJp, oJc, oJb, ncomp;  // component offsets of a row: res [0, NRr) | Jp | Jc | Jb, ncomp in all
  const int *view_inst, *view_rc, *view_cam;  // NV
  const int *view_col;                        // NV x KW: border column of a slot, -1 = none
  const long *view_off;                       // NV + 1: shot-major segments of the views
  const int *inst_view0;                      // S + 1: first view of an instance
  const int *o_view, *sm_view;                // per row, point-major / shot-major
  const unsigned char *o_kind, *sm_kind;      // per row or null: 0 reprojection, 1 depth prior on z, 2 radial depth prior
  double *cam, *cam_n;                        // NC x 16 native parameters, current / candidate
  const double *cam_prior, *cam_sigma;        // NC x 16
  double *rc, *rc_n;                          // NRC x 6 rig camera poses
  const double *rc_prior, *rc_sigma;          // NRC x 6 or null
  double *bias, *bias_n;                      // NC x 7
  double *rcR;                                // NRC x 36: rotation blocks of the rig cameras (shot_rot_kernel's layout)
  const unsigned char *rc_useful;             // the rig camera enters the projection (not a constant identity, bundle_adjuster.cc:17-20)
  const int *cam_col, *rc_col, *bias_col;     // first border column of a free block, -1 = constant
  const unsigned char *col_slot;              // NB x NV: the slot of view v that holds border column j, 255 = none
  double *vpart;                              // NV x 2 KW: per-view partial sums on the border slots
  double *yv;                                 // NV x KW: the border step gathered per view
  double *PI, *Bpri, *Cpri;                   // prior blocks: S x 36, NB x 6 S, NB x NB
  double *gpri;                               // nred: the priors' gradient
  double *bdot;                               // NB
  const double *gps, *gps_sigma;              // S x 3 each or null: instance position priors
  const int *inst_bias_cam;                   // S
  const double *up, *up_sigma;                // NV x 3, NV or null (unit vectors are formed in the kernel)
  const double *pan, *pan_sigma, *tilt, *tilt_sigma, *roll, *roll_sigma;  // NV each or null
  const double *pt_prior, *pt_prior_sigma;    // P x 3 each or null
  const unsigned char *pt_prior_alt;          // P
};

struct Dev {
  int S, P, NC;
  long M;
  int nred, cam0;
  // parameters (current / candidate)
  double *cams, *poses, *pts, *cams_n, *poses_n, *pts_n;
  const double *cam_prior, *cam_sigma, *gps, *gps_sigma;
  // absolute up-vector prior per shot (absolute_motion_errors.h:12-39, CauchyLoss(1), bundle_adjuster.cc:955-970)
  const double *up, *up_sigma;  // unit vectors (3 per shot), sd (<= 0: none) -- or null
  double *up_r, *up_J;          // corrected residual (3) and Jacobian w.r.t. the rotation (3x3) per shot
  double *prior_rot;            // J^T J of that prior: symmetric 3x3 per shot, packed (00,10,11,20,21,22)
  const uint8_t *cam_fixed, *shot_fixed, *point_fixed;
  const int *shot_camera;
  const int *cam_model;  // per camera (or null: all PERSPECTIVE)
  const double *cam_ext; // 16 native parameters per camera, for the constant cameras of models >= 2
  // observations, point-major
  const int *o_shot, *o_point;
  const double *o_x, *o_y, *o_sigma;
  const long *pt_off;    // P + 1
  const int *wg_pt;      // nwg + 1: point ranges of the cooperative mat-vec workgroups (<= kCoopObs observations each)
  int nwg;
  const long *shot_off;  // S + 1
  const int *shot_obs;   // M: indices into the point-major arrays, grouped by shot
  double *shotR;         // S x 36
  // per-observation residual + Jacobian blocks, components: res(2) Jp(6) Jc(12) Jk(6), kept once, in point-major order:
  double *Epm;           // [M][18] AoS, point-major: E_o = Jc_o^T Jp_o (6x3), operand of the band assembly
  double *Jpm;           // [kRowComps][M] SoA in POINT-major observation order (thread-per-observation kernels coalesce); generic mode: [g.ncomp][M]
  double *sm_wt;         // [M] in SHOT-major order: the robust weight sqrt(rho') of every observation at the linearisation point (written by the
                         // per-shot gradient kernel, read by the other per-shot kernels, which recompute the Jacobian rows they need -- sm_row; rounds 1-5
                         // kept a second, shot-major copy of all 26 components instead: a second evaluation launch, 208 bytes per observation written and re-read)
  const int *sm_shot, *sm_point;        // observation data in shot-major order (static)
  const double *sm_x, *sm_y, *sm_sigma;
  double *w;             // 2 x M
  // points
  double *g_pt, *Hpp, *Hhat, *sc_pt, *D_pt, *d_pt;
  // reduced unknowns (6 per shot, then 3 per camera)
  double *g_red, *diag_red, *prior_diag, *sc_red, *D_red;
  double *Hcc;      // 21 x S (obs part of the shot block), then 6 x NC
  double *Binv;     // 36 x S, then 9 x NC
  double *part;     // per-shot partials for camera blocks: 9 x S
  double *camred;   // 9 x NC: per-camera sums of `part` (deterministic block reduction)
  // banded preconditioner (shot-shot part of the Schur complement inside a block band of half-width bw)
  int bw;           // block half-bandwidth actually used (0: block-Jacobi only)
  double *band;     // S x (bw+1) x 36: block (s, s-k), after factorisation the Cholesky factor L
  const unsigned char *bslot;  // S x (bw + 1) bytes or null: band_assemble_compact_kernel's accumulator slot of block (s, s - dk), 255 = no common point
  double *dinv;     // S x 36: inverse of the diagonal blocks of L
  // band assembly on the matrix cores (band_mfma_kernel): points sorted by the first shot of their track (the anchor)
  const int *bp_pts;            // P: point ids in that order
  const int *bp_o0, *bp_last;   // P: first observation (point-major position) and last shot offset (from the first shot) of the sorted points
  const unsigned char *bp_pos;  // P x 16: which observation of the sorted point sits at shot offset 0 .. 15 (255: the track misses that shot)
  const int *bp_off;            // S + 1: first sorted position of every anchor
  int bpMode;                   // measurement knob (OSFM_BA_BM_MODE): 1 = no products, 2 = every load from row 0
  int bpNT, bpR, bpPC;          // 16-row tiles of an anchor's matrix, parts per anchor, points per staged chunk
  double *bp_part;              // (S * bpR) x (bpNT (bpNT + 1) / 2) x 256: every workgroup's accumulator tiles
  // cluster block-tridiagonal form of the same band (cs >= bw shots per cluster, dense ncd x ncd blocks)
  int cs, ncl, ncd;
  double *cD;               // ncl x ncd^2: diagonal blocks of the factor (lower triangular)
  double *cW, *cWt;         // L_{c,c-1} and its transpose
  double *cLi, *cLit;       // inverse of the diagonal Cholesky factors and its transpose
  // block cyclic reduction of the cluster-tridiagonal system (parallel replacement of the chain)
  double *bD, *bE, *bG, *bH;  // ncl x ncd^2 each: D / Dinv, coupling to the left neighbour, Dinv*E, Dinv*E_right^T
  double *bD2, *bGt, *bHt;    // the right neighbours' share of D (D = bD + bD2 until the cluster is eliminated); G^T, H^T
  double *bx;                 // ncl x ncd work vector
  // wide band (direct block LDL^T, kWB x kWB tiles): wNB block columns, window of wWb blocks below the diagonal
  int wNB, wWb;
  double *wA, *wL, *wLt, *wDinv, *wx;  // tiles A_{J+dI,J} (updated in place), L row-major / transposed, D^-1 blocks, right-hand sides
  // wide band, cyclic reduction over DENSE clusters of qcs >= bw shots (qm = 6 qcs unknowns, column-major qm x qm blocks in HBM)
  int qcs, qm, qN;
  double *qD;        // qN x qm^2: diagonal blocks (after a cluster's elimination: its inverse)
  double *qE[2];     // qN x qm^2 each: coupling of a cluster to its current left neighbour, ping-pong over the levels
  double *qX;        // qN x (qm x 2 qm): [G = D^-1 E | H = D^-1 E_right^T] of the eliminated clusters
  double *qXt;       // qN x (qm x 2 qm): [G^T | H^T]
  double *qx, *qy;   // right-hand sides (down sweep, in place) and results (up sweep), (qN qm) x NR, NR side by side
  int qT;            // panel width of the blocked inversion (<= 96, a multiple of 6)
  double *qP, *qR, *qRn, *qC;  // per cluster of a level: pivot block inverse (qT^2), row panel, P x row panel (qT x qm), column panel (qm x qT)
  double *qP2, *qBn;           // look-ahead: the next panel's pivot inverse (the two alternate) and its pivot block before the trailing product
  double *zc;       // nred (unscaled J^T w)
  double *y;        // nred
  // pcg
  double *x, *r, *z, *p, *Ap, *b;
  double *scal;     // device scalars
  double *partial;  // block partial sums
  double *partial2; // ... of candidate_points_kernel (it runs in front of candidate_kernel, which still reads the back-substitution's shares in `partial`)
  double *dotp;     // the mat-vec's shares of p . Ap, one per workgroup of the finish kernel
  double *rrp;      // pcg_step1_kernel's shares of r . r, one per workgroup (their own buffer: the straight-line iteration reads them after the
                    // back-substitution and the evaluation have reused `partial`)
  int gen;          // 1: generic mode, the fields of g are set
  GenDev g;
};

#define JA(o, c) d.Jpm[(long)(c) * d.M + (o)]  /* point-major SoA */
#define JL(o, c) ld_stream(&d.Jpm[(long)(c) * d.M + (o)])  /* ... a row's component read once by a streaming pass */
// Components of a point-major row (round 6: 17; rounds 1-5 stored 26 -- 208 bytes per observation through a write path that sustains ~2.5 TB/s
// here).  The translation columns of Jc are -Jp; the six entries of Jk follow from (u, v), the robust weight, sigma and the camera (row_jk:
// project_obs's own expressions): neither is stored.  res 2 | Jp 2 x 3 | Jr 2 x 3 (the ROTATION columns of Jc) | u | v | wt
constexpr int kRowComps = 17;
constexpr int R_JP = 2, R_JR = 8, R_U = 14, R_V = 15, R_WT = 16;

// Jk of the observation at point-major position o (of shot s), times the robust weight, from what its row keeps
__device__ __forceinline__ void row_jk(const Dev &d, long o, int s, double (&jk)[6]) {
  const int ci = d.shot_camera[s];
  const double u = JL(o, R_U), v = JL(o, R_V), wt = JL(o, R_WT), sg = d.o_sigma[o];
  const double *cam = d.cams + 3 * ci;
  const double k1 = cam[0], k2 = cam[1], f = cam[2];
  const bool other = d.cam_model && d.cam_model[ci] >= 2;  // a constant camera of another model: no [k1 k2 focal] columns
  double J[6];
  jk_entries(1.0 / sg, k1, k2, f, u, v, J);
#pragma unroll
  for (int i = 0; i < 6; i++) jk[i] = other ? 0.0 : wt * J[i];
}

__global__ void shot_rot_kernel(Dev d, const double *poses) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= d.S) return;
  rot_and_derivs(poses + 6 * s, d.shotR + 36 * (long)s, d.shotR + 36 * (long)s + 9);
}

// The frame of one shot as the per-shot kernels need it (wave-uniform: the compiler keeps it in scalar registers)
struct ShotFrame {
  int model;
  const double *cam, *pose, *R;
};
__device__ __forceinline__ ShotFrame shot_frame(const Dev &d, int s) {
  const int ci = d.shot_camera[s];
  ShotFrame f;
  f.model = d.cam_model ? d.cam_model[ci] : 0;
  f.cam = f.model >= 2 ? d.cam_ext + 16 * ci : d.cams + 3 * ci;
  f.pose = d.poses + 6 * (long)s;
  f.R = d.shotR + 36 * (long)s;
  return f;
}
// The row of shot-major position k (an observation of the shot of frame f) exactly as eval_kernel stores it in the point-major copy -- corrected
// residual, Jp, Jc, Jk, all times the robust weight wt -- recomputed at the CURRENT parameters (the linearisation point: d.cams / d.poses /
// d.pts, shotR).  Same device function, same operands, same operations as the stored copy: the same bits.  ~400 fp64 operations per
// observation against 208 bytes read (and, once per linearisation, written): at configs[4] 2 GFLOP per pass, 0.05 ms of the vector units.
__device__ __forceinline__ void sm_row(const Dev &d, const ShotFrame &f, long k, double wt, double (&r)[2], double (&Jp)[6], double (&Jc)[12], double (&Jk)[6]) {
  const int p = d.sm_point[k];
  project_obs<true>(f.model, d.pts + 3 * (long)p, f.pose, f.R, f.R + 9, f.cam, d.sm_x[k], d.sm_y[k], 1.0 / d.sm_sigma[k], r, Jp, Jc, Jk);
  r[0] *= wt;
  r[1] *= wt;
#pragma unroll
  for (int i = 0; i < 6; i++) Jp[i] *= wt;
#pragma unroll
  for (int i = 0; i < 12; i++) Jc[i] *= wt;
#pragma unroll
  for (int i = 0; i < 6; i++) Jk[i] *= wt;
}

// residuals, robust corrector, cost partials -- and with JAC the Jacobian rows (the point-major SoA copy; the E blocks when the per-shot band
// assembly wants them) and the points' gradient and J^T J blocks.  One workgroup owns a run of whole tracks with at most kCoopObs
// observations (the partition of the mat-vec, wg_pt): thread per observation evaluates and leaves the nine products of its point block
// in LDS, thread per point adds its track's in observation order -- what point_grad_kernel did from a second read of the rows in rounds
// 3-5 (0.10 ms at configs[4]).  Both variants sum the cost in this partition: a candidate's cost and the cost of the same point once it
// is accepted are the same bits.  A single track longer than the tile is walked by the whole workgroup, its point block added by one
// thread from the rows in memory (fixed order).
template <bool JAC>
__global__ void __launch_bounds__(kCoopObs) eval_kernel(Dev d, const double *cams, const double *poses, const double *pts, int loss, double a) {
  __shared__ double c[JAC ? 9 : 1][kCoopObs];
  __shared__ double lds[8];
  const int tid = threadIdx.x;
  const int p0 = d.wg_pt[blockIdx.x], p1 = d.wg_pt[blockIdx.x + 1];
  const long o0 = d.pt_off[p0], o1 = d.pt_off[p1];
  double acc[2] = {0.0, 0.0};
  // row: the kRowComps components of observation o as they are stored (res 2 | Jp 6 | Jr 6 | u | v | wt; residual and Jacobian entries times the
  // robust weight); the cost terms go to acc
  auto compute = [&](long o, double (&row)[kRowComps]) {
    const int s = d.o_shot[o], p = d.o_point[o];
    const double *R = d.shotR + 36 * (long)s;
    double r[2], Jp[6], Jc[12], Jk[6], uv[2];
    const double sg = d.o_sigma[o];
    const int cmodel = d.cam_model ? d.cam_model[d.shot_camera[s]] : 0;
    project_obs<JAC>(cmodel, pts + 3 * (long)p, poses + 6 * (long)s, R, R + 9, cmodel >= 2 ? d.cam_ext + 16 * d.shot_camera[s] : cams + 3 * d.shot_camera[s], d.o_x[o],
                     d.o_y[o], 1.0 / sg, r, Jp, Jc, Jk, uv);
    const double sq = r[0] * r[0] + r[1] * r[1];
    double rho, rho1;
    loss_eval(loss, a, sq, rho, rho1);
    acc[0] += 0.5 * rho;
    acc[1] += sq * sg * sg;
    if (JAC) {
      const double wt = sqrt(rho1);
      row[0] = wt * r[0];
      row[1] = wt * r[1];
#pragma unroll
      for (int i = 0; i < 6; i++) row[R_JP + i] = wt * Jp[i];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) row[R_JR + 3 * i + k] = wt * Jc[6 * i + k];
      row[R_U] = uv[0];
      row[R_V] = uv[1];
      row[R_WT] = wt;
    }
  };
  // the point block's products of a row (point_grad_kernel's expressions of rounds 3-5: the same sums, the same bits)
  auto products = [&](const double (&row)[kRowComps], double (&v)[9]) {
    const double r0 = row[0], r1 = row[1];
    const double *jp = row + R_JP;
#pragma unroll
    for (int j = 0; j < 3; j++) v[j] = jp[j] * r0 + jp[3 + j] * r1;
    v[3] = jp[0] * jp[0] + jp[3] * jp[3];
    v[4] = jp[0] * jp[1] + jp[3] * jp[4];
    v[5] = jp[0] * jp[2] + jp[3] * jp[5];
    v[6] = jp[1] * jp[1] + jp[4] * jp[4];
    v[7] = jp[1] * jp[2] + jp[4] * jp[5];
    v[8] = jp[2] * jp[2] + jp[5] * jp[5];
  };
  auto store = [&](long o, const double (&row)[kRowComps]) {
    if (d.bpMode == 3) return;  // measurement knob (OSFM_BA_BM_MODE=3): the evaluation without its row stores (results are garbage)
#pragma unroll
    for (int i = 0; i < kRowComps; i++) st_stream<1>(&JA(o, i), row[i]);
    if (d.Epm) {  // E_o = Jc_o^T Jp_o (6 x 3) of the corrected blocks, 144 contiguous bytes per observation: the per-shot band assembly's operand
      const double *jp = row + R_JP;
      double2 *dst = reinterpret_cast<double2 *>(d.Epm + 18 * o);
#pragma unroll
      for (int i = 0; i < 6; i += 2) {  // Jc[r][i] = Jr[r][i] for the rotation (i < 3), -Jp[r][i - 3] for the translation
        const double a0 = i < 3 ? row[R_JR + i] : -jp[i - 3], b0 = i < 3 ? row[R_JR + 3 + i] : -jp[3 + i - 3];
        const double a1 = i + 1 < 3 ? row[R_JR + i + 1] : -jp[i + 1 - 3], b1 = i + 1 < 3 ? row[R_JR + 3 + i + 1] : -jp[3 + i + 1 - 3];
        double e[6];
#pragma unroll
        for (int j = 0; j < 3; j++) {
          e[j] = a0 * jp[j] + b0 * jp[3 + j];
          e[3 + j] = a1 * jp[j] + b1 * jp[3 + j];
        }
        st_stream2<2>(reinterpret_cast<double *>(dst + 3 * (i / 2)), e[0], e[1]);
        st_stream2<2>(reinterpret_cast<double *>(dst + 3 * (i / 2) + 1), e[2], e[3]);
        st_stream2<2>(reinterpret_cast<double *>(dst + 3 * (i / 2) + 2), e[4], e[5]);
      }
    }
  };
  if (o1 - o0 <= kCoopObs) {
    const bool on = tid < o1 - o0;
    double row[kRowComps];
    if (on) {
      compute(o0 + tid, row);
      if (JAC) {
        double v[9];
        products(row, v);
#pragma unroll
        for (int q = 0; q < 9; q++) c[q][tid] = v[q];
      }
    }
    // the rows are stored LAST, behind the exchange through LDS and the cost's block sum: a barrier behind 26 (+ 9) stores in flight makes every
    // wavefront wait for their acknowledgement (the first version of this kernel: 0.52 ms at configs[4], as much as the two kernels it merged)
    block_sum<2>(acc, lds);  // (four wavefronts: its barrier also publishes the products)
    if (tid == 0) {
      d.partial[2 * blockIdx.x] = acc[0];
      d.partial[2 * blockIdx.x + 1] = acc[1];
    }
    if (JAC) {
      const int p = p0 + tid;
      if (p < p1) {
        double s9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (!(d.point_fixed && d.point_fixed[p]))
          for (int k = (int)(d.pt_off[p] - o0); k < (int)(d.pt_off[p + 1] - o0); k++)
#pragma unroll
            for (int q = 0; q < 9; q++) s9[q] += c[q][k];
#pragma unroll
        for (int j = 0; j < 3; j++) d.g_pt[3 * (long)p + j] = s9[j];
#pragma unroll
        for (int j = 0; j < 6; j++) d.Hpp[6 * (long)p + j] = s9[3 + j];
      }
      if (on) store(o0 + tid, row);
    }
    return;
  }
  {  // one track longer than the tile
    for (long o = o0 + tid; o < o1; o += kCoopObs) {
      double row[kRowComps];
      compute(o, row);
      if (JAC) store(o, row);
    }
    if (JAC) {
      __syncthreads();  // the track's rows are in memory
      if (tid == 0) {
        double s9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (!(d.point_fixed && d.point_fixed[p0]))
          for (long o = o0; o < o1; o++) {
            double row[kRowComps], v[9];
#pragma unroll
            for (int i = 0; i < R_JP + 6; i++) row[i] = JA(o, i);
            products(row, v);
#pragma unroll
            for (int q = 0; q < 9; q++) s9[q] += v[q];
          }
#pragma unroll
        for (int j = 0; j < 3; j++) d.g_pt[3 * (long)p0 + j] = s9[j];
#pragma unroll
        for (int j = 0; j < 6; j++) d.Hpp[6 * (long)p0 + j] = s9[3 + j];
      }
    }
  }
  __syncthreads();
  block_sum<2>(acc, lds);
  if (tid == 0) {
    d.partial[2 * blockIdx.x] = acc[0];
    d.partial[2 * blockIdx.x + 1] = acc[1];
  }
}

// out[c] = sum_i partial[i * ncomp + c]   (single block, deterministic)
__global__ void finish_reduce_kernel(const double *partial, long n, int ncomp, double *out) {
  __shared__ double lds[64];
  for (int c = 0; c < ncomp; c++) {
    double v[1] = {0.0};
    for (long i = threadIdx.x; i < n; i += blockDim.x) v[0] += partial[i * ncomp + c];
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) out[c] = v[0];
    __syncthreads();
  }
}

// cost of the prior residuals (camera intrinsics, shot position), added to out[0]
// up-vector residual of shot s from the rotation block shotR (R = world->camera, dR[k] = dR/da_k, a = -rot):
// r = (R^T u - e_z)/sd, J[i][k] = d r_i / d rot_k = -(dR_k^T u)_i / sd
__device__ __forceinline__ void up_residual(const Dev &d, int s, double r[3], double J[9]) {
  const double *R = d.shotR + 36 * (long)s, *u = d.up + 3 * (long)s;
  const double isd = 1.0 / d.up_sigma[s];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double z = R[i] * u[0] + R[3 + i] * u[1] + R[6 + i] * u[2];
    r[i] = isd * (z - (i == 2 ? 1.0 : 0.0));
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double *dR = R + 9 + 9 * k;
      J[3 * i + k] = -isd * (dR[i] * u[0] + dR[3 + i] * u[1] + dR[6 + i] * u[2]);
    }
  }
}

// (round 6: one launch with the reduction of the evaluation's cost partials in front -- out[0 .. 1] = their sums in block order, as
// finish_reduce_kernel forms them -- and, behind it, the clearing of absmax_kernel's slot: three launches of a linearisation less)
__global__ void __launch_bounds__(1024) prior_cost_kernel(Dev d, const double *cams, const double *poses, const double *partial, long npart, double *out, double *absmax_slot, int jac) {
  __shared__ double lds[64];
  for (int c = 0; c < 2; c++) {
    double v[1] = {0.0};
    for (long i = threadIdx.x; i < npart; i += blockDim.x) v[0] += partial[i * 2 + c];
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) out[c] = v[0];
    __syncthreads();
  }
  if (threadIdx.x == 0 && absmax_slot) *absmax_slot = 0.0;
  double v[1] = {0.0};
  for (int c = threadIdx.x; c < d.NC; c += blockDim.x) {
    if (d.cam_fixed[c]) continue;
    const double *q = cams + 3 * c, *pr = d.cam_prior + 3 * c, *sg = d.cam_sigma + 3 * c;
    const double e0 = (q[0] - pr[0]) / fmax(sg[0], kEps), e1 = (q[1] - pr[1]) / fmax(sg[1], kEps);
    const double e2 = log(q[2] / pr[2]) / fmax(sg[2], kEps);
    v[0] += 0.5 * (e0 * e0 + e1 * e1 + e2 * e2);
  }
  if (d.gps && d.gps_sigma)
    for (int s = threadIdx.x; s < d.S; s += blockDim.x) {  // every load of a shot requested before the test on its sigma
      const double sg = d.gps_sigma[s];
      const bool fixed = d.shot_fixed && d.shot_fixed[s];
      const double p3[3] = {poses[6 * s + 3], poses[6 * s + 4], poses[6 * s + 5]}, g3[3] = {d.gps[3 * s], d.gps[3 * s + 1], d.gps[3 * s + 2]};
      if (!(sg > 0) || fixed) continue;
      for (int i = 0; i < 3; i++) {
        const double e = (p3[i] - g3[i]) / sg;
        v[0] += 0.5 * e * e;
      }
    }
  if (d.up && d.up_sigma)
    for (int s = threadIdx.x; s < d.S; s += blockDim.x) {
      if (!(d.up_sigma[s] > 0) || (d.shot_fixed && d.shot_fixed[s])) continue;
      double r[3], J[9], rho, rho1;
      up_residual(d, s, r, J);
      loss_eval(OSFM_LOSS_CAUCHY, 1.0, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho, rho1);
      v[0] += 0.5 * rho;
      if (jac) {
        const double wt = sqrt(rho1);
        for (int i = 0; i < 3; i++) d.up_r[3 * s + i] = wt * r[i];
        for (int i = 0; i < 9; i++) d.up_J[9 * s + i] = wt * J[i];
      }
    }
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) out[0] += v[0];
}

// per shot (one wavefront): gradient, J^T J block, partials of the camera block.  The shot's rows are recomputed (sm_row's operations with
// the robust weight formed here, as eval_kernel forms it) and the weight of every observation is left in sm_wt for the other per-shot kernels
// of this linearisation: the launch stands in for rounds 1-5's second evaluation launch (which wrote the shot-major copy of the rows, 0.34 ms at
// configs[4]) AND their gradient kernel (which read it back, 0.16 ms).
template <int W>
__global__ void __launch_bounds__(64 * W) shot_grad_kernel(Dev d, const double *poses, int loss, double la) {
  __shared__ double red[W > 1 ? W * 36 : 1];
  const int s = (int)xcd_contiguous(blockIdx.x, gridDim.x), lane = threadIdx.x;
  const ShotFrame f = shot_frame(d, s);
  double v[36];  // g(6) H(21) gk(3) Hk(6)
#pragma unroll
  for (int i = 0; i < 36; i++) v[i] = 0;
  for (long k = d.shot_off[s] + lane; k < d.shot_off[s + 1]; k += 64 * W) {
    double r[2], Jp[6], Jc[12], Jk[6];
    const int p = d.sm_point[k];
    project_obs<true>(f.model, d.pts + 3 * (long)p, f.pose, f.R, f.R + 9, f.cam, d.sm_x[k], d.sm_y[k], 1.0 / d.sm_sigma[k], r, Jp, Jc, Jk);
    double rho, rho1;
    loss_eval(loss, la, r[0] * r[0] + r[1] * r[1], rho, rho1);
    const double wt = sqrt(rho1);
    d.sm_wt[k] = wt;
    const double r0 = wt * r[0], r1 = wt * r[1];
    double a[6], b[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      a[j] = wt * Jc[j];
      b[j] = wt * Jc[6 + j];
      v[j] += a[j] * r0 + b[j] * r1;
    }
    int q = 6;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) v[q++] += a[i] * a[j] + b[i] * b[j];
    double ka[3], kb[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      ka[j] = wt * Jk[j];
      kb[j] = wt * Jk[3 + j];
      v[27 + j] += ka[j] * r0 + kb[j] * r1;
    }
    v[30] += ka[0] * ka[0] + kb[0] * kb[0];
    v[31] += ka[1] * ka[0] + kb[1] * kb[0];
    v[32] += ka[1] * ka[1] + kb[1] * kb[1];
    v[33] += ka[2] * ka[0] + kb[2] * kb[0];
    v[34] += ka[2] * ka[1] + kb[2] * kb[1];
    v[35] += ka[2] * ka[2] + kb[2] * kb[2];
  }
  shot_sum<36, W>(v, red);
  if (lane == 0) {
    const bool fixed = d.shot_fixed && d.shot_fixed[s];
    double pd[6] = {0, 0, 0, 0, 0, 0};
    if (!fixed && d.gps && d.gps_sigma && d.gps_sigma[s] > 0) {
      const double wgt = 1.0 / d.gps_sigma[s];
      for (int k = 0; k < 3; k++) {
        v[3 + k] += wgt * wgt * (poses[6 * s + 3 + k] - d.gps[3 * s + k]);
        pd[3 + k] = wgt * wgt;
      }
    }
    if (d.prior_rot) {
      double pr[6] = {0, 0, 0, 0, 0, 0};
      if (!fixed && d.up && d.up_sigma && d.up_sigma[s] > 0) {
        const double *ru = d.up_r + 3 * s, *Ju = d.up_J + 9 * s;
        int q = 0;
        for (int i = 0; i < 3; i++) {
          v[i] += Ju[i] * ru[0] + Ju[3 + i] * ru[1] + Ju[6 + i] * ru[2];
          for (int j = 0; j <= i; j++) pr[q++] = Ju[i] * Ju[j] + Ju[3 + i] * Ju[3 + j] + Ju[6 + i] * Ju[6 + j];
        }
      }
      for (int q = 0; q < 6; q++) {
        d.prior_rot[6 * (long)s + q] = pr[q];
        v[6 + q] += pr[q];  // rotation block of H: the preconditioners and the LM diagonal see the prior through Hcc
      }
    }
    const int dg[6] = {0, 2, 5, 9, 14, 20};
    for (int i = 0; i < 6; i++) {
      d.g_red[6 * s + i] = fixed ? 0.0 : v[i];
      d.prior_diag[6 * s + i] = pd[i];
      d.diag_red[6 * s + i] = fixed ? 0.0 : v[6 + dg[i]] + pd[i];
    }
    for (int i = 0; i < 21; i++) d.Hcc[21 * (long)s + i] = v[6 + i];
    for (int i = 0; i < 9; i++) d.part[9 * (long)s + i] = v[27 + i];
  }
}


// camera blocks: per-shot partials v (g 3 | H 6, reduced over the camera's shots) + prior
__device__ __forceinline__ void cam_grad_one(const Dev &d, const double *cams, int c, const double (&v)[9]) {
  const bool fixed = d.cam_fixed[c];
  const double *q = cams + 3 * c, *pr = d.cam_prior + 3 * c, *sg = d.cam_sigma + 3 * c;
  const double w0 = 1.0 / fmax(sg[0], kEps), w1 = 1.0 / fmax(sg[1], kEps), w2 = 1.0 / fmax(sg[2], kEps);
  const double e[3] = {(q[0] - pr[0]) * w0, (q[1] - pr[1]) * w1, log(q[2] / pr[2]) * w2};
  const double j[3] = {w0, w1, w2 / q[2]};
  const int dg[3] = {3, 5, 8};
  for (int k = 0; k < 3; k++) {
    const int i = d.cam0 + 3 * c + k;
    d.g_red[i] = fixed ? 0.0 : v[k] + j[k] * e[k];
    d.prior_diag[i] = fixed ? 0.0 : j[k] * j[k];
    d.diag_red[i] = fixed ? 0.0 : v[dg[k]] + j[k] * j[k];
  }
  for (int i = 0; i < 6; i++) d.Hcc[21 * (long)d.S + 6 * c + i] = v[3 + i];
}

// camred[c][i] = sum over the shots of camera c of part[s][i]   (one block per camera, fixed order)
// (one workgroup per camera: 1 024 threads, four shots of a thread requested before they are added -- with 256 threads and a load behind
// a branch per shot this was 20 dependent round trips, 27 us, four to five times per LM iteration)
constexpr int kCamRedT = 1024;
__global__ void __launch_bounds__(kCamRedT) cam_reduce_kernel(Dev d, int ncomp, const double *grad_cams) {
  __shared__ double lds[16 * 9];
  const int c = blockIdx.x;
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int s0 = threadIdx.x; s0 < d.S; s0 += 4 * kCamRedT) {
    double p[4][9];
    bool on[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int s = s0 + u * kCamRedT;
      const bool in = s < d.S;
      on[u] = in && d.shot_camera[in ? s : 0] == c;
#pragma unroll
      for (int i = 0; i < 9; i++) p[u][i] = (in && i < ncomp) ? d.part[9 * (long)s + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (on[u])
#pragma unroll
        for (int i = 0; i < 9; i++) v[i] += p[u][i];
  }
  block_sum<9>(v, lds);
  if (threadIdx.x == 0) {
    for (int i = 0; i < ncomp; i++) d.camred[9 * c + i] = v[i];
    if (grad_cams) cam_grad_one(d, grad_cams, c, v);  // (ncomp = 9: the gradient kernel's sums; one launch less per linearisation)
  }
}

// every camera constant (camred: zeros since setup): the camera rows of the gradient from the priors alone
__global__ void cam_grad_kernel(Dev d, const double *cams) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.NC) return;
  double v[9];
  for (int i = 0; i < 9; i++) v[i] = d.camred[9 * c + i];
  cam_grad_one(d, cams, c, v);
}

__global__ void scale_init_kernel(Dev d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.nred) {
    bool fixed;
    if (i < d.cam0)
      fixed = d.shot_fixed && d.shot_fixed[i / 6];
    else
      fixed = d.gen ? false : (bool)d.cam_fixed[(i - d.cam0) / 3];
    d.sc_red[i] = fixed ? 0.0 : 1.0 / (1.0 + sqrt(d.diag_red[i]));
  }
  if (i < 3L * d.P) {
    const long p = i / 3;
    const int j = (int)(i - 3 * p);
    const int dg[3] = {0, 3, 5};
    const bool fixed = d.point_fixed && d.point_fixed[p];
    d.sc_pt[i] = fixed ? 0.0 : 1.0 / (1.0 + sqrt(d.Hpp[6 * p + dg[j]]));
  }
}

// LM diagonal: clamp(diag(J^T J) of the scaled Jacobian) (levenberg_marquardt_strategy.cc)
__global__ void lm_diag_kernel(Dev d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < d.nred) d.D_red[i] = fmin(fmax(d.diag_red[i] * d.sc_red[i] * d.sc_red[i], 1e-6), 1e32);
  if (i < 3L * d.P) {
    const long p = i / 3;
    const int j = (int)(i - 3 * p);
    const int dg[3] = {0, 3, 5};
    d.D_pt[i] = fmin(fmax(d.Hpp[6 * p + dg[j]] * d.sc_pt[i] * d.sc_pt[i], 1e-6), 1e32);
  }
}

// Hhat_p = Dp (Dp Hpp Dp + D/radius)^-1 Dp  (maps unscaled gradients to unscaled point steps)
__global__ void __launch_bounds__(TPB) point_hhat_kernel(Dev d, double radius) {
  const int p = blockIdx.x * TPB + threadIdx.x;
  if (p >= d.P) return;
  const double *Hs = d.Hpp + 6 * (long)p;
  const double s0 = d.sc_pt[3 * (long)p], s1 = d.sc_pt[3 * (long)p + 1], s2 = d.sc_pt[3 * (long)p + 2];
  const double H00 = Hs[0] * s0 * s0 + d.D_pt[3 * (long)p] / radius, H01 = Hs[1] * s0 * s1, H02 = Hs[2] * s0 * s2;
  const double H11 = Hs[3] * s1 * s1 + d.D_pt[3 * (long)p + 1] / radius, H12 = Hs[4] * s1 * s2;
  const double H22 = Hs[5] * s2 * s2 + d.D_pt[3 * (long)p + 2] / radius;
  const double c00 = H11 * H22 - H12 * H12, c01 = H12 * H02 - H01 * H22, c02 = H01 * H12 - H11 * H02;
  const double det = H00 * c00 + H01 * c01 + H02 * c02, id = 1.0 / det;
  double *o = d.Hhat + 6 * (long)p;
  o[0] = c00 * id * s0 * s0;
  o[1] = c01 * id * s0 * s1;
  o[2] = c02 * id * s0 * s2;
  o[3] = (H00 * H22 - H02 * H02) * id * s1 * s1;
  o[4] = (H02 * H01 - H00 * H12) * id * s1 * s2;
  o[5] = (H00 * H11 - H01 * H01) * id * s2 * s2;
}

// block-Jacobi preconditioner: inverse of the diagonal 6x6 (shot) blocks of the scaled Schur complement
__global__ void __launch_bounds__(64) precond_shot_kernel(Dev d, double radius) {
  const int s = blockIdx.x, lane = threadIdx.x;
  double v[27];  // 21 shot + 6 camera partial
#pragma unroll
  for (int i = 0; i < 27; i++) v[i] = 0;
  const ShotFrame f = shot_frame(d, s);
  for (long k = d.shot_off[s] + lane; k < d.shot_off[s + 1]; k += 64) {
    const int p = d.sm_point[k];
    const double *Hh = d.Hhat + 6 * (long)p;
    const double h[9] = {Hh[0], Hh[1], Hh[2], Hh[1], Hh[3], Hh[4], Hh[2], Hh[4], Hh[5]};
    double rr[2], jp[6], jc[12], jk[6];
    sm_row(d, f, k, d.sm_wt[k], rr, jp, jc, jk);
    double E[9][3];  // rows: 6 shot + 3 camera ; E = Jred^T Jp
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const double a = jc[i], b = jc[6 + i];
#pragma unroll
      for (int j = 0; j < 3; j++) E[i][j] = a * jp[j] + b * jp[3 + j];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double a = jk[i], b = jk[3 + i];
#pragma unroll
      for (int j = 0; j < 3; j++) E[6 + i][j] = a * jp[j] + b * jp[3 + j];
    }
    double EH[9][3];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) EH[i][j] = E[i][0] * h[j] + E[i][1] * h[3 + j] + E[i][2] * h[6 + j];
    int q = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) v[q++] += EH[i][0] * E[j][0] + EH[i][1] * E[j][1] + EH[i][2] * E[j][2];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) v[q++] += EH[6 + i][0] * E[6 + j][0] + EH[6 + i][1] * E[6 + j][1] + EH[6 + i][2] * E[6 + j][2];
  }
#pragma unroll
  for (int i = 0; i < 27; i++)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v[i] += __shfl_xor(v[i], m);
  if (lane == 0) {
    for (int i = 0; i < 6; i++) d.part[9 * (long)s + i] = v[21 + i];
    double B[6][6];
    int q = 0;
    for (int i = 0; i < 6; i++)
      for (int j = 0; j <= i; j++, q++) {
        const double si = d.sc_red[6 * s + i], sj = d.sc_red[6 * s + j];
        double val = (d.Hcc[21 * (long)s + q] - v[q]) * si * sj;
        if (i == j) val += d.prior_diag[6 * s + i] * si * si + d.D_red[6 * s + i] / radius;
        B[i][j] = B[j][i] = val;
      }
    // Cholesky B = L L^T, then B^-1 = L^-T L^-1
    double L[6][6];
    bool ok = true;
    for (int i = 0; i < 6; i++)
      for (int j = 0; j <= i; j++) {
        double sum = B[i][j];
        for (int k = 0; k < j; k++) sum -= L[i][k] * L[j][k];
        if (i == j) {
          if (!(sum > 0)) { ok = false; sum = 1.0; }
          L[i][i] = sqrt(sum);
        } else {
          L[i][j] = sum / L[j][j];
        }
      }
    double Li[6][6];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) Li[i][j] = 0;
    for (int c = 0; c < 6; c++) {
      for (int i = c; i < 6; i++) {
        double sum = (i == c) ? 1.0 : 0.0;
        for (int k = c; k < i; k++) sum -= L[i][k] * Li[k][c];
        Li[i][c] = sum / L[i][i];
      }
    }
    double *o = d.Binv + 36 * (long)s;
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) {
        double sum = 0;
        for (int k = (i > j ? i : j); k < 6; k++) sum += Li[k][i] * Li[k][j];
        o[6 * i + j] = ok ? sum : ((i == j) ? 1.0 / fmax(B[i][i], 1e-300) : 0.0);
      }
  }
}

__device__ __forceinline__ void precond_cam_one(const Dev &d, int c, double radius);
__global__ void precond_cam_kernel(Dev d, double radius) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d.NC) return;
  precond_cam_one(d, c, radius);
}
// the 3 x 3 block-Jacobi block of camera c (inverse into Binv's camera part)
__device__ __forceinline__ void precond_cam_one(const Dev &d, int c, double radius) {
  double v[6];
  for (int i = 0; i < 6; i++) v[i] = d.camred[9 * c + i];
  double B[3][3];
  int q = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j <= i; j++, q++) {
      const double si = d.sc_red[d.cam0 + 3 * c + i], sj = d.sc_red[d.cam0 + 3 * c + j];
      double val = (d.Hcc[21 * (long)d.S + 6 * c + q] - v[q]) * si * sj;
      if (i == j) val += d.prior_diag[d.cam0 + 3 * c + i] * si * si + d.D_red[d.cam0 + 3 * c + i] / radius;
      B[i][j] = B[j][i] = val;
    }
  const double c00 = B[1][1] * B[2][2] - B[1][2] * B[1][2], c01 = B[1][2] * B[0][2] - B[0][1] * B[2][2];
  const double c02 = B[0][1] * B[1][2] - B[1][1] * B[0][2];
  const double det = B[0][0] * c00 + B[0][1] * c01 + B[0][2] * c02;
  double *o = d.Binv + 36 * (long)d.S + 9 * c;
  if (!(det > 0)) {
    for (int i = 0; i < 9; i++) o[i] = 0;
    for (int i = 0; i < 3; i++) o[4 * i] = 1.0 / fmax(B[i][i], 1e-300);
    return;
  }
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = o[1]; o[4] = (B[0][0] * B[2][2] - B[0][2] * B[0][2]) * id; o[5] = (B[0][2] * B[0][1] - B[0][0] * B[1][2]) * id;
  o[6] = o[2]; o[7] = o[5]; o[8] = (B[0][0] * B[1][1] - B[0][1] * B[0][1]) * id;
}


// ---- banded preconditioner ---------------------------------------------------------------------
// Sequence-like captures (the street-level case OpenSfM was written for) give a reduced camera
// system whose shot-shot part is block banded: two shots interact only if they share a point.
// M = band_bw(S_shots) (+ the 3x3 camera blocks) is assembled explicitly, factorised by a
// sequential block Cholesky inside ONE wavefront (the dependency chain is S long, each link is a
// few 6x6 products: latency bound, ~1 us per block row), and applied by two banded triangular
// sweeps per CG iteration.  When bw covers every co-visibility, M equals the shot part of the
// Schur complement exactly and CG only has to resolve the rank-3 coupling to the shared camera.
constexpr int kMaxBw = 15;
constexpr int kBandCopies = 8;

// Single-wavefront kernels: LDS operations of one wave are executed in issue order, so a compiler
// barrier is all that is needed between a ds_write and a dependent ds_read of another lane.
// (__syncthreads() would also drain vmcnt and kill the software prefetch of the next band row.)
__device__ __forceinline__ void WAVE_SYNC() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// E_o = Jc_o^T Jp_o of the observation at shot-major position k: the evaluation kernel's E block (the per-shot assembly's operand array, in
// point-major order), found through shot_obs.  (Rounds 1-5 formed it from the shot-major copy of the rows: the same expression, the same bits.)
__device__ __forceinline__ void jred_jp(const Dev &d, long k, double E[6][3]) {
  const double2 *src = reinterpret_cast<const double2 *>(d.Epm + 18 * (long)d.shot_obs[k]);
#pragma unroll
  for (int q = 0; q < 9; q++) {
    const double2 v = src[q];
    E[(2 * q) / 3][(2 * q) % 3] = v.x;
    E[(2 * q + 1) / 3][(2 * q + 1) % 3] = v.y;
  }
}
// Band assembly: one workgroup per shot s builds the blocks (s, s - dk), dk = 0 .. bw:
//   S_(s, s2) -= sum over the points p seen by both of  (E_o Hhat_p) E_o2^T,   o / o2 = the observations of p in s / s2.
// Thread per observation of s, its track's E rows gathered from the point-major array, products added into LDS with fp64 atomics
// (privatised x8; the summation order, hence the last bits of the PRECONDITIONER, vary from run to run -- the mat-vec, gradients and
// cost are atomics-free).  An atomics-free variant was measured in round 2 (partner table per band column, one (dk, row, part) thread
// per six outputs accumulating in registers, deterministic): 2.28 ms with one gather in flight, 2.77 ms with two and 512 threads,
// against 1.76 ms here -- the kernel is bound by the ~4 GB gather of partner blocks (every E row is wanted by the ~10 shots that see
// its point, in ten different orders: the reuse misses the 4 MB L2s), not by the accumulation.
// Half-widths beyond what one workgroup's LDS holds (round 4): the launch is repeated per slice of band columns dk_lo <= dk < dk_lo + ndk.
__global__ void __launch_bounds__(TPB) band_assemble_kernel(Dev d, double radius, int copies, int dk_lo, int ndk) {
  // kBandCopies private copies of the accumulators, copy = lane & (kBandCopies - 1), interleaved so that the copies of one entry sit
  // in different banks: the lanes of a wavefront add into a handful of (dk, i, j) entries at a time, and same-address fp64 LDS
  // atomics serialise
  extern __shared__ __attribute__((aligned(16))) double acc[];  // (bw + 1) * 36 * copies (copies: a power of two, 8 where LDS allows)
  // blocks are dealt round-robin to the 8 XCDs: give every XCD a contiguous range of shots, whose workgroups gather the same E rows
  const int s = (int)xcd_contiguous(blockIdx.x, gridDim.x), nb = ndk * 36;
  const int copy = threadIdx.x & (copies - 1);
  for (int t = threadIdx.x; t < nb * copies; t += TPB) acc[t] = 0.0;
  __syncthreads();
  for (long k = d.shot_off[s] + threadIdx.x; k < d.shot_off[s + 1]; k += TPB) {
    const int p = d.sm_point[k];
    const double *Hh = d.Hhat + 6 * (long)p;
    const double h[9] = {Hh[0], Hh[1], Hh[2], Hh[1], Hh[3], Hh[4], Hh[2], Hh[4], Hh[5]};
    double Ea[6][3], EH[6][3];
    jred_jp(d, k, Ea);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) EH[i][j] = Ea[i][0] * h[j] + Ea[i][1] * h[3 + j] + Ea[i][2] * h[6 + j];
    for (long o2 = d.pt_off[p]; o2 < d.pt_off[p + 1]; o2++) {
      const int dk = s - d.o_shot[o2] - dk_lo;
      if (dk < 0 || dk >= ndk) continue;
      double Eb[6][3];
      {
        const double2 *src = reinterpret_cast<const double2 *>(d.Epm + 18 * o2);
#pragma unroll
        for (int q = 0; q < 9; q++) {
          const double2 v = src[q];
          Eb[(2 * q) / 3][(2 * q) % 3] = v.x;
          Eb[(2 * q + 1) / 3][(2 * q + 1) % 3] = v.y;
        }
      }
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++)
          atomicAdd(&acc[(dk * 36 + i * 6 + j) * copies + copy], EH[i][0] * Eb[j][0] + EH[i][1] * Eb[j][1] + EH[i][2] * Eb[j][2]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nb; t += TPB) {
    const int dk = dk_lo + t / 36, ij = t % 36, i = ij / 6, j = ij % 6;
    const int s2 = s - dk;
    double val = 0.0;
    if (s2 >= 0) {
      double sum = 0.0;
      for (int c = 0; c < copies; c++) sum += acc[t * copies + c];
      val = -sum;
      if (dk == 0) {
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        val += d.Hcc[21 * (long)s + hi * (hi + 1) / 2 + lo];
        if (i == j) val += d.prior_diag[6 * s + i];
      }
      val *= d.sc_red[6 * s + i] * d.sc_red[6 * s2 + j];
      if (dk == 0 && i == j) val += d.D_red[6 * s + i] / radius;
    }
    d.band[((long)s * (d.bw + 1) + dk) * 36 + ij] = val;
  }
}

// Wide bands (round 4).  A shot of a block survey has a band row of 100+ blocks of which ~15 hold anything (its neighbours along and
// across the flight lines); accumulators for all of them leave room for four private copies and one workgroup per CU, and the kernel is
// bound by same-address LDS atomics.  The topology does not change over the LM iterations: band_slots_kernel gives every shot, once at
// setup, a table dk -> slot over the partners it really has; the accumulators then cover the slots only (eight copies, several
// workgroups per CU, and no slicing whatever the half-width).  Same sums as band_assemble_kernel.
__global__ void __launch_bounds__(TPB) band_slots_kernel(Dev d, unsigned char *table, int *max_slots) {
  extern __shared__ int bs_present[];  // bw + 1
  const int s = blockIdx.x, R1 = d.bw + 1;
  for (int t = threadIdx.x; t < R1; t += TPB) bs_present[t] = 0;
  __syncthreads();
  for (long k = d.shot_off[s] + threadIdx.x; k < d.shot_off[s + 1]; k += TPB) {
    const int p = d.sm_point[k];
    for (long o2 = d.pt_off[p]; o2 < d.pt_off[p + 1]; o2++) {
      const int dk = s - d.o_shot[o2];
      if (dk >= 0 && dk <= d.bw) bs_present[dk] = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0;
    for (int t = 0; t < R1; t++) {
      const bool on = bs_present[t] != 0 || t == 0;  // the diagonal block always exists
      table[(long)s * R1 + t] = on ? (unsigned char)(n < 255 ? n : 255) : (unsigned char)255;
      n += on ? 1 : 0;
    }
    atomicMax(max_slots, n);
  }
}
__global__ void __launch_bounds__(TPB) band_assemble_compact_kernel(Dev d, double radius, int copies, int nslots) {
  extern __shared__ __attribute__((aligned(16))) double acc[];  // nslots * 36 * copies doubles, then the shot's slot table (bw + 1 bytes)
  const int s = (int)xcd_contiguous(blockIdx.x, gridDim.x), R1 = d.bw + 1;
  unsigned char *slot = reinterpret_cast<unsigned char *>(acc + (long)nslots * 36 * copies);
  const int copy = threadIdx.x & (copies - 1);
  for (int t = threadIdx.x; t < nslots * 36 * copies; t += TPB) acc[t] = 0.0;
  for (int t = threadIdx.x; t < R1; t += TPB) slot[t] = d.bslot[(long)s * R1 + t];
  __syncthreads();
  for (long k = d.shot_off[s] + threadIdx.x; k < d.shot_off[s + 1]; k += TPB) {
    const int p = d.sm_point[k];
    const double *Hh = d.Hhat + 6 * (long)p;
    const double h[9] = {Hh[0], Hh[1], Hh[2], Hh[1], Hh[3], Hh[4], Hh[2], Hh[4], Hh[5]};
    double Ea[6][3], EH[6][3];
    jred_jp(d, k, Ea);
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) EH[i][j] = Ea[i][0] * h[j] + Ea[i][1] * h[3 + j] + Ea[i][2] * h[6 + j];
    for (long o2 = d.pt_off[p]; o2 < d.pt_off[p + 1]; o2++) {
      const int dk = s - d.o_shot[o2];
      if (dk < 0 || dk > d.bw) continue;
      const int sl = slot[dk];
      double Eb[6][3];
      {
        const double2 *src = reinterpret_cast<const double2 *>(d.Epm + 18 * o2);
#pragma unroll
        for (int q = 0; q < 9; q++) {
          const double2 v = src[q];
          Eb[(2 * q) / 3][(2 * q) % 3] = v.x;
          Eb[(2 * q + 1) / 3][(2 * q + 1) % 3] = v.y;
        }
      }
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++)
          atomicAdd(&acc[(sl * 36 + i * 6 + j) * copies + copy], EH[i][0] * Eb[j][0] + EH[i][1] * Eb[j][1] + EH[i][2] * Eb[j][2]);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < R1 * 36; t += TPB) {
    const int dk = t / 36, ij = t % 36, i = ij / 6, j = ij % 6;
    const int s2 = s - dk, sl = slot[dk];
    double val = 0.0;
    if (s2 >= 0) {
      double sum = 0.0;
      if (sl != 255)
        for (int c = 0; c < copies; c++) sum += acc[(sl * 36 + ij) * copies + c];
      val = -sum;
      if (dk == 0) {
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        val += d.Hcc[21 * (long)s + hi * (hi + 1) / 2 + lo];
        if (i == j) val += d.prior_diag[6 * s + i];
      }
      val *= d.sc_red[6 * s + i] * d.sc_red[6 * s2 + j];
      if (dk == 0 && i == j) val += d.D_red[6 * s + i] / radius;
    }
    d.band[((long)s * R1 + dk) * 36 + ij] = val;
  }
}

// ---- band assembly on the matrix cores (round 4) ---------------------------------------------------------------------------------
// The per-shot kernel above issues 36 fp64 LDS atomics per (observation, co-visible shot): 990 M at configs[4], 1.16 ms at the ~1.4 adds
// per cycle and CU the LDS sustains.  Here nothing is added through memory.  Points are taken in the order of the FIRST shot of their
// track (sorted once at setup); a unit of work = the points with first shot a (split into R parts when there are many).  For such a
// point stack Y = [F_o] (6 rows per shot offset 0 .. bw from a, zero rows for the shots that do not see it; F_o = E_o Hhat_p, 6 x 3) and
// Z = [E_o]: the point's contribution to every block (a + x, a + y) of the band is the 6 x 6 block (x, y) of Y Z^T -- a rank-3 update of
// ONE matrix per anchor a, whatever shots inside the window the track has (no two points need share a shot set).  Y Z^T summed over the
// unit's points is a GEMM with K = 3 per point: v_mfma_f64_16x16x4_f64 (K padded to 4), lower-triangular 16 x 16 tiles of the
// 6 (bw + 1) rows, accumulators in registers for the whole unit.  The four wavefronts of a workgroup share the TILES (every wavefront
// walks every point of a chunk staged in LDS; no reduction between them), operands come from LDS through a byte table
// pos[point][shot offset] -> slot.  band_finish_kernel then adds, per band block (s, s - dk), the (at most bw + 1 - dk) anchors x R
// parts that hold it, in a fixed order: no atomics anywhere, the band -- hence the preconditioner and the whole solve -- is
// reproducible bit for bit.  Needs the exact band (bw = the widest track, at most kMaxBw) and one observation per (track, shot);
// anything else keeps the kernel above.
constexpr int kBmSlots = 128;          // (point, shot offset) slots staged per chunk: E and F, 2 x 18 x 136 doubles = 39 KB, twice (double buffer): two workgroups per CU
constexpr int kBmStride = kBmSlots + 8;  // slot kBmSlots holds zeros: the operand of a lane whose row lies outside the band or whose K index is the pad
typedef double bm_v4d __attribute__((ext_vector_type(4)));

// LDS-only workgroup barrier: __syncthreads() would also wait for the global loads of the NEXT chunk, which are meant to stay in flight
// underneath the products of this one
__device__ __forceinline__ void bm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT>  // 16-row tiles: 6 (bw + 1) <= 16 NT
__global__ void __launch_bounds__(384) band_mfma_kernel(Dev d) {
  constexpr int NTILES = NT * (NT + 1) / 2, MAXT = (NTILES + 3) / 4;
  constexpr int kBuf = 36 * kBmStride;  // doubles of one staging buffer (E then F, 18 SoA rows each)
  extern __shared__ __attribute__((aligned(16))) double bm_lds[];  // two staging buffers: one is filled while the other is multiplied
  const int TL = d.bw + 1, PC = d.bpPC;
  const int a = blockIdx.x / d.bpR, part = blockIdx.x - a * d.bpR;
  const int n_all = d.bp_off[a + 1] - d.bp_off[a];
  const int i0 = d.bp_off[a] + (int)((long)n_all * part / d.bpR), i1 = d.bp_off[a] + (int)((long)n_all * (part + 1) / d.bpR);
  const int nch = (i1 - i0 + PC - 1) / PC;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 72) bm_lds[(threadIdx.x / 36) * kBuf + (threadIdx.x % 36) * kBmStride + kBmSlots] = 0.0;  // the zero slot of both buffers
  // Six wavefronts, two trades.  Wavefronts 4 and 5 STAGE: thread (t, sh) puts the observation of point t of the chunk in the shot a + sh
  // into slot t TL + sh (zeros when the track misses that shot): E = Jc^T Jp from the point-major Jacobian copy, F = E Hhat.  What they
  // need from the sorted tables is fetched two chunks ahead, the Jacobian entries and Hhat one chunk ahead; every load is unconditional
  // (a thread with nothing to stage reads row 0 and discards it: a branch around the loads would make the compiler wait at its end).
  // Wavefronts 0 .. 3 MULTIPLY the chunk staged one step before, each its own tiles.  One LDS barrier per chunk.
  if (wave >= 4) {
    const int lt = threadIdx.x - 256;
    const int ut = lt / TL, ush = lt - ut * TL;
    const bool loader = ut < PC;
    const int u = ut * TL + ush;
    int n_p, n_o0, n_k;
    auto fetch_index = [&](int c) {
      const bool on = loader && c + ut < i1;
      const long r = d.bpMode == 2 ? 0 : on ? c + ut : i0 < i1 ? i0 : 0;
      n_p = d.bp_pts[r];
      n_o0 = d.bp_o0[r];
      const int k = d.bp_pos[16 * r + ush];
      n_k = on ? k : 255;
    };
    double jv[18], hh[6];
    bool c_have = false;
    auto fetch_blocks = [&]() {  // of the chunk whose index is in n_*
      c_have = n_k != 255;
      const long o = n_o0 + (c_have ? n_k : 0);
      // jv: Jp (6) | Jc row 0 (6) | Jc row 1 (6).  The generic rows store all of it (res 2 | Jp 6 | Jc 12); the [k1 k2 focal] rows keep the rotation
      // columns only (R_JR) -- the translation columns are -Jp, read a second time with the other sign (the same addresses: cache hits) so that the
      // loads stay unconditional
      const int rot1 = d.gen ? 14 : R_JR + 3, tr0 = R_JP, tr1 = R_JP + 3;  // (the generic rows do not store their translation columns either: round 6)
      const double sgn = -1.0;
#pragma unroll
      for (int x = 0; x < 6; x++) jv[x] = JA(o, R_JP + x);
#pragma unroll
      for (int x = 0; x < 3; x++) {
        jv[6 + x] = JA(o, R_JR + x);
        jv[12 + x] = JA(o, rot1 + x);
        jv[9 + x] = sgn * JA(o, tr0 + x);
        jv[15 + x] = sgn * JA(o, tr1 + x);
      }
      const double *Hh = d.Hhat + 6L * n_p;
#pragma unroll
      for (int x = 0; x < 6; x++) hh[x] = Hh[x];
    };
    fetch_index(i0);
    fetch_blocks();
    fetch_index(i0 + PC);
    for (int c = 0; c <= nch; c++) {
      if (c < nch && loader) {
        double *Es = bm_lds + (c & 1) * kBuf, *Fs = Es + 18 * kBmStride;
        const double zf = c_have ? 1.0 : 0.0;
        double E[18];  // E[i][j] = Jc[0][i] Jp[0][j] + Jc[1][i] Jp[1][j], the expression of eval_kernel's E block (jred_jp)
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) E[3 * i + j] = jv[6 + i] * jv[j] + jv[12 + i] * jv[3 + j];
        const double h[9] = {hh[0], hh[1], hh[2], hh[1], hh[3], hh[4], hh[2], hh[4], hh[5]};
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            Es[(3 * i + j) * kBmStride + u] = c_have ? E[3 * i + j] : 0.0;
            Fs[(3 * i + j) * kBmStride + u] = zf * (E[3 * i] * h[j] + E[3 * i + 1] * h[3 + j] + E[3 * i + 2] * h[6 + j]);
          }
      }
      fetch_blocks();  // chunk c + 1: its index arrived during the chunk before
      fetch_index(i0 + (c + 2) * PC);
      bm_lds_barrier();
    }
    return;
  }
  const int li = lane & 15, kk = lane >> 4;
  // this wavefront's tiles tau = wave + 4 m (tau enumerates (I, J), J <= I, row by row) and the lane's operand addresses in them
  int adA[MAXT], adB[MAXT];
  bool have[MAXT];
#pragma unroll
  for (int m = 0; m < MAXT; m++) {
    const int tau = wave + 4 * m;
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= tau) I++;
    const int J = tau - I * (I + 1) / 2;
    have[m] = tau < NTILES;
    const int yA = 16 * I + li, yB = 16 * J + li;
    const int sA = yA / 6, sB = yB / 6;
    adA[m] = (have[m] && kk < 3 && sA <= d.bw) ? ((yA - 6 * sA) * 3 + kk) * kBmStride + sA : -1;
    adB[m] = (have[m] && kk < 3 && sB <= d.bw) ? ((yB - 6 * sB) * 3 + kk) * kBmStride + sB : -1;
  }
  bm_v4d acc[MAXT];
#pragma unroll
  for (int m = 0; m < MAXT; m++) acc[m] = (bm_v4d){0.0, 0.0, 0.0, 0.0};
  for (int c = 0; c <= nch; c++) {
    if (c >= 1) {
      const int npt = min(PC, i1 - (i0 + (c - 1) * PC));
      const int bo = ((c - 1) & 1) * kBuf;
      int ia[MAXT], ib[MAXT];  // (indices, not pointers: a pointer into LDS that went through an array becomes a flat access, which waits for vmcnt)
#pragma unroll
      for (int m = 0; m < MAXT; m++) {
        ia[m] = bo + 18 * kBmStride + (adA[m] < 0 ? kBmSlots : adA[m]);
        ib[m] = bo + (adB[m] < 0 ? kBmSlots : adB[m]);
      }
      for (int t = 0; t < (d.bpMode == 1 ? 0 : npt); t++) {
        double va[MAXT], vb[MAXT];
#pragma unroll
        for (int m = 0; m < MAXT; m++) {
          va[m] = bm_lds[ia[m]];
          vb[m] = bm_lds[ib[m]];
          ia[m] += adA[m] < 0 ? 0 : TL;
          ib[m] += adB[m] < 0 ? 0 : TL;
        }
#pragma unroll
        for (int m = 0; m < MAXT; m++) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[m], vb[m], acc[m], 0, 0, 0);
      }
    }
    bm_lds_barrier();
  }
#pragma unroll
  for (int m = 0; m < MAXT; m++)
    if (have[m]) {
      double *dst = d.bp_part + ((long)blockIdx.x * NTILES + wave + 4 * m) * 256 + lane;
#pragma unroll
      for (int r = 0; r < 4; r++) dst[64 * r] = acc[m][r];
    }
}

// block (s, s - dk) of the band: the sums of every anchor a = s - bw .. s - dk that holds both shots (rows 6 (s - a) + i, columns
// 6 (s - dk - a) + j of its matrix; element (row, col) of a 16 x 16 accumulator tile sits in register row / 4 of lane 16 (row % 4) + col),
// added in a fixed order, then what band_assemble_kernel does at its end -- sign, the shot's own J^T J block and priors on the
// diagonal, Jacobi scaling, the LM diagonal
__global__ void __launch_bounds__(TPB) band_finish_kernel(Dev d, double radius) {
  const long t = (long)blockIdx.x * TPB + threadIdx.x;
  const int R1 = d.bw + 1;
  if (t >= (long)d.S * R1 * 36) return;
  const int ij = (int)(t % 36), dk = (int)((t / 36) % R1), s = (int)(t / (36 * R1));
  const int i = ij / 6, j = ij - 6 * i, s2 = s - dk;
  double val = 0.0;
  if (s2 >= 0) {
    double sum = 0.0;
    const int ntiles = d.bpNT * (d.bpNT + 1) / 2;
    for (int a = max(0, s - d.bw); a <= s2; a++) {
      int row = 6 * (s - a) + i, col = 6 * (s2 - a) + j;
      if (col > row) {  // a diagonal block across a tile boundary: only the lower tiles exist, and E Hhat E^T is symmetric
        const int tmp = row;
        row = col;
        col = tmp;
      }
      const int I = row >> 4, J = col >> 4, ri = row & 15, cj = col & 15;
      const long e = ((long)(I * (I + 1) / 2 + J) * 4 + (ri >> 2)) * 64 + ((ri & 3) << 4) + cj;
      for (int r = 0; r < d.bpR; r++) sum += d.bp_part[((long)a * d.bpR + r) * ntiles * 256 + e];
    }
    val = -sum;
    if (dk == 0) {
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      val += d.Hcc[21 * (long)s + hi * (hi + 1) / 2 + lo];
      if (i == j) val += d.prior_diag[6 * s + i];
    }
    val *= d.sc_red[6 * s + i] * d.sc_red[6 * s2 + j];
    if (dk == 0 && i == j) val += d.D_red[6 * s + i] / radius;
  }
  d.band[t] = val;
}

// sequential banded block Cholesky, one wavefront; ring[] keeps the last bw+1 factor rows in LDS
__global__ void __launch_bounds__(64) band_cholesky_kernel(Dev d, int *status) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int R = d.bw + 1;            // blocks per row
  constexpr int NR = kMaxBw + 1;     // ring rows (power of two: slot = row & kMaxBw)
  double *ring = lds;                 // [NR][R*36]
  double *dinvr = ring + NR * R * 36; // [NR][36]
  double *T = dinvr + NR * 36;        // [36]
  double *Lc = T + 36;               // [36] work: diagonal factor
  const int lane = threadIdx.x;
  const int r = lane / 6, c = lane % 6;  // valid for lane < 36
  int bad = 0;
  if (lane == 0) *status = 0;
  // software prefetch of the next band row (the factorisation itself is a latency-bound chain)
  constexpr int kPre = ((kMaxBw + 1) * 36 + 63) / 64;
  double pre[kPre];
#pragma unroll
  for (int u = 0; u < kPre; u++) pre[u] = (lane + 64 * u < R * 36) ? d.band[lane + 64 * u] : 0.0;
  for (int i = 0; i < d.S; i++) {
    double *cur = ring + (i & kMaxBw) * R * 36;
#pragma unroll
    for (int u = 0; u < kPre; u++)
      if (lane + 64 * u < R * 36) cur[lane + 64 * u] = pre[u];
    if (i + 1 < d.S) {
#pragma unroll
      for (int u = 0; u < kPre; u++) pre[u] = (lane + 64 * u < R * 36) ? d.band[(long)(i + 1) * R * 36 + lane + 64 * u] : 0.0;
    }
    WAVE_SYNC();
    for (int kk = d.bw; kk >= 1; kk--) {
      const int j = i - kk;
      if (j < 0) continue;
      const double *rowj = ring + (j & kMaxBw) * R * 36;
      if (lane < 36) {
        const int m0 = (i - d.bw > j - d.bw ? i - d.bw : j - d.bw);
        double a0 = 0, a1 = 0, a2 = 0;  // independent partial sums: the chain was FMA-latency bound
        for (int m = (m0 < 0 ? 0 : m0); m < j; m++) {
          const double *Lim = cur + (i - m) * 36, *Ljm = rowj + (j - m) * 36;
          a0 += Lim[r * 6 + 0] * Ljm[c * 6 + 0] + Lim[r * 6 + 3] * Ljm[c * 6 + 3];
          a1 += Lim[r * 6 + 1] * Ljm[c * 6 + 1] + Lim[r * 6 + 4] * Ljm[c * 6 + 4];
          a2 += Lim[r * 6 + 2] * Ljm[c * 6 + 2] + Lim[r * 6 + 5] * Ljm[c * 6 + 5];
        }
        T[lane] = cur[kk * 36 + lane] - ((a0 + a1) + a2);
      }
      WAVE_SYNC();
      if (lane < 36) {
        const double *dj = dinvr + (j & kMaxBw) * 36;  // L_jj^-1 ; L_ij = T * L_jj^-T
        double v = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) v += T[r * 6 + q] * dj[c * 6 + q];
        cur[kk * 36 + lane] = v;
      }
      WAVE_SYNC();
    }
    // diagonal block
    if (lane < 36) {
      const int m0 = i - d.bw;
      double a0 = 0, a1 = 0, a2 = 0;
      for (int m = (m0 < 0 ? 0 : m0); m < i; m++) {
        const double *Lim = cur + (i - m) * 36;
        a0 += Lim[r * 6 + 0] * Lim[c * 6 + 0] + Lim[r * 6 + 3] * Lim[c * 6 + 3];
        a1 += Lim[r * 6 + 1] * Lim[c * 6 + 1] + Lim[r * 6 + 4] * Lim[c * 6 + 4];
        a2 += Lim[r * 6 + 2] * Lim[c * 6 + 2] + Lim[r * 6 + 5] * Lim[c * 6 + 5];
      }
      T[lane] = cur[lane] - ((a0 + a1) + a2);
      Lc[lane] = 0.0;
    }
    WAVE_SYNC();
    // 6x6 Cholesky, lane rr owns row rr
    for (int cc = 0; cc < 6; cc++) {
      if (lane < 6 && lane >= cc) {
        double sum = T[lane * 6 + cc];
        for (int q = 0; q < cc; q++) sum -= Lc[lane * 6 + q] * Lc[cc * 6 + q];
        if (lane == cc) {
          if (!(sum > 0)) { bad = 1; sum = 1.0; }
          Lc[cc * 6 + cc] = sqrt(sum);
        } else {
          T[lane * 6 + cc] = sum;  // numerator, divided once the pivot is known
        }
      }
      WAVE_SYNC();
      if (lane < 6 && lane > cc) Lc[lane * 6 + cc] = T[lane * 6 + cc] / Lc[cc * 6 + cc];
      WAVE_SYNC();
    }
    // inverse of the lower-triangular factor: lane cc computes column cc
    double *di = dinvr + (i & kMaxBw) * 36;
    if (lane < 6) {
      const int cc = lane;
      double col[6];
      for (int rr = 0; rr < 6; rr++) {
        double sum = (rr == cc) ? 1.0 : 0.0;
        for (int q = cc; q < rr; q++) sum -= Lc[rr * 6 + q] * col[q];
        col[rr] = (rr >= cc) ? sum / Lc[rr * 6 + rr] : 0.0;
      }
      for (int rr = 0; rr < 6; rr++) di[rr * 6 + cc] = col[rr];
    }
    if (lane < 36) cur[lane] = Lc[lane];
    WAVE_SYNC();
    for (int t = lane; t < R * 36; t += 64) d.band[(long)i * R * 36 + t] = cur[t];
    if (lane < 36) d.dinv[(long)i * 36 + lane] = di[lane];
    WAVE_SYNC();
  }
  if (bad) *status = 1;
}

// z_shots = (L L^T)^-1 r_shots   (one wavefront, two banded sweeps); camera rows: 3x3 block Jacobi.
// The sweeps are dependency chains S links long; what can be taken off the chain is: operands are
// fetched kPD rows ahead into a register pipeline (an L2/MALL miss is ~1 us, a link ~0.3 us), and
// the band product is accumulated in three independent partial sums.
constexpr int kPD = 4;
// BW (block half-bandwidth) is a template parameter: the per-row instruction count IS the run time
// of this kernel (one wavefront issues ~1 instruction per 4 cycles), so no masks, no runtime loops.
template <int BW>
__global__ void __launch_bounds__(64) band_solve_kernel(Dev d, const double *rin, double *z) {
  __shared__ double part[36];
  __shared__ double tv[6];
  __shared__ double yring[(kMaxBw + 1) * 6];
  constexpr int R = BW + 1;
  constexpr long kRow = (long)R * 36;
  const int lane = threadIdx.x, r = lane / 6, c = lane % 6;
  const int l36 = lane < 36 ? lane : 0, l6 = lane < 6 ? lane : 0;
  const int lt = lane < 36 ? c * 6 + r : 0;
  double pb[kPD][BW], pdv[kPD][6], pr[kPD];
  for (int t = lane; t < (kMaxBw + 1) * 6; t += 64) yring[t] = 0.0;  // rows < 0 / >= S contribute zero
  WAVE_SYNC();
  // ---------------- forward: y_i = Linv_ii (r_i - sum_k L_{i,i-k} y_{i-k}) ----------------
  const double *bp = d.band + 36 + l36;    // block k = 1 of row 0, this lane's element
  const double *dp = d.dinv + l6 * 6;
#pragma unroll
  for (int u = 0; u < kPD; u++) {
    const long row = u < d.S ? u : d.S - 1;
#pragma unroll
    for (int k = 0; k < BW; k++) pb[u][k] = bp[row * kRow + k * 36];
#pragma unroll
    for (int q = 0; q < 6; q++) pdv[u][q] = dp[row * 36 + q];
    pr[u] = rin[6 * row + l6];
  }
  for (int i0 = 0; i0 < d.S; i0 += kPD) {
#pragma unroll
    for (int u = 0; u < kPD; u++) {  // slot u always serves rows == u (mod kPD): no register rotation
      const int i = i0 + u;
      if (i < d.S) {
        double cb[BW], cd[6];
        const double cr = pr[u];
#pragma unroll
        for (int k = 0; k < BW; k++) cb[k] = pb[u][k];
#pragma unroll
        for (int q = 0; q < 6; q++) cd[q] = pdv[u][q];
        {
          const long row = i + kPD < d.S ? i + kPD : d.S - 1;
#pragma unroll
          for (int k = 0; k < BW; k++) pb[u][k] = bp[row * kRow + k * 36];
#pragma unroll
          for (int q = 0; q < 6; q++) pdv[u][q] = dp[row * 36 + q];
          pr[u] = rin[6 * row + l6];
        }
        {
          // band rows of the first BW shots hold zero blocks for columns < 0 (band_assemble), and
          // yring starts zeroed, so no boundary predicate is needed
          double a[3] = {0, 0, 0};
#pragma unroll
          for (int k = 1; k <= BW; k++) a[(k - 1) % 3] += cb[k - 1] * yring[((i - k) & kMaxBw) * 6 + c];
          if (lane < 36) part[lane] = (a[0] + a[1]) + a[2];
        }
        WAVE_SYNC();
        if (lane < 6) {
          const double t = cr - (((part[lane * 6] + part[lane * 6 + 1]) + (part[lane * 6 + 2] + part[lane * 6 + 3])) +
                                 (part[lane * 6 + 4] + part[lane * 6 + 5]));
          tv[lane] = t;
        }
        WAVE_SYNC();
        if (lane < 6) {
          const double y = ((cd[0] * tv[0] + cd[1] * tv[1]) + (cd[2] * tv[2] + cd[3] * tv[3])) + (cd[4] * tv[4] + cd[5] * tv[5]);
          yring[(i & kMaxBw) * 6 + lane] = y;
          z[6 * i + lane] = y;
        }
        WAVE_SYNC();
      }
    }
  }
  // ---------------- backward: x_i = Linv_ii^T (y_i - sum_k L_{i+k,i}^T x_{i+k}) ----------------
  // x of rows >= S must read as zero: clear the ring slots the first BW rows will look at
  for (int t = lane; t < (kMaxBw + 1) * 6; t += 64) yring[t] = 0.0;
  WAVE_SYNC();
  const double *bq = d.band + lt;  // element (c, r) of a block: L^T
  const double *dq = d.dinv + l6;
#pragma unroll
  for (int u = 0; u < kPD; u++) {
    const long row = d.S - 1 - u >= 0 ? d.S - 1 - u : 0;
#pragma unroll
    for (int k = 1; k <= BW; k++) pb[u][k - 1] = bq[(row + k < d.S ? row + k : row) * kRow + k * 36];
#pragma unroll
    for (int q = 0; q < 6; q++) pdv[u][q] = dq[row * 36 + q * 6];
    pr[u] = z[6 * row + l6];  // y of the forward sweep (written by this lane)
  }
  for (int i0 = d.S - 1; i0 >= 0; i0 -= kPD) {
#pragma unroll
    for (int u = 0; u < kPD; u++) {
      const int i = i0 - u;
      if (i >= 0) {
        double cb[BW], cd[6];
#pragma unroll
        for (int k = 0; k < BW; k++) cb[k] = pb[u][k];
#pragma unroll
        for (int q = 0; q < 6; q++) cd[q] = pdv[u][q];
        const double yi = pr[u];
        {
          const long row = i - kPD >= 0 ? i - kPD : 0;
#pragma unroll
          for (int k = 1; k <= BW; k++) pb[u][k - 1] = bq[(row + k < d.S ? row + k : row) * kRow + k * 36];
#pragma unroll
          for (int q = 0; q < 6; q++) pdv[u][q] = dq[row * 36 + q * 6];
          pr[u] = z[6 * row + l6];
        }
        {
          double a[3] = {0, 0, 0};
#pragma unroll
          for (int k = 1; k <= BW; k++) a[(k - 1) % 3] += cb[k - 1] * yring[((i + k) & kMaxBw) * 6 + c];
          if (lane < 36) part[lane] = (a[0] + a[1]) + a[2];
        }
        WAVE_SYNC();
        if (lane < 6) {
          const double t = yi - (((part[lane * 6] + part[lane * 6 + 1]) + (part[lane * 6 + 2] + part[lane * 6 + 3])) +
                                 (part[lane * 6 + 4] + part[lane * 6 + 5]));
          tv[lane] = t;
        }
        WAVE_SYNC();
        if (lane < 6) {
          const double x = ((cd[0] * tv[0] + cd[1] * tv[1]) + (cd[2] * tv[2] + cd[3] * tv[3])) + (cd[4] * tv[4] + cd[5] * tv[5]);
          yring[(i & kMaxBw) * 6 + lane] = x;
          z[6 * i + lane] = x;
        }
        WAVE_SYNC();
      }
    }
  }
  // camera blocks (generic mode: the border rows are gen_precond_border_kernel's)
  for (int cm = lane; cm < (d.gen ? 0 : d.NC); cm += 64) {
    const double *Bi = d.Binv + 36 * (long)d.S + 9 * cm, *rr = rin + d.cam0 + 3 * cm;
    for (int i = 0; i < 3; i++) z[d.cam0 + 3 * cm + i] = Bi[3 * i] * rr[0] + Bi[3 * i + 1] * rr[1] + Bi[3 * i + 2] * rr[2];
  }
}

typedef void (*band_solve_fn)(Dev, const double *, double *);
inline band_solve_fn band_solve_for(int bw) {
  switch (bw) {
    case 1: return band_solve_kernel<1>;
    case 2: return band_solve_kernel<2>;
    case 3: return band_solve_kernel<3>;
    case 4: return band_solve_kernel<4>;
    case 5: return band_solve_kernel<5>;
    case 6: return band_solve_kernel<6>;
    case 7: return band_solve_kernel<7>;
    case 8: return band_solve_kernel<8>;
    case 9: return band_solve_kernel<9>;
    case 10: return band_solve_kernel<10>;
    case 11: return band_solve_kernel<11>;
    case 12: return band_solve_kernel<12>;
    case 13: return band_solve_kernel<13>;
    case 14: return band_solve_kernel<14>;
    default: return band_solve_kernel<15>;
  }
}


// ---- cluster block-tridiagonal factorisation of the band ---------------------------------------
// With cs >= bw shots per cluster the banded shot system is exactly block TRIdiagonal in dense
// (6 cs)^2 blocks.  One 256-thread workgroup walks the chain: per cluster two small GEMMs, a dense
// Cholesky and a triangular inverse, all in LDS; the solve is then two sweeps of dense mat-vecs.
// ~500 chain links instead of 5 000, each with enough parallel work for a whole workgroup.
// The cluster factors are just a re-blocking of the band Cholesky factor (Cholesky is unique):
// L_{c,c} = the (6 cs)^2 lower-triangular diagonal block, L_{c,c-1} = the block below it.  So the
// sequential part stays band_cholesky_kernel; what is added is embarrassingly parallel: scatter
// the 6x6 factor blocks into dense cluster blocks and invert every diagonal block (one workgroup
// per cluster), so that the triangular sweeps become dense mat-vecs.
__global__ void ctri_scatterL_kernel(Dev d) {
  const int s = blockIdx.x;
  const int R1 = d.bw + 1;
  const long n2 = (long)d.ncd * d.ncd;
  for (int t = threadIdx.x; t < R1 * 36; t += blockDim.x) {
    const int k = t / 36, ij = t % 36, i = ij / 6, j = ij % 6;
    const int s2 = s - k;
    if (s2 < 0) continue;
    const double val = d.band[((long)s * R1 + k) * 36 + ij];
    const int c = s / d.cs, c2 = s2 / d.cs;
    const int rl = 6 * s + i - c * d.ncd;
    if (c2 == c) {
      const int cl = 6 * s2 + j - c * d.ncd;
      d.cD[c * n2 + (long)rl * d.ncd + cl] = val;  // lower triangular (k == 0 blocks have a zero upper part)
    } else {
      const int cl = 6 * s2 + j - c2 * d.ncd;
      d.cW[c * n2 + (long)rl * d.ncd + cl] = val;
      d.cWt[c * n2 + (long)cl * d.ncd + rl] = val;
    }
  }
}
__global__ void ctri_pad_kernel(Dev d) {  // identity on the padding rows of the last cluster
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int first = 6 * d.S - (d.ncl - 1) * d.ncd;
  if (p >= first && p < d.ncd) d.cD[(long)(d.ncl - 1) * d.ncd * d.ncd + (long)p * d.ncd + p] = 1.0;
}
// one workgroup (one wavefront) per cluster: inverse of the lower triangular diagonal block
__global__ void __launch_bounds__(64) ctri_inverse_kernel(Dev d) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int n = d.ncd, n2 = n * n, c = blockIdx.x, tid = threadIdx.x;
  double *L = lds, *X = lds + n2;
  const double *gL = d.cD + (long)c * n2;
  for (int t = tid; t < n2; t += 64) L[t] = gL[t];
  __syncthreads();
  if (tid < n) {
    const int cc = tid;
    for (int r = 0; r < n; r++) {
      double sum = (r == cc) ? 1.0 : 0.0;
      if (r >= cc) {
        double s0 = 0, s1 = 0;
        int q = cc;
        for (; q + 1 < r; q += 2) {
          s0 += L[r * n + q] * X[q * n + cc];
          s1 += L[r * n + q + 1] * X[(q + 1) * n + cc];
        }
        if (q < r) s0 += L[r * n + q] * X[q * n + cc];
        X[r * n + cc] = (sum - (s0 + s1)) / L[r * n + r];
      } else {
        X[r * n + cc] = 0.0;
      }
    }
  }
  __syncthreads();
  for (int t = tid; t < n2; t += 64) {
    const int r = t / n, q = t - r * n;
    d.cLi[(long)c * n2 + t] = X[t];
    d.cLit[(long)c * n2 + (long)q * n + r] = X[t];
  }
}

// ---- block cyclic reduction (BCR) of the cluster-tridiagonal system ----------------------------
// The chain L L^T above is inherently sequential (S/cs links).  Odd-even reduction removes every
// second cluster at once: at level l (stride st = 2^l) the clusters (2m+1) st are eliminated IN
// PARALLEL (one workgroup each: dense SPD inverse + two products), the clusters 2m st receive a
// Schur update and a new coupling to their neighbour 2 st away -- still block tridiagonal.  log2(N)
// levels instead of N links; the solve is one small kernel per level down and up.
//   eliminated i:  Dinv_i = D_i^-1,  G_i = Dinv_i E_i,  H_i = Dinv_i E_{i+st}^T      (E_k = A_{k,k-st})
//   kept j:        D_j -= E_j H_{j-st} + E_{j+st}^T G_{j+st},   E_j <- -E_j G_{j-st}
//   solve down:    b_j -= H_{j-st}^T b_{j-st} + G_{j+st}^T b_{j+st}
//   solve up:      x_i  = Dinv_i b_i - G_i x_{i-st} - H_i x_{i+st}
// ---- one launch per level (round 3) -----------------------------------------------------------------------------------
// The workgroup of the eliminated cluster i does everything that depends on D_i alone:
//   [D_i | E_i | E_r^T]  ->  [Dinv_i | G_i | H_i]   by in-place block Gauss-Jordan with 6 x 6 pivots (D_i is SPD: every pivot block is a
//                            Schur complement of it, no pivoting needed); one 6 x 6 tile of the n x 3n array per thread, CS steps of
//                            two barriers each -- the former Cholesky + triangular inverse + X^T X + two products were 114 us of
//                            mostly serial work per level, and the neighbours' update a second launch
//   D_{i+st} -= E_r H_i      into bD  (this workgroup is the only LEFT neighbour of i + st on this level)
//   D_{i-st} -= E_i^T G_i    into bD2 (... the only RIGHT neighbour of i - st): D_j = bD[j] + bD2[j], summed when j is loaded,
//                            so the two contributions need neither atomics nor a second launch and the sum is reproducible
//   E_{i+st} <- -E_r G_i     the new coupling of i + st to i - st
// G and H are also kept transposed (bGt, bHt): the solve's up sweep then reads columns, coalesced, like its down sweep.
__device__ __forceinline__ void inv6_spd(double (&a)[6][6], int &bad) {
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const double p = a[c][c];
    if (!(p > 0)) bad = 1;
    const double ip = 1.0 / p;
#pragma unroll
    for (int j = 0; j < 6; j++) a[c][j] = (j == c) ? ip : a[c][j] * ip;
#pragma unroll
    for (int r = 0; r < 6; r++) {
      if (r == c) continue;
      const double f = a[r][c];
#pragma unroll
      for (int j = 0; j < 6; j++) a[r][j] = (j == c) ? -f * ip : __builtin_fma(-f, a[c][j], a[r][j]);
    }
  }
}
#ifdef OSFM_BCR_UBENCH
__device__ int osfm_bcr_variant;
#define OSFM_BCR_SKIP(bit) (osfm_bcr_variant & (bit))
#else
#define OSFM_BCR_SKIP(bit) false
#endif
template <int CS>
struct BcrShape {
  static constexpr int n = 6 * CS, W = 3 * n, n2 = n * n;
  static constexpr int tiles = 3 * CS * CS;                 // 6 x 6 tiles of the augmented array = threads that own one
  static constexpr int threads = (tiles + 63) / 64 * 64;
  static constexpr size_t lds_bytes = (size_t)5 * n2 * sizeof(double);  // n x 3n + the two original couplings
};
template <int CS>
__global__ void __launch_bounds__(BcrShape<CS>::threads) bcr_level_kernel(Dev d, int st, int root, int *status) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int n = BcrShape<CS>::n, W = BcrShape<CS>::W, n2 = BcrShape<CS>::n2, T = BcrShape<CS>::threads;
  double *Aug = lds, *Ei = lds + n * W, *Er = Ei + n2;
  const int tid = threadIdx.x;
  const int i = root ? 0 : (2 * blockIdx.x + 1) * st;
  if (i >= d.ncl) return;
  const bool hasL = !root && i - st >= 0, hasR = !root && i + st < d.ncl;
  {
    // every global load of this workgroup is issued before the first LDS write (the blocks were written by the previous launch,
    // mostly on another XCD: each dependent round trip is a microsecond)
    const double *gD = d.bD + (long)i * n2, *gD2 = d.bD2 + (long)i * n2;
    const double *gEi = d.bE + (long)i * n2, *gEr = d.bE + (long)(hasR ? i + st : i) * n2;
    constexpr int NL = (n2 + T - 1) / T;
    double vD[NL], vE[NL], vR[NL];
#pragma unroll
    for (int u = 0; u < NL; u++) {
      const int t = tid + u * T;
      const bool in = t < n2;
      if (OSFM_BCR_SKIP(8)) {
        vD[u] = (t / n == t % n) ? 4.0 : 0.01;
        vE[u] = vR[u] = 0.02;
        continue;
      }
      vD[u] = in ? gD[t] + gD2[t] : 0.0;
      vE[u] = (in && hasL) ? gEi[t] : 0.0;
      vR[u] = (in && hasR) ? gEr[t] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NL; u++) {
      const int t = tid + u * T;
      if (t < n2) {
        const int r = t / n, c = t - r * n;
        Aug[r * W + c] = vD[u];
        Ei[t] = vE[u];
        Er[t] = vR[u];
        Aug[r * W + n + c] = vE[u];
        Aug[c * W + 2 * n + r] = vR[u];
      }
    }
  }
  __syncthreads();
  const bool has_tile = tid < BcrShape<CS>::tiles;
  const int tr = tid / (3 * CS), tq = tid - tr * (3 * CS);
  int bad = 0;
  double own[6][6];  // this thread's tile stays in registers over the steps (LDS keeps the copy the other threads read)
  if (has_tile) {
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) own[r][c] = Aug[(6 * tr + r) * W + 6 * tq + c];
  }
#pragma unroll 1
  for (int k = OSFM_BCR_SKIP(1) ? CS : 0; k < CS; k++) {
    if (has_tile) {
      double nr[6][6];  // the pivot block row after scaling, this tile's columns (the pivot block itself becomes Pinv)
      {
        double P[6][6];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int b = 0; b < 6; b++) P[a][b] = Aug[(6 * k + a) * W + 6 * k + b];
        inv6_spd(P, bad);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int c = 0; c < 6; c++) nr[a][c] = 0.0;
#pragma unroll
        for (int b = 0; b < 6; b++) {
          double pr[6];
#pragma unroll
          for (int c = 0; c < 6; c++) pr[c] = Aug[(6 * k + b) * W + 6 * tq + c];
#pragma unroll
          for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 6; c++) nr[a][c] = __builtin_fma(P[a][b], pr[c], nr[a][c]);
        }
        if (tq == k) {
#pragma unroll
          for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 6; c++) nr[a][c] = P[a][c];
        }
      }
#pragma unroll
      for (int r = 0; r < 6; r++) {
        double m[6];
#pragma unroll
        for (int b = 0; b < 6; b++) m[b] = Aug[(6 * tr + r) * W + 6 * k + b];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          double acc = (tq == k) ? 0.0 : own[r][c];
#pragma unroll
          for (int b = 0; b < 6; b++) acc = __builtin_fma(-m[b], nr[b][c], acc);
          own[r][c] = (tr == k) ? nr[r][c] : acc;
        }
      }
    }
    __syncthreads();
    if (has_tile) {
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) Aug[(6 * tr + r) * W + 6 * tq + c] = own[r][c];
    }
    __syncthreads();
  }
  if (bad) *status = 1;
  if (!OSFM_BCR_SKIP(2)) {
    double *oD = d.bD + (long)i * n2, *oG = d.bG + (long)i * n2, *oH = d.bH + (long)i * n2;
    double *oGt = d.bGt + (long)i * n2, *oHt = d.bHt + (long)i * n2;
    for (int t = tid; t < n2; t += T) {
      const int r = t / n, c = t - r * n;
      oD[t] = Aug[r * W + c];
      if (hasL) {
        oG[t] = Aug[r * W + n + c];
        oGt[t] = Aug[c * W + n + r];
      }
      if (hasR) {
        oH[t] = Aug[r * W + 2 * n + c];
        oHt[t] = Aug[c * W + 2 * n + r];
      }
    }
  }
  if (root || OSFM_BCR_SKIP(4)) return;
  // the three products for the neighbours: CS x CS tiles each, one per thread; the results go through LDS (the augmented array is
  // free by then) so that the read-modify-write of the neighbours' blocks is coalesced
  const int which = tid / (CS * CS), tl = tid - which * (CS * CS), pr = tl / CS, pc = tl - pr * CS;
  const bool act = has_tile && !((which == 0 && !hasR) || (which == 1 && !hasL) || (which == 2 && !(hasL && hasR)));
  double acc[6][6];
  if (act) {
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = 0; b < 6; b++) acc[a][b] = 0.0;
    const int ycol = (which == 0 ? 2 * n : n) + 6 * pc;
#pragma unroll 2
    for (int v = 0; v < n; v++) {
      double x[6], y[6];
#pragma unroll
      for (int a = 0; a < 6; a++) x[a] = (which == 1) ? Ei[v * n + 6 * pr + a] : Er[(6 * pr + a) * n + v];
#pragma unroll
      for (int b = 0; b < 6; b++) y[b] = Aug[v * W + ycol + b];
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 6; b++) acc[a][b] = __builtin_fma(x[a], y[b], acc[a][b]);
    }
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = 0; b < 6; b++) Aug[which * n2 + (6 * pr + a) * n + 6 * pc + b] = acc[a][b];
  }
  __syncthreads();
  if (hasR) {
    double *dD = d.bD + (long)(i + st) * n2;
    for (int t = tid; t < n2; t += T) dD[t] -= Aug[t];
  }
  if (hasL) {
    double *dD2 = d.bD2 + (long)(i - st) * n2;
    for (int t = tid; t < n2; t += T) dD2[t] -= Aug[n2 + t];
  }
  if (hasL && hasR) {
    double *dE = d.bE + (long)(i + st) * n2;
    for (int t = tid; t < n2; t += T) dE[t] = -Aug[2 * n2 + t];
  }
}

// assembled band -> level-0 blocks, one workgroup per cluster (gather: every entry of D, D2 = 0 and E is written, nothing to clear first)
__global__ void __launch_bounds__(256) bcr_build_kernel(Dev d, int *status) {
  const int c = blockIdx.x, n = d.ncd, R1 = d.bw + 1;
  const long n2 = (long)n * n;
  if (c == 0 && threadIdx.x == 0) *status = 0;
  for (int t = threadIdx.x; t < n2; t += 256) {
    const int rl = t / n, cl = t - rl * n;
    const int s = c * d.cs + rl / 6, i = rl % 6, s2 = c * d.cs + cl / 6, j = cl % 6;
    double v, e = 0.0;
    if (s >= d.S || s2 >= d.S)
      v = (rl == cl) ? 1.0 : 0.0;  // identity on the padding rows of the last cluster
    else {
      const int k = s - s2;
      v = k >= 0 ? (k <= d.bw ? d.band[((long)s * R1 + k) * 36 + i * 6 + j] : 0.0) : (-k <= d.bw ? d.band[((long)s2 * R1 - k) * 36 + j * 6 + i] : 0.0);
    }
    if (c > 0 && s < d.S) {
      const int ke = s - ((c - 1) * d.cs + cl / 6);
      if (ke <= d.bw) e = d.band[((long)s * R1 + ke) * 36 + i * 6 + j];
    }
    d.bD[c * n2 + t] = v;
    d.bD2[c * n2 + t] = 0.0;
    d.bE[c * n2 + t] = e;
  }
}

// ---- BCR solve: one 64-lane workgroup per cluster, dense mat-vecs from global (L2) ----
// sum_k (M_k^T x_k)[r] for up to three n x n blocks (n even, <= 64; a null block contributes nothing; x_k has 64 entries, zero from n
// on).  The loads of a 16-row chunk of every block are issued before the first multiply: the sweeps are chains of L2 round trips, and
// a plain dot-product loop waits for one of them per pair of rows (27 per block).  Per block the products are added in the order
// (even rows) + (odd rows).
template <int K>
__device__ __forceinline__ void coldots(const double *const (&M)[K], const double *const (&x)[K], int n, int r, double (&out)[K]) {
  double a0[K], a1[K];
#pragma unroll
  for (int k = 0; k < K; k++) a0[k] = a1[k] = 0.0;
  for (int q0 = 0; q0 < n; q0 += 16) {
    double m[K][16];
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
      for (int u = 0; u < 16; u++) m[k][u] = (M[k] && q0 + u < n) ? M[k][(q0 + u) * n + r] : 0.0;
#pragma unroll
    for (int k = 0; k < K; k++)
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        a0[k] += m[k][u] * x[k][q0 + u];
        a1[k] += m[k][u + 1] * x[k][q0 + u + 1];
      }
  }
#pragma unroll
  for (int k = 0; k < K; k++) out[k] = a0[k] + a1[k];
}
// The solve kernels take several right-hand sides at once (grid y = right-hand side q; vectors rin + q * rin_stride, z + q * z_stride,
// work vector bx + q * ncl * ncd): the columns of the camera border go through one walk of the levels instead of one walk each.
// (A single persistent launch for the whole walk was measured and dropped in round 2: cross-XCD synchronisation inside a kernel --
// agent-scope fences ~60 us, an arrival counter ~25 us, per-cluster progress words with sc1 accesses ~25 us per level -- costs more
// than the 5-10 us of a back-to-back launch on this part.)  Round 3: the levels that have at most kMidWaves clusters left run inside
// ONE workgroup (bcr_mid_kernel, a wavefront per cluster, __syncthreads between levels: no cross-XCD traffic at all), the first down
// level reads the right-hand side itself and the last up level writes the result: 2 log2(N) + 3 launches become 2 log2(st0) + 1.
// The right-hand sides of one walk: nrhs vectors r + q r_stride -> z + q z_stride, except index qx (if any), which is rx -> zx (the
// camera border's columns and the solve's own right-hand side live in different buffers but go through one walk); the right-hand
// side cam_q (if any) also gets the camera rows of its result (3x3 block Jacobi).
struct RhsSet {
  const double *r;
  long r_stride;
  double *z;
  long z_stride;
  int nrhs;
  const double *rx;
  double *zx;
  int qx, cam_q;
  __host__ __device__ const double *in(int q) const { return q == qx ? rx : r + q * r_stride; }
  __host__ __device__ double *out(int q) const { return q == qx ? zx : z + q * z_stride; }
};
__device__ __forceinline__ double bcr_rhs(const Dev &d, const double *rin, int cl, int r) {  // entry r of cluster cl of the padded rhs
  const long g = (long)cl * d.ncd + r;
  return g < 6L * d.S ? rin[g] : 0.0;
}
// first: b comes from the right-hand side itself (and the odd clusters' part is copied into bx on the way)
__global__ void __launch_bounds__(64) bcr_down_kernel(Dev d, int st, RhsSet rs, int first) {
  __shared__ double xl[64], xr[64];
  const int n = d.ncd, n2 = n * n, r = threadIdx.x;
  const int j = 2 * blockIdx.x * st;
  if (j >= d.ncl) return;
  double *bx = d.bx + (long)blockIdx.y * d.ncl * n;
  const double *b = first ? rs.in(blockIdx.y) : nullptr;
  const int i1 = j - st, i2 = j + st;
  if (b) {
    xl[r] = (i1 >= 0 && r < n) ? bcr_rhs(d, b, i1, r) : 0.0;
    xr[r] = (i2 < d.ncl && r < n) ? bcr_rhs(d, b, i2, r) : 0.0;
    if (i2 < d.ncl && r < n) bx[(long)i2 * n + r] = xr[r];
  } else {
    xl[r] = (i1 >= 0 && r < n) ? bx[(long)i1 * n + r] : 0.0;
    xr[r] = (i2 < d.ncl && r < n) ? bx[(long)i2 * n + r] : 0.0;
  }
  __syncthreads();
  if (r < n) {
    double v = b ? bcr_rhs(d, b, j, r) : bx[(long)j * n + r];
    const double *const Ms[2] = {i1 >= 0 ? d.bH + (long)i1 * n2 : nullptr, i2 < d.ncl ? d.bG + (long)i2 * n2 : nullptr};
    const double *const xs2[2] = {xl, xr};
    double t[2];
    coldots<2>(Ms, xs2, n, r, t);
    v -= t[0];
    v -= t[1];
    bx[(long)j * n + r] = v;
  }
}
// last: the result goes to its z, the even clusters' part is copied along, and the right-hand side cam_q gets its camera rows
__global__ void __launch_bounds__(64) bcr_up_kernel(Dev d, int st, RhsSet rs, int last) {
  __shared__ double xs[64], xl[64], xr[64];
  const int n = d.ncd, n2 = n * n, r = threadIdx.x;
  const int i = (2 * blockIdx.x + 1) * st;
  double *bx = d.bx + (long)blockIdx.y * d.ncl * n;
  double *zq = last ? rs.out(blockIdx.y) : nullptr;
  const int l = i - st, rr = i + st;
  if (zq) {
    if (l < d.ncl && r < n && (long)l * n + r < 6L * d.S) zq[(long)l * n + r] = bx[(long)l * n + r];
    if ((int)blockIdx.y == rs.cam_q) {
      const int g = blockIdx.x * 64 + r;
      if (g < d.NC) {
        const double *Bi = d.Binv + 36 * (long)d.S + 9 * g, *rc = rs.in(blockIdx.y) + d.cam0 + 3 * g;
        for (int q = 0; q < 3; q++) zq[d.cam0 + 3 * g + q] = Bi[3 * q] * rc[0] + Bi[3 * q + 1] * rc[1] + Bi[3 * q + 2] * rc[2];
      }
    }
  }
  if (i >= d.ncl) return;
  xs[r] = r < n ? bx[(long)i * n + r] : 0.0;
  xl[r] = (l >= 0 && r < n) ? bx[(long)l * n + r] : 0.0;
  xr[r] = (rr < d.ncl && r < n) ? bx[(long)rr * n + r] : 0.0;
  __syncthreads();
  if (r < n) {
    const double *const Ms[3] = {d.bD + (long)i * n2, l >= 0 ? d.bGt + (long)i * n2 : nullptr, rr < d.ncl ? d.bHt + (long)i * n2 : nullptr};
    const double *const xs3[3] = {xs, xl, xr};
    double t[3];
    coldots<3>(Ms, xs3, n, r, t);  // Dinv is symmetric
    const double v = (t[0] - t[1]) - t[2];
    if (zq) {
      if ((long)i * n + r < 6L * d.S) zq[(long)i * n + r] = v;
    } else
      bx[(long)i * n + r] = v;
  }
}
constexpr int kMidWaves = 16;
// right-hand sides per walk of the levels (grid y).  Seven until round 6: the sixteen border columns of a Brown camera with a free bias, and the solve's own
// right-hand side, were three walks of ~0.2 ms each whatever they carry -- the walk is the latency of its eleven levels, not their bytes
constexpr int kWalkRhs = 34;
__global__ void __launch_bounds__(64 * kMidWaves) bcr_mid_kernel(Dev d, int st0) {
  __shared__ double xs[kMidWaves][64], xl[kMidWaves][64], xr[kMidWaves][64];
  const int n = d.ncd, n2 = n * n, N = d.ncl;
  const int r = threadIdx.x & 63, w = threadIdx.x >> 6;
  double *bx = d.bx + (long)blockIdx.y * N * n;
  int st = st0;
  for (; st < N; st *= 2) {
    const int ne = (N + 2 * st - 1) / (2 * st);
    for (int b0 = 0; b0 < ne; b0 += kMidWaves) {
      const int j = 2 * (b0 + w) * st, i1 = j - st, i2 = j + st;
      const bool on = j < N;
      if (on) {
        xl[w][r] = (i1 >= 0 && r < n) ? bx[(long)i1 * n + r] : 0.0;
        xr[w][r] = (i2 < N && r < n) ? bx[(long)i2 * n + r] : 0.0;
      }
      __syncthreads();
      if (on && r < n) {
        double v = bx[(long)j * n + r];
        const double *const Ms[2] = {i1 >= 0 ? d.bH + (long)i1 * n2 : nullptr, i2 < N ? d.bG + (long)i2 * n2 : nullptr};
        const double *const xs2[2] = {xl[w], xr[w]};
        double t[2];
        coldots<2>(Ms, xs2, n, r, t);
        v -= t[0];
        v -= t[1];
        bx[(long)j * n + r] = v;
      }
      __syncthreads();
    }
  }
  if (w == 0) xs[0][r] = r < n ? bx[r] : 0.0;
  __syncthreads();
  if (w == 0 && r < n) {
    const double *const Ms[1] = {d.bD};
    const double *const xs1[1] = {xs[0]};
    double t[1];
    coldots<1>(Ms, xs1, n, r, t);
    bx[r] = t[0];
  }
  __syncthreads();
  for (st /= 2; st >= st0; st /= 2) {
    const int ne = (N + 2 * st - 1) / (2 * st);
    for (int b0 = 0; b0 < ne; b0 += kMidWaves) {
      const int i = (2 * (b0 + w) + 1) * st, l = i - st, rr = i + st;
      const bool on = i < N;
      if (on) {
        xs[w][r] = r < n ? bx[(long)i * n + r] : 0.0;
        xl[w][r] = (l >= 0 && r < n) ? bx[(long)l * n + r] : 0.0;
        xr[w][r] = (rr < N && r < n) ? bx[(long)rr * n + r] : 0.0;
      }
      __syncthreads();
      if (on && r < n) {
        const double *const Ms[3] = {d.bD + (long)i * n2, l >= 0 ? d.bGt + (long)i * n2 : nullptr, rr < N ? d.bHt + (long)i * n2 : nullptr};
        const double *const xs3[3] = {xs[w], xl[w], xr[w]};
        double t[3];
        coldots<3>(Ms, xs3, n, r, t);
        bx[(long)i * n + r] = (t[0] - t[1]) - t[2];
      }
      __syncthreads();
    }
  }
}

typedef void (*bcr_level_fn)(Dev, int, int, int *);
struct BcrLaunch {
  bcr_level_fn fn;
  int threads;
  size_t lds_bytes;
};
#define OSFM_CS_CASE(q) \
  case q: return BcrLaunch{bcr_level_kernel<q>, BcrShape<q>::threads, BcrShape<q>::lds_bytes};
inline BcrLaunch bcr_level_for(int cs) {
  switch (cs) {
    OSFM_CS_CASE(2) OSFM_CS_CASE(3) OSFM_CS_CASE(4) OSFM_CS_CASE(5) OSFM_CS_CASE(6) OSFM_CS_CASE(7) OSFM_CS_CASE(8) OSFM_CS_CASE(9)
    default: return BcrLaunch{bcr_level_kernel<10>, BcrShape<10>::threads, BcrShape<10>::lds_bytes};
  }
}

// ---- small problems: the exact band factorised by ONE workgroup, A = L D L^T (round 6) ---------------------------------------------
// Local bundle adjustment (BAHelpers::BundleLocal, ba_helpers.cc:117-311: ~50 shots, once per added image) has a handful of clusters: the
// cyclic reduction above is then build + four dependent level launches of one to three workgroups (36-48 us each: nine pivot steps of a
// 54 x 54 Gauss-Jordan and three products per level) + three walk launches -- 227 us of the 390 us an LM iteration takes.  When the whole
// band (S (bw + 1) blocks of 6 x 6) fits one workgroup's LDS it is factorised right-looking with 6 x 6 pivots in one launch:
//   step j:  Dinv_j = A_jj^-1,  L_ij = A_ij Dinv_j (j < i <= j + bw),  A_ik -= L_ij A_kj^T (j < k <= i <= j + bw)
// The chain is the pivot inverse (the same in-register Gauss-Jordan as the cyclic reduction's, ~0.5 us), so it is taken off the rest:
// wavefront 0 updates the next pivot block A_(j+1)(j+1) (an entry per lane) and inverts it while the other three wavefronts apply step j to
// every other block of the window (a 3 x 3 quarter per lane); behind a barrier wavefront 0 forms the rows of L_(., j+1) from the inverse in its
// registers; two barriers per step.  (Round 5 measured a one-workgroup band Cholesky at 4.3 us per shot -- five barriers and a serial 6 x 6
// factor per shot -- and dropped it.)
// Output, in the band's own layout: slot 0 of row j = the pivot block D_j (the solve inverts it again while it loads the factor: 48 threads, ~1 us),
// slot a = L_(j, j-a).  The band itself is left as assembled (the fallback reads it).
// Measured (profiles/r06_local_ba_kernels_by_grid.txt): 2.0 us per step -- ~450 fp64 / LDS instructions of wavefront 0 at the ~8 cycles each that a
// single wavefront per SIMD sustains -- 93 us for 48 shots against 209 us for build + four levels.  Measured and dropped: the pivot's inverse through
// its 3 x 3 quarters (adjugates, two divisions instead of six: 97 -> 93 us, and the facade's two-shot scene with weak priors lost three digits).
constexpr int kSbThreads = 256;
constexpr size_t kSbLdsMax = 160 * 1024 - 512;
constexpr int kSbSolveWaves = 4;  // right-hand sides per workgroup of the solve (a wavefront each, one copy of the factor in LDS)
inline size_t sband_factor_lds(int S, int bw) { return ((size_t)S * (bw + 1) * 36 + (size_t)2 * bw * 36 + (size_t)6 * S) * sizeof(double); }
inline size_t sband_solve_lds(int S, int bw) { return ((size_t)S * (bw + 1) * 36 + (size_t)kSbSolveWaves * 6 * S) * sizeof(double); }
// A_ik[3 qr .. +3][3 qc .. +3] -= (L_ij A_kj^T)[...]: Lr = rows 3 qr.. of L_ij, Ak = rows 3 qc.. of A_kj, Aq = the quarter's first entry
__device__ __forceinline__ void sband_quarter(const double *Lr, const double *Ak, double *Aq) {
  double l[3][6], a[3][6];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int m = 0; m < 6; m++) {
      l[r][m] = Lr[6 * r + m];
      a[r][m] = Ak[6 * r + m];
    }
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double acc = Aq[6 * r + c];
#pragma unroll
      for (int m = 0; m < 6; m++) acc = __builtin_fma(-l[r][m], a[c][m], acc);
      Aq[6 * r + c] = acc;
    }
}
// z_q = (L D L^T)^-1 r_q: a wavefront per right-hand side, kSbSolveWaves of them share the workgroup's copy of the factor.  Both sweeps are
// column-oriented (no reductions across lanes): forward, lane (a, r) takes L_(j+a, j)[r, :] y_j off row (j + a, r); backward, lane (a, c)
// takes L_(i, i-a)[:, c]^T x_i off row (i - a, c).  The camera rows of right-hand side cam_q: 3 x 3 block Jacobi, as bcr_up_kernel does.
// fuse.on (one right-hand side, every camera constant: local / pose-only bundle adjustment): the camera rows' blocks are formed here
// (precond_cam_kernel's launch) and the start of PCG follows the solve (pcg_init_kernel's launch: x = 0, r = b, p = z, y = sc p, r . z, b . b)
struct SbFuse {
  int on;
  double radius;
  double *x, *r, *p, *y, *o_rz, *o_bb;
  const double *sc;
};
// slot 0 of every row of the factor's LDS copy: the pivot block D_j -> Dinv_j (positive definite: the factorisation's status said so); all threads
__device__ __forceinline__ void sband_invert_pivots(double *Lb, int S, int R1, int tid, int nthreads) {
  for (int j = tid; j < S; j += nthreads) {
    double P[6][6];
    double *Dj = Lb + (size_t)j * R1 * 36;
    int bad = 0;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) P[r][c] = Dj[6 * r + c];
    inv6_spd(P, bad);
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) Dj[6 * r + c] = P[r][c];
  }
}
// one wavefront: xv = y (the forward sweep's result) -> z = Dinv y -> x = L^-T z -> the result's rows, the camera rows, the start of PCG
__device__ __forceinline__ void sband_back_half(const Dev &d, const double *Lb, double *xv, const RhsSet &rs, int q, const SbFuse &fuse, int lane) {
  const int S = d.S, bw = d.bw, R1 = bw + 1, n6 = 6 * S;
  const int a = 1 + lane / 6, r = lane - 6 * (a - 1);
  for (int t0 = 0; t0 < n6; t0 += 60) {  // z_j = Dinv_j y_j, ten shots at a time (a shot's six rows are read before any of them is written)
    const int t = t0 + lane;
    double s = 0.0;
    if (lane < 60 && t < n6) {
      const int j = t / 6, rr = t - 6 * j;
      const double *Dr = Lb + (size_t)j * R1 * 36 + 6 * rr;
#pragma unroll
      for (int m = 0; m < 6; m++) s = __builtin_fma(Dr[m], xv[6 * j + m], s);
    }
    WAVE_SYNC();
    if (lane < 60 && t < n6) xv[t] = s;
  }
  WAVE_SYNC();
  for (int i = S - 1; i >= 1; i--) {
    if (a <= min(bw, i)) {
      const double *Lc = Lb + ((size_t)i * R1 + a) * 36 + r;
      double s = xv[6 * (i - a) + r];
#pragma unroll
      for (int m = 0; m < 6; m++) s = __builtin_fma(-Lc[6 * m], xv[6 * i + m], s);
      xv[6 * (i - a) + r] = s;
    }
    WAVE_SYNC();
  }
  double *zq = rs.out(q);
  const double *rin = rs.in(q);
  double dots[2] = {0.0, 0.0};
  auto start_pcg = [&](int t, double zi) {
    const double bi = rin[t];
    fuse.x[t] = 0.0;
    fuse.r[t] = bi;
    fuse.p[t] = zi;
    fuse.y[t] = fuse.sc[t] * zi;
    dots[0] += bi * zi;
    dots[1] += bi * bi;
  };
  for (int t = lane; t < n6; t += 64) {
    zq[t] = xv[t];
    if (fuse.on) start_pcg(t, xv[t]);
  }
  if (q == rs.cam_q) {
    for (int g = lane; g < d.NC; g += 64) {
      if (fuse.on) precond_cam_one(d, g, fuse.radius);  // (this lane reads back what it has just written)
      const double *Bi = d.Binv + 36 * (long)d.S + 9 * g, *rc = rin + d.cam0 + 3 * g;
      for (int k = 0; k < 3; k++) {
        const double zi = Bi[3 * k] * rc[0] + Bi[3 * k + 1] * rc[1] + Bi[3 * k + 2] * rc[2];
        zq[d.cam0 + 3 * g + k] = zi;
        if (fuse.on) start_pcg(d.cam0 + 3 * g + k, zi);
      }
    }
  }
  if (fuse.on) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      dots[0] += __shfl_xor(dots[0], m);
      dots[1] += __shfl_xor(dots[1], m);
    }
    if (lane == 0) {
      *fuse.o_rz = dots[0];
      *fuse.o_bb = dots[1];
    }
  }
}
__global__ void __launch_bounds__(64 * kSbSolveWaves) sband_solve_kernel(Dev d, const double *sbL, RhsSet rs, SbFuse fuse) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int S = d.S, bw = d.bw, R1 = bw + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *Lb = lds, *xv = lds + (size_t)S * R1 * 36 + (size_t)wave * 6 * S;
  const int total = S * R1 * 36, n6 = 6 * S;
  for (int t = tid; t < total; t += 64 * kSbSolveWaves) Lb[t] = sbL[t];
  const int q = blockIdx.x * kSbSolveWaves + wave;
  const bool on = q < rs.nrhs;
  if (on) {
    const double *in = rs.in(q);
    for (int t = lane; t < n6; t += 64) xv[t] = in[t];
  }
  __syncthreads();
  sband_invert_pivots(Lb, S, R1, tid, 64 * kSbSolveWaves);
  __syncthreads();
  if (!on) return;
  const int a = 1 + lane / 6, r = lane - 6 * (a - 1);
  for (int j = 0; j + 1 < S; j++) {
    if (a <= min(bw, S - 1 - j)) {
      const double *Lr = Lb + ((size_t)(j + a) * R1 + a) * 36 + 6 * r;
      double s = xv[6 * (j + a) + r];
#pragma unroll
      for (int m = 0; m < 6; m++) s = __builtin_fma(-Lr[m], xv[6 * j + m], s);
      xv[6 * (j + a) + r] = s;
    }
    WAVE_SYNC();
  }
  sband_back_half(d, Lb, xv, rs, q, fuse, lane);
}

// fuse.on (one right-hand side rs, every camera constant): the solve rides along -- the forward sweep's step j by wavefront 3 beside the window's update,
// L_(., j) written over the dead blocks of column j so that the factor is in LDS when the loop ends, then sband_back_half: sband_solve_kernel's launch,
// its 138 KB load of the factor and its forward sweep are gone (local bundle adjustment: 68 + 28 us -> ~80 us)
__global__ void __launch_bounds__(kSbThreads) sband_factor_kernel(Dev d, double *sbL, int *status, RhsSet rs, SbFuse fuse) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ unsigned char pair_a[64], pair_b[64];
  const int S = d.S, bw = d.bw, R1 = bw + 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double *Ab = lds, *Lb = lds + (size_t)S * R1 * 36;  // the band; the rows of L_(., j) of two consecutive steps
  double *xv = Lb + (size_t)2 * bw * 36;             // fuse.on: the right-hand side on its way to the solution
  const int total = S * R1 * 36;
  for (int t = tid; t < total; t += kSbThreads) Ab[t] = d.band[t];
  if (fuse.on) {
    const double *in = rs.in(0);
    for (int t = tid; t < 6 * S; t += kSbThreads) xv[t] = in[t];
  }
  const int npairs = bw * (bw + 1) / 2 - 1;  // (a, b), 1 <= b <= a <= bw without (1, 1): the blocks wavefronts 1 .. 3 update
  if (tid == 0) {
    *status = 0;
    int n = 0;
    for (int a = 2; a <= bw; a++)
      for (int b = 1; b <= a; b++) {
        pair_a[n] = (unsigned char)a;
        pair_b[n] = (unsigned char)b;
        n++;
      }
  }
  __syncthreads();
  int bad = 0;
  double P[6][6];  // wavefront 0: Dinv_j, in every lane
  auto invert = [&](int j) {
    const double *Ajj = Ab + (size_t)j * R1 * 36;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) P[r][c] = Ajj[6 * r + c];
    inv6_spd(P, bad);
  };
  auto l_rows = [&](int j) {  // the rows of L_(., j) = A_(., j) Dinv_j: lane = (a - 1) 6 + r
    const int na = min(bw, S - 1 - j);
    double *Lj = Lb + (size_t)(j & 1) * bw * 36;
    if (lane < 6 * na) {
      const int a = 1 + lane / 6, r = lane - 6 * (a - 1);
      const double *Ar = Ab + ((size_t)(j + a) * R1 + a) * 36 + 6 * r;
      double x[6], v[6];
#pragma unroll
      for (int m = 0; m < 6; m++) x[m] = Ar[m];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 6; m++) s = __builtin_fma(x[m], P[m][c], s);
        v[c] = s;
      }
      double *go = sbL + ((size_t)(j + a) * R1 + a) * 36 + 6 * r;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        Lj[(a - 1) * 36 + 6 * r + c] = v[c];
        go[c] = v[c];
      }
    }
  };
  if (wave == 0) {
    invert(0);
    l_rows(0);
  }
  __syncthreads();
  for (int j = 0; j + 1 < S; j++) {
    const double *Lj = Lb + (size_t)(j & 1) * bw * 36;
    const int na = min(bw, S - 1 - j);
    if (wave == 0) {  // the next pivot block, an entry per lane, and its inverse
      if (lane < 36) {
        const int r = lane / 6, c = lane - 6 * r;
        const double *Lr = Lj + 6 * r, *Ak = Ab + ((size_t)(j + 1) * R1 + 1) * 36 + 6 * c;
        double *Aq = Ab + (size_t)(j + 1) * R1 * 36 + lane;
        double acc = *Aq;
#pragma unroll
        for (int m = 0; m < 6; m++) acc = __builtin_fma(-Lr[m], Ak[m], acc);
        *Aq = acc;
      }
      WAVE_SYNC();
      invert(j + 1);
    } else {  // every other block of the window
      for (int t = tid - 64; t < 4 * npairs; t += kSbThreads - 64) {
        const int p = t >> 2, q = t & 3, qr = q >> 1, qc = q & 1, a = pair_a[p], b = pair_b[p];
        if (a > na) continue;
        sband_quarter(Lj + (a - 1) * 36 + 18 * qr, Ab + ((size_t)(j + b) * R1 + b) * 36 + 18 * qc, Ab + ((size_t)(j + a) * R1 + (a - b)) * 36 + 18 * qr + 3 * qc);
      }
      if (fuse.on && wave == 3 && lane < 6 * na) {  // forward sweep, step j: y_j is final, row (j + a, r) loses L_(j+a, j)[r, :] y_j
        const int a = 1 + lane / 6, r = lane - 6 * (a - 1);
        const double *Lr = Lj + (a - 1) * 36 + 6 * r;
        double sacc = xv[6 * (j + a) + r];
#pragma unroll
        for (int m = 0; m < 6; m++) sacc = __builtin_fma(-Lr[m], xv[6 * j + m], sacc);
        xv[6 * (j + a) + r] = sacc;
      }
    }
    __syncthreads();
    if (wave == 0) l_rows(j + 1);  // (column j + 1 is complete: wavefronts 1 .. 3 updated its blocks below the pivot)
    if (fuse.on && wave == 1)      // column j's blocks are dead (step j has read them): L_(., j) takes their place
      for (int t = lane; t < 36 * na; t += 64) {
        const int a = 1 + t / 36, e = t - 36 * (a - 1);
        Ab[((size_t)(j + a) * R1 + a) * 36 + e] = Lj[t];
      }
    __syncthreads();
  }
  // the pivot blocks as they were when they were inverted (slot 0 of their rows: nothing writes them afterwards); the solve inverts them again
  // on its way in -- the inverse's 36 stores by masked lanes were 40 instructions on wavefront 0's chain of every step
  for (int t = tid; t < S * 36; t += kSbThreads) {
    const int j = t / 36, e = t - 36 * j;
    sbL[(size_t)j * R1 * 36 + e] = Ab[(size_t)j * R1 * 36 + e];
  }
  if (bad) *status = 1;
  if (!fuse.on) return;
  __syncthreads();
  sband_invert_pivots(Ab, S, R1, tid, kSbThreads);
  __syncthreads();
  if (wave == 0) sband_back_half(d, Ab, xv, rs, 0, fuse, lane);
}
// ---- wide band: direct block LDL^T of the shot-shot Schur complement (round 3) --------------------------------------------------
// Block surveys, loops, unordered collections: the co-visibility half-width is tens to hundreds of shots, beyond what the
// cluster-tridiagonal cyclic reduction can hold in LDS (and a band truncated to 15 shots is a preconditioner in name only: ~1000 CG
// iterations per LM iteration on a 50 x 100 grid).  The exact band is then factorised directly,  A = L D L^T  with kWB x kWB blocks
// (kWcs = 16 shots): D_J are the pivot blocks themselves (not factorised further: their inverses come from the same block
// Gauss-Jordan as the cyclic reduction's), L_IJ = A_IJ D_J^-1.  One launch per block column J, right-looking; the workgroup of window
// tile (I, K), J < K <= I <= J + Wb, inverts D_J itself (the launch is a chain of 16 pivot steps either way; a separate launch for it
// would only add its overhead), forms L_IJ and subtracts L_IJ A_KJ^T from A_IK.  Flops are n w^2 (11-44 GFLOP at configs[4] size with
// w = 600-1200): irrelevant; the factorisation is the latency of S / 16 launches x 16 pivot steps.
constexpr int kWcs = 16, kWB = 6 * kWcs, kWLd = kWB + 2;  // shots per block, block order, LDS row stride (even: 16-byte rows; 6 rows apart = 24 banks)
constexpr int kBandSlice = 520;                            // band columns the assembly's LDS accumulators hold (one copy of 520 x 36 doubles = 150 KB)
constexpr int kWMaxBw = 2079;                              // widest exact band: four assembly passes; the dense-cluster blocks are (6 bw)^2 doubles each

// in-place inverse of the SPD n x n block X (LDS, row stride kWLd), n = kWB: 16 x 16 tiles of 6 x 6, one per thread (256 threads)
__device__ __forceinline__ void wide_gj_inverse(double *X, int tid, int &bad, int npiv = kWB / 6) {
  constexpr int NT = kWB / 6;
  const int tr = tid / NT, tq = tid - tr * NT;
  double own[6][6];
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c < 6; c++) own[r][c] = X[(6 * tr + r) * kWLd + 6 * tq + c];
#pragma unroll 1
  for (int k = 0; k < npiv; k++) {  // (the rows from 6 npiv on are an identity block: nothing to eliminate)
    double nr[6][6];
    {
      double P[6][6];
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 6; b++) P[a][b] = X[(6 * k + a) * kWLd + 6 * k + b];
      inv6_spd(P, bad);
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c < 6; c++) nr[a][c] = 0.0;
#pragma unroll
      for (int b = 0; b < 6; b++) {
        double pr[6];
#pragma unroll
        for (int c = 0; c < 6; c++) pr[c] = X[(6 * k + b) * kWLd + 6 * tq + c];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int c = 0; c < 6; c++) nr[a][c] = __builtin_fma(P[a][b], pr[c], nr[a][c]);
      }
      if (tq == k) {
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int c = 0; c < 6; c++) nr[a][c] = P[a][c];
      }
    }
#pragma unroll
    for (int r = 0; r < 6; r++) {
      double m[6];
#pragma unroll
      for (int b = 0; b < 6; b++) m[b] = X[(6 * tr + r) * kWLd + 6 * k + b];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double acc = (tq == k) ? 0.0 : own[r][c];
#pragma unroll
        for (int b = 0; b < 6; b++) acc = __builtin_fma(-m[b], nr[b][c], acc);
        own[r][c] = (tr == k) ? nr[r][c] : acc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c < 6; c++) X[(6 * tr + r) * kWLd + 6 * tq + c] = own[r][c];
    __syncthreads();
  }
}

// ---- the same inverse on the matrix cores (round 5): 16 x 16 pivots ----
// Nothing ties the pivot size to a shot's six unknowns: every leading block of an SPD matrix can be eliminated without pivoting.  Here a
// step eliminates SIXTEEN unknowns with v_mfma_f64_16x16x4 on 16 x 16 tiles in LDS:
//   A  wavefront 0 inverts the pivot block (inv16_spd_wave: Gauss-Jordan on registers, a lane owns a column of four rows, rows and columns
//      travel by cross-lane reads)
//   B  the pivot block row in place, R'_j = P X_kj for the other column tiles, tiles dealt to the four wavefronts
//   C  X_ij -= X_ik R'_j for the tiles of the other rows; their pivot-column tiles -X_ik P go to Cbuf (X_ik is an operand of its whole row)
//   D  Cbuf -> the pivot columns
// six steps of four barriers for 96 unknowns.  Timestamps inside the kernel (MI355X, one workgroup per cluster, 90 unknowns): a step is
// A 3.0 + B 0.6 + C 2.7 + D 1.1 (first version: R' and the columns both through buffers) = 7.4 us -- the chain of sixteen divisions with
// two cross-lane round trips each sets the pace; 44 us for six steps against 45 us for the sixteen 6 x 6 steps of wide_gj_inverse, whose
// threads invert the pivot redundantly in registers.  Measured and dropped: tile (k + 1, k + 1) updated and inverted by wavefront 0
// BESIDE step k's other tiles (look-ahead inside the workgroup): 54 us against 50 for the kernel -- the cross-lane chain slows down by
// more than it hides when three other wavefronts keep the LDS busy.  What did pay in dgj_pivot_kernel was outside the inversion: the 36
// loads of the block issued before the first LDS write (60 -> 52 us).  X: row stride kWLd, identity beyond the block's order (its
// 16-blocks invert to themselves); n16 = 16-blocks to eliminate.  scratch: kGj16Scratch doubles of LDS.
constexpr int kGj16LdC = 18, kGj16Scratch = kWB * kGj16LdC;
typedef double gj_v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void inv16_spd_wave(double *M, int lane, int &bad) {  // M: 16 x 16, row stride kWLd, in LDS; one whole wavefront
  const int j = lane & 15, r0 = (lane >> 4) * 4;
  double v[4];
#pragma unroll
  for (int q = 0; q < 4; q++) v[q] = M[(r0 + q) * kWLd + j];
  // four pivots per trip of a loop that is NOT unrolled further: the code of a kernel that runs as one workgroup per cluster is fetched
  // cold, about a microsecond per kilobyte -- sixteen unrolled pivots were 19 us of instruction fetch before the first of them retired
#pragma unroll 1
  for (int cb = 0; cb < 4; cb++) {
    const int owner = cb << 4;  // the lanes that hold rows 4 cb .. 4 cb + 3
#pragma unroll
    for (int cq = 0; cq < 4; cq++) {
      const int c = 4 * cb + cq;
      const double rowc = __shfl(v[cq], owner | j);
      const double p = __shfl(v[cq], owner | c);
      if (!(p > 0)) bad = 1;
      const double ip = 1.0 / p;
      const double srow = (j == c) ? ip : rowc * ip;
      double f[4];
#pragma unroll
      for (int q = 0; q < 4; q++) f[q] = __shfl(v[q], (lane & 48) | c);
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = (r0 + q == c) ? srow : ((j == c) ? -f[q] * ip : __builtin_fma(-f[q], srow, v[q]));
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) M[(r0 + q) * kWLd + j] = v[q];
}
__device__ __forceinline__ void wide_gj_inverse_mfma(double *X, double *scratch, int tid, int &bad, int n16) {
  double *Cbuf = scratch;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll 1
  for (int k = 0; k < n16; k++) {
    double *Pk = X + (16 * k) * kWLd + 16 * k;
    if (wave == 0) inv16_spd_wave(Pk, lane, bad);
    __syncthreads();
    // B: the pivot block row in place, R'_j = P X_kj
    for (int j = wave; j < n16; j += 4) {
      if (j == k) continue;
      double *T = X + (16 * k) * kWLd + 16 * j;
      gj_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c = 0; c < 4; c++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pk[li * kWLd + 4 * c + lk], T[(4 * c + lk) * kWLd + li], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) T[(4 * r + lk) * kWLd + li] = acc[r];  // (the MFMAs have consumed the tile: one wavefront owns it)
    }
    __syncthreads();
    // C: X_ij -= X_ik R'_j for the other rows; their pivot-column tiles -X_ik P go to Cbuf (X_ik is an operand of the whole row)
    for (int u = wave; u < n16 * n16; u += 4) {
      const int i = u / n16, j = u - i * n16;
      if (i == k) continue;
      double a[4];
#pragma unroll
      for (int c = 0; c < 4; c++) a[c] = -X[(16 * i + li) * kWLd + 16 * k + 4 * c + lk];
      gj_v4d acc;
      if (j == k) {
        acc = (gj_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 4; c++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c], Pk[(4 * c + lk) * kWLd + li], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) Cbuf[(16 * i + 4 * r + lk) * kGj16LdC + li] = acc[r];
      } else {
        double *T = X + (16 * i) * kWLd + 16 * j;
        const double *Rj = X + (16 * k) * kWLd + 16 * j;
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = T[(4 * r + lk) * kWLd + li];
#pragma unroll
        for (int c = 0; c < 4; c++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c], Rj[(4 * c + lk) * kWLd + li], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) T[(4 * r + lk) * kWLd + li] = acc[r];
      }
    }
    __syncthreads();
    for (int t = tid; t < 16 * 16 * n16; t += 256) {  // Cbuf -> the pivot columns
      const int rr = t >> 4, cc = t & 15;
      if ((rr >> 4) != k) X[rr * kWLd + 16 * k + cc] = Cbuf[rr * kGj16LdC + cc];
    }
    __syncthreads();
  }
}

// assembled shot band -> kWB x kWB tiles: tile (J, dI) = A_{J+dI, J}, dI = 0 .. Wb (identity on the padding rows of the last block)
__global__ void __launch_bounds__(256) wide_tiles_kernel(Dev d, int *status) {
  const int J = blockIdx.x, dI = blockIdx.y, I = J + dI, R1 = d.bw + 1;
  if (J == 0 && dI == 0 && threadIdx.x == 0) status[2] = 0;
  double *T = d.wA + ((long)J * (d.wWb + 1) + dI) * kWB * kWB;
  for (int t = threadIdx.x; t < kWB * kWB; t += 256) {
    const int r = t / kWB, c = t - r * kWB;
    const int s = I * kWcs + r / 6, i = r % 6, s2 = J * kWcs + c / 6, j = c % 6;
    double v = 0.0;
    if (I < d.wNB) {
      if (s >= d.S || s2 >= d.S)
        v = (dI == 0 && r == c) ? 1.0 : 0.0;
      else {
        const int k = s - s2;
        if (k >= 0 && k <= d.bw) v = d.band[((long)s * R1 + k) * 36 + i * 6 + j];
        else if (k < 0 && -k <= d.bw) v = d.band[((long)s2 * R1 - k) * 36 + j * 6 + i];
      }
    }
    T[t] = v;
  }
}

// one block column of the factorisation; blockIdx.x = 0: store D_J^-1; else the window tile (dI, dK), 1 <= dK <= dI <= Wb
__global__ void __launch_bounds__(256) wide_factor_kernel(Dev d, int J, int *status) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *X = lds, *Y = lds + kWB * kWLd;
  const int tid = threadIdx.x, W1 = d.wWb + 1;
  int dI = 0, dK = 0;
  if (blockIdx.x > 0) {  // row-major enumeration of the lower triangle: index = dI (dI - 1) / 2 + (dK - 1)
    const int q = blockIdx.x - 1;
    dI = (int)((1.0 + sqrt(1.0 + 8.0 * q)) * 0.5);
    while (dI * (dI - 1) / 2 > q) dI--;
    while ((dI + 1) * dI / 2 <= q) dI++;
    dK = q - dI * (dI - 1) / 2 + 1;
    if (J + dI >= d.wNB) return;
  }
  const double *AJJ = d.wA + (long)J * W1 * kWB * kWB;
  double v[kWB * kWB / 256];
#pragma unroll
  for (int u = 0; u < kWB * kWB / 256; u++) v[u] = AJJ[tid + u * 256];
#pragma unroll
  for (int u = 0; u < kWB * kWB / 256; u++) {
    const int t = tid + u * 256;
    X[(t / kWB) * kWLd + t % kWB] = v[u];
  }
  if (blockIdx.x > 0) {  // A_IJ, issued before the inversion starts
    const double *AIJ = AJJ + (long)dI * kWB * kWB;
#pragma unroll
    for (int u = 0; u < kWB * kWB / 256; u++) v[u] = AIJ[tid + u * 256];
#pragma unroll
    for (int u = 0; u < kWB * kWB / 256; u++) {
      const int t = tid + u * 256;
      Y[(t / kWB) * kWLd + t % kWB] = v[u];
    }
  }
  __syncthreads();
  int bad = 0;
  wide_gj_inverse(X, tid, bad);
  if (bad) status[2] = 1;
  if (blockIdx.x == 0) {
    double *Dv = d.wDinv + (long)J * kWB * kWB;
    for (int t = tid; t < kWB * kWB; t += 256) Dv[t] = X[(t / kWB) * kWLd + t % kWB];
    return;
  }
  constexpr int NT = kWB / 6;
  const int tr = tid / NT, tq = tid - tr * NT;
  double acc[6][6];
  // L_IJ = A_IJ D^-1
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = 0; b < 6; b++) acc[a][b] = 0.0;
#pragma unroll 2
  for (int k = 0; k < kWB; k++) {
    double x[6], y[6];
#pragma unroll
    for (int a = 0; a < 6; a++) x[a] = Y[(6 * tr + a) * kWLd + k];
#pragma unroll
    for (int b = 0; b < 6; b++) y[b] = X[k * kWLd + 6 * tq + b];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = 0; b < 6; b++) acc[a][b] = __builtin_fma(x[a], y[b], acc[a][b]);
  }
  // A_KJ (transposed into X once D^-1 has been read by everyone), L_IJ into Y
  const double *AKJ = AJJ + (long)dK * kWB * kWB;
#pragma unroll
  for (int u = 0; u < kWB * kWB / 256; u++) v[u] = AKJ[tid + u * 256];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = 0; b < 6; b++) Y[(6 * tr + a) * kWLd + 6 * tq + b] = acc[a][b];
#pragma unroll
  for (int u = 0; u < kWB * kWB / 256; u++) {
    const int t = tid + u * 256;
    X[(t % kWB) * kWLd + t / kWB] = v[u];  // X[k][c] = A_KJ[c][k]
  }
  __syncthreads();
  if (dK == 1) {  // this workgroup keeps L_IJ for the solves, row-major and transposed
    double *Lr = d.wL + ((long)J * W1 + dI) * kWB * kWB, *Lt = d.wLt + ((long)J * W1 + dI) * kWB * kWB;
    for (int t = tid; t < kWB * kWB; t += 256) {
      Lr[t] = Y[(t / kWB) * kWLd + t % kWB];
      Lt[t] = Y[(t % kWB) * kWLd + t / kWB];
    }
  }
  // A_IK -= L_IJ A_KJ^T
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = 0; b < 6; b++) acc[a][b] = 0.0;
#pragma unroll 2
  for (int k = 0; k < kWB; k++) {
    double x[6], y[6];
#pragma unroll
    for (int a = 0; a < 6; a++) x[a] = Y[(6 * tr + a) * kWLd + k];
#pragma unroll
    for (int b = 0; b < 6; b++) y[b] = X[k * kWLd + 6 * tq + b];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = 0; b < 6; b++) acc[a][b] = __builtin_fma(x[a], y[b], acc[a][b]);
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int b = 0; b < 6; b++) X[(6 * tr + a) * kWLd + 6 * tq + b] = acc[a][b];
  __syncthreads();
  double *AIK = d.wA + ((long)(J + dK) * W1 + (dI - dK)) * kWB * kWB;
  for (int t = tid; t < kWB * kWB; t += 256) AIK[t] -= X[(t / kWB) * kWLd + t % kWB];
}

// ---- wide band solve: L y = b, z = D^-1 y (all blocks at once), L^T x = z.  NR right-hand sides side by side:
// wx[(block * kWB + r) * NR + q].  Every dot product is split over kWKS adjacent lanes.
constexpr int kWKS = 4;
template <int NR>
__device__ __forceinline__ void lds_vec(const double *p, double (&v)[NR]) {
  if constexpr (NR == 4) {
    const double2 a = reinterpret_cast<const double2 *>(p)[0], b = reinterpret_cast<const double2 *>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
  } else {
#pragma unroll
    for (int q = 0; q < NR; q++) v[q] = p[q];
  }
}
template <int NR>
__global__ void __launch_bounds__(256) wide_diag_kernel(Dev d) {
  __shared__ double ys[kWB * NR];
  const int J = blockIdx.x, tid = threadIdx.x;
  for (int t = tid; t < kWB * NR; t += 256) ys[t] = d.wx[(long)J * kWB * NR + t];
  __syncthreads();
  if (tid < kWB) {
    const double *Dv = d.wDinv + (long)J * kWB * kWB + tid;  // symmetric: column tid
    double a[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) a[q] = 0.0;
    for (int k0 = 0; k0 < kWB; k0 += 16) {
      double m[16];
#pragma unroll
      for (int u = 0; u < 16; u++) m[u] = Dv[(long)(k0 + u) * kWB];
#pragma unroll
      for (int u = 0; u < 16; u++)
#pragma unroll
        for (int q = 0; q < NR; q++) a[q] = __builtin_fma(m[u], ys[(k0 + u) * NR + q], a[q]);
    }
#pragma unroll
    for (int q = 0; q < NR; q++) d.wx[((long)J * kWB + tid) * NR + q] = a[q];
  }
}
// The two sweeps, one small launch per block column in "push" form: the finished block J subtracts its products from the blocks it
// couples to -- forward the Wb blocks below it through L, backward the Wb blocks above it through L^T.  A workgroup per target block, a
// row (column) per four lanes, 24 loads in flight per lane: a step is a launch gap plus one round trip spread over Wb CUs.  (Round 3
// first had ONE workgroup walk all block columns with the window of the right-hand side in an LDS ring: 30 us per step -- one CU
// streaming half a megabyte and re-reading y from LDS in every lane -- against ~9 us here: 57.8 -> 39.2 ms per LM iteration on the
// 50 x 100 grid.)
template <int NR, bool BACK>
__global__ void __launch_bounds__(kWB * kWKS) wide_push_kernel(Dev d, int J) {
  __shared__ __attribute__((aligned(16))) double ys[kWB * NR];
  const int tid = threadIdx.x, W1 = d.wWb + 1, dI = blockIdx.x + 1;
  constexpr int KC = kWB / kWKS, BN = kWB * NR;
  const int I = BACK ? J - dI : J + dI;  // the block that receives
  if (I < 0 || I >= d.wNB) return;
  for (int t = tid; t < BN; t += kWB * kWKS) ys[t] = d.wx[(long)J * BN + t];
  __syncthreads();
  const int row = tid / kWKS, ks = tid % kWKS;
  // forward: L_{I,J} = tile (J, dI), transposed copy [k][r];  backward: (L_{J,I})^T, L_{J,I} = tile (I, dI), row-major [r][c]
  const double *T = (BACK ? d.wL + ((long)I * W1 + dI) * kWB * kWB : d.wLt + ((long)J * W1 + dI) * kWB * kWB) + (long)ks * KC * kWB + row;
  double m[KC];
#pragma unroll
  for (int k = 0; k < KC; k++) m[k] = T[(long)k * kWB];
  double a[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) a[q] = 0.0;
#pragma unroll
  for (int k = 0; k < KC; k++) {
    double y[NR];
    lds_vec<NR>(ys + (ks * KC + k) * NR, y);
#pragma unroll
    for (int q = 0; q < NR; q++) a[q] = __builtin_fma(m[k], y[q], a[q]);
  }
#pragma unroll
  for (int q = 0; q < NR; q++) {
    a[q] += __shfl_xor(a[q], 1);
    a[q] += __shfl_xor(a[q], 2);
  }
  if (ks == 0) {
    double *dst = d.wx + ((long)I * kWB + row) * NR;
#pragma unroll
    for (int q = 0; q < NR; q++) dst[q] -= a[q];
  }
}

// right-hand sides in / results out (NR side by side in wx)
template <int NR>
__global__ void wide_load_kernel(Dev d, RhsSet rs) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long)d.wNB * kWB) return;
#pragma unroll
  for (int q = 0; q < NR; q++) d.wx[g * NR + q] = (q < rs.nrhs && g < 6L * d.S) ? rs.in(q)[g] : 0.0;
}
template <int NR>
__global__ void wide_store_kernel(Dev d, RhsSet rs) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < 6L * d.S)
    for (int q = 0; q < rs.nrhs; q++) rs.out(q)[g] = d.wx[g * NR + q];
  if (rs.cam_q >= 0 && g < d.NC) {
    const double *Bi = d.Binv + 36 * (long)d.S + 9 * g, *rr = rs.in(rs.cam_q) + d.cam0 + 3 * g;
    double *z = rs.out(rs.cam_q);
    for (int i = 0; i < 3; i++) z[d.cam0 + 3 * g + i] = Bi[3 * i] * rr[0] + Bi[3 * i + 1] * rr[1] + Bi[3 * i + 2] * rr[2];
  }
}


// ---- wide band, round 4: block cyclic reduction over DENSE clusters ---------------------------------------------------------------
// The block LDL^T above is a chain of S / 16 launches (28 ms at 5 000 shots) whatever the half-width.  Grouping qcs >= bw shots into
// one cluster makes the band matrix block tridiagonal with qm = 6 qcs (150 ... 3 000) unknowns per block, and cyclic reduction
// eliminates every second cluster of a level at once: log2(S / bw) levels (6-8) instead of S / 16 steps.  The blocks no longer fit a
// workgroup's LDS, so they live in HBM (column-major) and a level works on all its clusters at once:
//   D_i <- D_i^-1                       blocked Gauss-Jordan, in place: per panel of <= 96 columns the pivot block is inverted in LDS
//                                       (dgj_pivot_kernel: the 6 x 6-pivot elimination of the other solvers) and the panel / trailing
//                                       products are batched dgemm (dgemm_mfma_kernel below, hand-written since round 5); ceil(qm / 96) steps per level
//   G_i = D_i^-1 E_i,  H_i = D_i^-1 E_r^T,   D_{i+st} -= E_r H_i,   D_{i-st} -= E_i^T G_i,   E_{i+st} <- -E_r G_i      batched dgemm
// (same algebra as bcr_level_kernel; the two updates of a surviving D are separate launches on one stream, so no second accumulator).
// rocSOLVER's batched potrf / potrs were measured first (round 4): at qm = 612 they are chains of 24- and 32-thread kernels, 25 ms per
// LM iteration on the 50 x 100 grid.  The solve's sweeps (a wavefront per row, deterministic sums) are hand-written below.
__global__ void __launch_bounds__(256) dbcr_build_kernel(Dev d, int *status) {
  const int c = blockIdx.y, m = d.qm, R1 = d.bw + 1;
  const long m2 = (long)m * m, t = (long)blockIdx.x * 256 + threadIdx.x;
  if (c == 0 && blockIdx.x == 0 && threadIdx.x == 0) status[2] = 0;
  if (t >= m2) return;
  const int cl = (int)(t / m), rl = (int)(t - (long)cl * m);  // column-major: entry (rl, cl)
  const int s = c * d.qcs + rl / 6, i = rl % 6, s2 = c * d.qcs + cl / 6, j = cl % 6;
  double v, e = 0.0;
  if (s >= d.S || s2 >= d.S)
    v = (rl == cl) ? 1.0 : 0.0;  // identity on the padding rows of the last cluster
  else {
    const int k = s - s2;
    v = k >= 0 ? (k <= d.bw ? d.band[((long)s * R1 + k) * 36 + i * 6 + j] : 0.0) : (-k <= d.bw ? d.band[((long)s2 * R1 - k) * 36 + j * 6 + i] : 0.0);
  }
  if (c > 0 && s < d.S) {  // E_c(r, col): row r of cluster c against column col of cluster c - 1
    const int ke = s - ((c - 1) * d.qcs + cl / 6);
    if (ke <= d.bw) e = d.band[((long)s * R1 + ke) * 36 + i * 6 + j];
  }
  d.qD[(long)c * m2 + t] = v;
  d.qE[0][(long)c * m2 + t] = e;
}
// One panel step of the in-place inversion, first launch: workgroup 0 of a cluster inverts the pivot block A[j0 : j0 + w, j0 : j0 + w] (a
// Schur complement of an SPD matrix: symmetric, no pivoting) into P; the others copy the row panel (w x m -> R, leading dimension T) and
// the column panel (m x w -> C, its pivot rows zeroed: the trailing product must leave the pivot rows alone).
__global__ void __launch_bounds__(256) dgj_pivot_kernel(double *A0, long strideA, int m, int T, int j0, int w, double *P0, double *R0, double *C0,
                                                        int *status) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  double *A = A0 + (long)blockIdx.y * strideA;
  if (blockIdx.x == 0) {
    double *X = lds;
    {
      // all 36 loads of a thread are in flight before the first LDS write (a load -> store loop is 36 round trips to L2: 25 us of the
      // kernel's 60 when one workgroup per cluster is all there is on the chip)
      constexpr int NL = kWB * kWB / 256;
      double v[NL];
#pragma unroll
      for (int u = 0; u < NL; u++) {
        const int t = tid + 256 * u, c = t / kWB, r = t - c * kWB;
        v[u] = (r < w && c < w) ? A[(long)(j0 + c) * m + j0 + r] : (r == c ? 1.0 : 0.0);
      }
#pragma unroll
      for (int u = 0; u < NL; u++) {
        const int t = tid + 256 * u, c = t / kWB, r = t - c * kWB;
        X[r * kWLd + c] = v[u];
      }
    }
    __syncthreads();
    int bad = 0;
    wide_gj_inverse_mfma(X, lds + kWB * kWLd, tid, bad, (w + 15) / 16);
    if (bad) status[2] = 1;
    double *P = P0 + (long)blockIdx.y * T * T;
    for (int t = tid; t < w * w; t += 256) {
      const int c = t / w, r = t - c * w;
      P[(long)c * T + r] = X[r * kWLd + c];
    }
    return;
  }
  double *R = R0 + (long)blockIdx.y * T * m, *C = C0 + (long)blockIdx.y * T * m;
  const long n = (long)w * m, nth = (long)(gridDim.x - 1) * 256;
  for (long t = (long)(blockIdx.x - 1) * 256 + tid; t < n; t += nth) {
    {  // row panel: entry (r, c), r < w: consecutive threads along a column of A
      const int c = (int)(t / w), r = (int)(t - (long)c * w);
      R[(long)c * T + r] = A[(long)c * m + j0 + r];
    }
    {  // column panel: entry (r, c), c < w
      const int c = (int)(t / m), r = (int)(t - (long)c * m);
      C[(long)c * m + r] = (r >= j0 && r < j0 + w) ? 0.0 : A[(long)(j0 + c) * m + r];
    }
  }
}
// last launch of the step: the pivot rows become P A_J,: (Rn), the pivot block P
__global__ void __launch_bounds__(256) dgj_scatter_kernel(double *A0, long strideA, int m, int T, int j0, int w, const double *P0, const double *Rn0) {
  double *A = A0 + (long)blockIdx.y * strideA;
  const double *P = P0 + (long)blockIdx.y * T * T, *Rn = Rn0 + (long)blockIdx.y * T * m;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)w * m) return;
  const int c = (int)(t / w), r = (int)(t - (long)c * w);
  A[(long)c * m + j0 + r] = (c >= j0 && c < j0 + w) ? P[(long)(c - j0) * T + r] : Rn[(long)c * T + r];
}
// ---- the same inversion with the pivot chain taken off the trailing product (look-ahead) ----
// Of everything panel J + 1 needs from panel J's trailing product A -= C R', only its own pivot block is on the critical path: it is
// B - C[rows J + 1] R'[:, columns J + 1], a 96^3 product.  dgj_copy_kernel saves that block (B) with the panels before the products of
// panel J start; as soon as R' exists, dgj_pivot_ahead_kernel forms the updated block itself, inverts it on a second stream and hands
// P_{J+1} over with an event, while the main stream runs the three products and the scatter of panel J.
__global__ void __launch_bounds__(256) dgj_copy_kernel(const double *A0, long strideA, int m, int T, int j0, int w, int jn, int wn, double *R0, double *C0,
                                                       double *B0) {
  const double *A = A0 + (long)blockIdx.y * strideA;
  double *R = R0 + (long)blockIdx.y * T * m, *C = C0 + (long)blockIdx.y * T * m, *B = B0 + (long)blockIdx.y * T * T;
  const long n = (long)w * m, nth = (long)gridDim.x * 256;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += nth) {
    {
      const int c = (int)(t / w), r = (int)(t - (long)c * w);
      R[(long)c * T + r] = A[(long)c * m + j0 + r];
    }
    {
      const int c = (int)(t / m), r = (int)(t - (long)c * m);
      C[(long)c * m + r] = (r >= j0 && r < j0 + w) ? 0.0 : A[(long)(j0 + c) * m + r];
    }
    if (t < (long)wn * wn) {
      const int c = (int)(t / wn), r = (int)(t - (long)c * wn);
      B[(long)c * T + r] = A[(long)(jn + c) * m + jn + r];
    }
  }
}
// first: the pivot block is A's own (panel 0 of a level); else B - C[rows jn ..] R'[:, columns jn ..] with the panels of width w before it
__global__ void __launch_bounds__(256) dgj_pivot_ahead_kernel(const double *A0, long strideA, int m, int T, int first, int w, int jn, int wn, const double *C0,
                                                              const double *Rn0, const double *B0, double *P0, int *status) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *X = lds, *Z = lds + kWB * kWLd;
  const int tid = threadIdx.x;
  if (first) {
    const double *A = A0 + (long)blockIdx.y * strideA;
    for (int t = tid; t < kWB * kWB; t += 256) {
      const int c = t / kWB, r = t - c * kWB;
      X[r * kWLd + c] = (r < wn && c < wn) ? A[(long)(jn + c) * m + jn + r] : (r == c ? 1.0 : 0.0);
    }
  } else {
    const double *C = C0 + (long)blockIdx.y * T * m, *Rn = Rn0 + (long)blockIdx.y * T * m, *B = B0 + (long)blockIdx.y * T * T;
    for (int t = tid; t < kWB * kWB; t += 256) {
      const int k = t / kWB, r = t - k * kWB;  // X[r][k] = C(jn + r, k): consecutive threads along a column of C
      X[r * kWLd + k] = (r < wn && k < w) ? C[(long)k * m + jn + r] : 0.0;
    }
    for (int t = tid; t < kWB * kWB; t += 256) {
      const int c = t / kWB, k = t - c * kWB;  // Z[k][c] = R'(k, jn + c)
      Z[k * kWLd + c] = (k < w && c < wn) ? Rn[(long)(jn + c) * T + k] : 0.0;
    }
    __syncthreads();
    constexpr int NT = kWB / 6;
    const int tr = tid / NT, tq = tid - tr * NT;
    double acc[6][6];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = 0; b < 6; b++) acc[a][b] = 0.0;
#pragma unroll 2
    for (int k = 0; k < kWB; k++) {
      double x[6], y[6];
#pragma unroll
      for (int a = 0; a < 6; a++) x[a] = X[(6 * tr + a) * kWLd + k];
#pragma unroll
      for (int b = 0; b < 6; b++) y[b] = Z[k * kWLd + 6 * tq + b];
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 6; b++) acc[a][b] = __builtin_fma(x[a], y[b], acc[a][b]);
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int b = 0; b < 6; b++) {
        const int r = 6 * tr + a, c = 6 * tq + b;
        X[r * kWLd + c] = (r < wn && c < wn) ? B[(long)c * T + r] - acc[a][b] : (r == c ? 1.0 : 0.0);
      }
  }
  __syncthreads();
  int bad = 0;
  wide_gj_inverse_mfma(X, Z, tid, bad, (wn + 15) / 16);  // (Z held an operand of the product: free by now)
  if (bad) status[2] = 1;
  double *P = P0 + (long)blockIdx.y * T * T;
  for (int t = tid; t < wn * wn; t += 256) {
    const int c = t / wn, r = t - c * wn;
    P[(long)c * T + r] = X[r * kWLd + c];
  }
}
// dst_(k, w) = src_(k, w)^T for m x m column-major blocks; block (k, w) at base + k * stride_k + w * stride_w; 32 x 32 tiles through LDS
__global__ void __launch_bounds__(256) dbcr_transpose_kernel(const double *src, long src_k, long src_w, double *dst, long dst_k, long dst_w, int m, int nw) {
  __shared__ double tile[32][33];
  const int k = blockIdx.z / nw, w = blockIdx.z - k * nw;
  const double *S = src + (long)k * src_k + (long)w * src_w;
  double *D = dst + (long)k * dst_k + (long)w * dst_w;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int c = c0 + ty + 8 * u, r = r0 + tx;
    if (r < m && c < m) tile[ty + 8 * u][tx] = S[(long)c * m + r];  // tile[c - c0][r - r0] = S(r, c)
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int c = r0 + ty + 8 * u, r = c0 + tx;  // D(r, c) = S(c, r)
    if (r < m && c < m) D[(long)c * m + r] = tile[tx][ty + 8 * u];
  }
}
// One level of a solve, NR right-hand sides side by side: a wavefront per row of a target cluster, its lanes along the (contiguous)
// column of every block that feeds the row, fixed summation order.  mode 0 (down): b_j -= H_{j-st}^T b_{j-st} + G_{j+st}^T b_{j+st} in
// place in qx; mode 1 (up): x_i = D_i^-1 b_i - G_i x_{i-st} - H_i x_{i+st} into qy (transposed copies: row r of G = column r of G^T);
// mode 2: x_0 = D_0^-1 b_0
template <int NR>
__global__ void __launch_bounds__(256) dbcr_sweep_kernel(Dev d, int st, int mode) {
  const int m = d.qm, N = d.qN;
  const long m2 = (long)m * m;
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  int tgt;
  const double *M[3] = {nullptr, nullptr, nullptr}, *x[3] = {nullptr, nullptr, nullptr};
  if (mode == 0) {
    tgt = 2 * blockIdx.y * st;
    if (tgt >= N) return;
    const int i1 = tgt - st, i2 = tgt + st;
    if (i1 >= 0) {
      M[0] = d.qX + (long)i1 * 2 * m2 + m2;
      x[0] = d.qx + (long)i1 * m * NR;
    }
    if (i2 < N) {
      M[1] = d.qX + (long)i2 * 2 * m2;
      x[1] = d.qx + (long)i2 * m * NR;
    }
  } else if (mode == 1) {
    tgt = (2 * blockIdx.y + 1) * st;
    if (tgt >= N) return;
    M[0] = d.qD + (long)tgt * m2;  // D^-1 is symmetric: row r = column r
    x[0] = d.qx + (long)tgt * m * NR;
    M[1] = d.qXt + (long)tgt * 2 * m2;
    x[1] = d.qy + (long)(tgt - st) * m * NR;
    if (tgt + st < N) {
      M[2] = d.qXt + (long)tgt * 2 * m2 + m2;
      x[2] = d.qy + (long)(tgt + st) * m * NR;
    }
  } else {
    tgt = 0;
    M[0] = d.qD;
    x[0] = d.qx;
  }
  if (r >= m) return;  // whole wavefronts
  double acc[3][NR];
#pragma unroll
  for (int p = 0; p < 3; p++)
#pragma unroll
    for (int q = 0; q < NR; q++) acc[p][q] = 0.0;
#pragma unroll
  for (int p = 0; p < 3; p++) {
    if (!M[p]) continue;
    const double *col = M[p] + (long)r * m;
    for (int k0 = lane; k0 < m; k0 += 256) {  // four loads of the column in flight per lane
      double mv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) mv[u] = k0 + 64 * u < m ? col[k0 + 64 * u] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (k0 + 64 * u >= m) continue;
        const double *xv = x[p] + (long)(k0 + 64 * u) * NR;
#pragma unroll
        for (int q = 0; q < NR; q++) acc[p][q] = __builtin_fma(mv[u], xv[q], acc[p][q]);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 3; p++)
#pragma unroll
    for (int q = 0; q < NR; q++)
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) acc[p][q] += __shfl_xor(acc[p][q], o);
  if (lane == 0) {
    if (mode == 0) {
      double *b = d.qx + ((long)tgt * m + r) * NR;
#pragma unroll
      for (int q = 0; q < NR; q++) b[q] = (b[q] - acc[0][q]) - acc[1][q];
    } else {
      double *o = d.qy + ((long)tgt * m + r) * NR;
#pragma unroll
      for (int q = 0; q < NR; q++) o[q] = (acc[0][q] - acc[1][q]) - acc[2][q];
    }
  }
}
template <int NR>
__global__ void dbcr_load_kernel(Dev d, RhsSet rs) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long)d.qN * d.qm) return;
#pragma unroll
  for (int q = 0; q < NR; q++) d.qx[g * NR + q] = (q < rs.nrhs && g < 6L * d.S) ? rs.in(q)[g] : 0.0;
}
template <int NR>
__global__ void dbcr_store_kernel(Dev d, RhsSet rs) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < 6L * d.S)
    for (int q = 0; q < rs.nrhs; q++) rs.out(q)[g] = d.qy[g * NR + q];
  if (rs.cam_q >= 0 && g < d.NC) {
    const double *Bi = d.Binv + 36 * (long)d.S + 9 * g, *rr = rs.in(rs.cam_q) + d.cam0 + 3 * g;
    double *z = rs.out(rs.cam_q);
    for (int i = 0; i < 3; i++) z[d.cam0 + 3 * g + i] = Bi[3 * i] * rr[0] + Bi[3 * i + 1] * rr[1] + Bi[3 * i + 2] * rr[2];
  }
}

// z_shots = (L L^T)^-1 r_shots with the cluster factors; camera rows: 3x3 block Jacobi
__global__ void __launch_bounds__(256) ctri_solve_kernel(Dev d, const double *rin, double *z) {
  __shared__ double yprev[64], tvec[64], part[4][64];
  const int n = d.ncd, n2 = n * n;
  const int tid = threadIdx.x, r = tid & 63, pq = tid >> 6;
  const int q0 = pq * 16;
  const int nshot = 6 * d.S;
  if (tid < 64) yprev[tid] = 0.0;
  __syncthreads();
  // forward: y_c = Li_c (b_c - W_c y_{c-1})
  for (int c = 0; c < d.ncl; c++) {
    const double *Wc = d.cW + (long)c * n2, *Li = d.cLi + (long)c * n2;
    double acc = 0;
    if (c > 0 && r < n)
#pragma unroll 4
      for (int q = q0; q < q0 + 16 && q < n; q++) acc += Wc[r * n + q] * yprev[q];
    part[pq][r] = acc;
    __syncthreads();
    if (tid < 64) {
      const int g = c * n + tid;
      const double b = (tid < n && g < nshot) ? rin[g] : 0.0;
      tvec[tid] = b - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
    }
    __syncthreads();
    acc = 0;
    if (r < n)
#pragma unroll 4
      for (int q = q0; q < q0 + 16 && q < n; q++) acc += Li[r * n + q] * tvec[q];
    part[pq][r] = acc;
    __syncthreads();
    if (tid < 64) {
      const double y = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
      yprev[tid] = y;
      const int g = c * n + tid;
      if (tid < n && g < nshot) z[g] = y;
    }
    __syncthreads();
  }
  // backward: x_c = Li_c^T (y_c - W_{c+1}^T x_{c+1})
  if (tid < 64) yprev[tid] = 0.0;
  __syncthreads();
  for (int c = d.ncl - 1; c >= 0; c--) {
    const double *Wt = d.cWt + (long)(c + 1) * n2, *Lit = d.cLit + (long)c * n2;
    double acc = 0;
    if (c + 1 < d.ncl && r < n)
#pragma unroll 4
      for (int q = q0; q < q0 + 16 && q < n; q++) acc += Wt[r * n + q] * yprev[q];
    part[pq][r] = acc;
    __syncthreads();
    if (tid < 64) {
      const int g = c * n + tid;
      const double y = (tid < n && g < nshot) ? z[g] : 0.0;
      tvec[tid] = y - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
    }
    __syncthreads();
    acc = 0;
    if (r < n)
#pragma unroll 4
      for (int q = q0; q < q0 + 16 && q < n; q++) acc += Lit[r * n + q] * tvec[q];
    part[pq][r] = acc;
    __syncthreads();
    if (tid < 64) {
      const double x = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
      yprev[tid] = x;
      const int g = c * n + tid;
      if (tid < n && g < nshot) z[g] = x;
    }
    __syncthreads();
  }
  for (int cm = tid; cm < (d.gen ? 0 : d.NC); cm += 256) {
    const double *Bi = d.Binv + 36 * (long)d.S + 9 * cm, *rr = rin + d.cam0 + 3 * cm;
    for (int i = 0; i < 3; i++) z[d.cam0 + 3 * cm + i] = Bi[3 * i] * rr[0] + Bi[3 * i + 1] * rr[1] + Bi[3 * i + 2] * rr[2];
  }
}

// ---- Schur mat-vec ------------------------------------------------------------------------
__global__ void scale_vec_kernel(const double *sc, const double *x, double *y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = sc[i] * x[i];
}

// pass A of the mat-vec (mode 0), cooperative form: one workgroup owns a run of whole points with
// <= kCoopObs observations.  Thread-per-observation reads its Jacobian blocks (SoA: coalesced) and
// forms t_o and Jp^T t_o; thread-per-point sums its observations from LDS in a fixed order
// (deterministic) and applies Hhat; thread-per-observation finishes w_o.  A single point with more
// observations than the tile takes the strided path at the bottom.
// MODE 0: w_o = t_o - Jp_o Hhat sum(Jp^T t),  t_o = Jc_o y_s + Jk_o y_k     (mat-vec)
// MODE 1: w_o = -Jp_o Hhat (-g_p)                                           (right-hand side)
// MODE 2: d_pt = Hhat(-g_p - sum(Jp^T t))                                   (back-substitution)
template <int MODE>
__global__ void __launch_bounds__(kCoopObs) schur_point_coop_kernel(Dev d, const double *y) {
  __shared__ double gsum[kCoopObs * 3];
  __shared__ double vpt[kCoopObs * 3];
  const int tid = threadIdx.x;
  const int p0 = d.wg_pt[blockIdx.x], p1 = d.wg_pt[blockIdx.x + 1];
  const long o0 = d.pt_off[p0], o1 = d.pt_off[p1];
  const int nobs = (int)(o1 - o0);
  // t_o = Jc_o y_s + Jk_o y_k from the row's components, in the order the 26-component rows of rounds 1-5 were multiplied: the rotation columns
  // (Jr), the translation columns (-Jp: (-a) b added = a b subtracted, the same bits), then Jk (row_jk)
  auto row_t = [&](long o, const double (&jp)[6], double &t0, double &t1) {
    const int s = d.o_shot[o];
    const double *ys = y + 6 * (long)s, *yk = y + d.cam0 + 3 * d.shot_camera[s];
    double jk[6];
    row_jk(d, o, s, jk);
    t0 = 0;
    t1 = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double yj = ys[j];
      t0 += JL(o, R_JR + j) * yj;
      t1 += JL(o, R_JR + 3 + j) * yj;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double yj = ys[3 + j];
      t0 += (-jp[j]) * yj;
      t1 += (-jp[3 + j]) * yj;
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double yj = yk[j];
      t0 += jk[j] * yj;
      t1 += jk[3 + j] * yj;
    }
  };
  if (nobs <= kCoopObs) {  // uniform per workgroup
    double t0 = 0, t1 = 0, jp[6] = {0, 0, 0, 0, 0, 0};
    int pl = 0;
    const long o = o0 + tid;
    if (tid < nobs) {
      pl = d.o_point[o] - p0;
#pragma unroll
      for (int j = 0; j < 6; j++) jp[j] = JL(o, R_JP + j);
      if (MODE != 1) {
        row_t(o, jp, t0, t1);
#pragma unroll
        for (int j = 0; j < 3; j++) gsum[3 * tid + j] = jp[j] * t0 + jp[3 + j] * t1;
      }
    }
    __syncthreads();
    if (tid < p1 - p0) {
      const int p = p0 + tid;
      double u[3] = {0, 0, 0};
      if (MODE != 1) {
        const int a = (int)(d.pt_off[p] - o0), b = (int)(d.pt_off[p + 1] - o0);
        for (int k = a; k < b; k++) {
          u[0] += gsum[3 * k];
          u[1] += gsum[3 * k + 1];
          u[2] += gsum[3 * k + 2];
        }
      }
      if (MODE == 1)
        for (int j = 0; j < 3; j++) u[j] = -d.g_pt[3 * (long)p + j];
      if (MODE == 2)
        for (int j = 0; j < 3; j++) u[j] = -d.g_pt[3 * (long)p + j] - u[j];
      const double *Hh = d.Hhat + 6 * (long)p;
      const double v0 = Hh[0] * u[0] + Hh[1] * u[1] + Hh[2] * u[2];
      const double v1 = Hh[1] * u[0] + Hh[3] * u[1] + Hh[4] * u[2];
      const double v2 = Hh[2] * u[0] + Hh[4] * u[1] + Hh[5] * u[2];
      if (MODE == 2) {
        d.d_pt[3 * (long)p] = v0;
        d.d_pt[3 * (long)p + 1] = v1;
        d.d_pt[3 * (long)p + 2] = v2;
      }
      vpt[3 * tid] = v0;
      vpt[3 * tid + 1] = v1;
      vpt[3 * tid + 2] = v2;
    }
    __syncthreads();
    if (MODE == 2) {  // the observation part of the model cost change -m^T (r + m / 2), m = J delta, while Jp and Jc delta_c are at hand
      double acc[1] = {0.0};
      if (tid < nobs) {
        const double v0 = vpt[3 * pl], v1 = vpt[3 * pl + 1], v2 = vpt[3 * pl + 2];
        const double m0 = (jp[0] * v0 + jp[1] * v1 + jp[2] * v2) + t0, m1 = (jp[3] * v0 + jp[4] * v1 + jp[5] * v2) + t1;
        acc[0] = -(m0 * (JA(o, 0) + 0.5 * m0) + m1 * (JA(o, 1) + 0.5 * m1));
      }
      __syncthreads();
      block_sum<1>(acc, gsum);
      if (tid == 0) d.partial[blockIdx.x] = acc[0];
      return;
    }
    if (tid < nobs) {
      const double v0 = vpt[3 * pl], v1 = vpt[3 * pl + 1], v2 = vpt[3 * pl + 2];
      st_stream2<2>(d.w + 2 * o, t0 - (jp[0] * v0 + jp[1] * v1 + jp[2] * v2), t1 - (jp[3] * v0 + jp[4] * v1 + jp[5] * v2));
    }
    return;
  }
  // one long track: strided over the workgroup, fixed-order tree reduction
  double u[3] = {0, 0, 0};
  if (MODE != 1) {
    for (long o = o0 + tid; o < o1; o += kCoopObs) {
      double jp[6], t0, t1;
      for (int j = 0; j < 6; j++) jp[j] = JA(o, R_JP + j);
      row_t(o, jp, t0, t1);
      if (MODE == 0) {
        d.w[2 * o] = t0;
        d.w[2 * o + 1] = t1;
      }
      for (int j = 0; j < 3; j++) u[j] += jp[j] * t0 + jp[3 + j] * t1;
    }
  }
  for (int j = 0; j < 3; j++) gsum[3 * tid + j] = u[j];
  __syncthreads();
  for (int h = kCoopObs / 2; h >= 1; h >>= 1) {
    if (tid < h)
      for (int j = 0; j < 3; j++) gsum[3 * tid + j] += gsum[3 * (tid + h) + j];
    __syncthreads();
  }
  const double *Hh = d.Hhat + 6 * (long)p0;
  double u0 = gsum[0], u1 = gsum[1], u2 = gsum[2];
  if (MODE == 1) { u0 = -d.g_pt[3 * (long)p0]; u1 = -d.g_pt[3 * (long)p0 + 1]; u2 = -d.g_pt[3 * (long)p0 + 2]; }
  if (MODE == 2) { u0 = -d.g_pt[3 * (long)p0] - u0; u1 = -d.g_pt[3 * (long)p0 + 1] - u1; u2 = -d.g_pt[3 * (long)p0 + 2] - u2; }
  const double v0 = Hh[0] * u0 + Hh[1] * u1 + Hh[2] * u2;
  const double v1 = Hh[1] * u0 + Hh[3] * u1 + Hh[4] * u2;
  const double v2 = Hh[2] * u0 + Hh[4] * u1 + Hh[5] * u2;
  if (MODE == 2) {
    if (tid == 0) {
      d.d_pt[3 * (long)p0] = v0;
      d.d_pt[3 * (long)p0 + 1] = v1;
      d.d_pt[3 * (long)p0 + 2] = v2;
    }
    double acc[1] = {0.0};
    for (long o = o0 + tid; o < o1; o += kCoopObs) {
      double jp[6], t0, t1;
      for (int j = 0; j < 6; j++) jp[j] = JA(o, R_JP + j);
      row_t(o, jp, t0, t1);
      const double m0 = (jp[0] * v0 + jp[1] * v1 + jp[2] * v2) + t0, m1 = (jp[3] * v0 + jp[4] * v1 + jp[5] * v2) + t1;
      acc[0] += -(m0 * (JA(o, 0) + 0.5 * m0) + m1 * (JA(o, 1) + 0.5 * m1));
    }
    __syncthreads();
    block_sum<1>(acc, gsum);
    if (tid == 0) d.partial[blockIdx.x] = acc[0];
    return;
  }
  for (long o = o0 + tid; o < o1; o += kCoopObs) {
    const double m0 = JA(o, R_JP) * v0 + JA(o, R_JP + 1) * v1 + JA(o, R_JP + 2) * v2;
    const double m1 = JA(o, R_JP + 3) * v0 + JA(o, R_JP + 4) * v1 + JA(o, R_JP + 5) * v2;
    if (MODE == 0) {
      d.w[2 * o] -= m0;
      d.w[2 * o + 1] -= m1;
    } else {
      d.w[2 * o] = -m0;
      d.w[2 * o + 1] = -m1;
    }
  }
}

// pass B, wavefront per shot: zc_s = sum Jc^T w ; camera partials
template <int W>
__global__ void __launch_bounds__(64 * W) schur_shot_kernel(Dev d) {
  __shared__ double red[W > 1 ? W * 9 : 1];
  const int s = (int)xcd_contiguous(blockIdx.x, gridDim.x), lane = threadIdx.x;
  const ShotFrame f = shot_frame(d, s);
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (long k = d.shot_off[s] + lane; k < d.shot_off[s + 1]; k += 64 * W) {
    const long o = d.shot_obs[k];  // w lives in point-major order: 2 gathered doubles per observation
    const double w0 = d.w[2 * o], w1 = d.w[2 * o + 1];
    double rr[2], jp[6], jc[12], jk[6];  // (the rows recomputed: sm_row; the shot-major copy this kernel read until round 5 is gone)
    sm_row(d, f, k, d.sm_wt[k], rr, jp, jc, jk);
#pragma unroll
    for (int j = 0; j < 6; j++) v[j] += jc[j] * w0 + jc[6 + j] * w1;
#pragma unroll
    for (int j = 0; j < 3; j++) v[6 + j] += jk[j] * w0 + jk[3 + j] * w1;
  }
  shot_sum<9, W>(v, red);
  if (lane == 0) {
    for (int j = 0; j < 6; j++) d.zc[6 * s + j] = v[j];
    for (int j = 0; j < 3; j++) d.part[9 * (long)s + j] = v[6 + j];
  }
}

// finish: out = sc*(zc + prior_diag*y) + (D/radius)*x   (mode 0)
//      or out = sc*(-g + zc)                          (mode 1, rhs)
// Grid: nblk(cam0) workgroups for the shot rows, then ONE PER CAMERA for its three rows: the per-shot partials part[9 s + k] (pass B) are summed
// over the camera's shots here, in a fixed order (until round 5 a launch of cam_reduce_kernel in front of this one; inert: every camera
// constant, the sums are zero).  dot_part (mode 0, optional): the workgroup's share of x . out -- PCG's p . Ap without a launch of its own
// (pcg_step1_kernel adds the shares in workgroup order).
__global__ void __launch_bounds__(TPB) schur_finish_kernel(Dev d, const double *x, const double *y, double *out, double radius, int mode, int inert, double *dot_part) {
  __shared__ double lds[16];
  const int nshot_blocks = (d.cam0 + TPB - 1) / TPB;
  double dot[1] = {0.0};
  if ((int)blockIdx.x < nshot_blocks) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i < d.cam0) {
      double zc = d.zc[i];
      if (mode == 0 && d.prior_rot && (i % 6) < 3) {  // up-vector prior: dense 3x3 on the rotation
        const int s = i / 6, k = i % 6;
        const double *pr = d.prior_rot + 6 * (long)s, *yr = y + 6 * s;
        const int ix[3][3] = {{0, 1, 3}, {1, 2, 4}, {3, 4, 5}};
        zc += pr[ix[k][0]] * yr[0] + pr[ix[k][1]] * yr[1] + pr[ix[k][2]] * yr[2];
      }
      double o;
      if (mode == 0) o = d.sc_red[i] * (zc + d.prior_diag[i] * y[i]) + d.D_red[i] / radius * x[i];
      else o = d.sc_red[i] * (-d.g_red[i] + zc);
      out[i] = o;
      if (mode == 0 && dot_part) dot[0] = x[i] * o;
    }
  } else {
    const int c = blockIdx.x - nshot_blocks;
    double v[3] = {0.0, 0.0, 0.0};
    if (!inert)
      for (int s = threadIdx.x; s < d.S; s += TPB)
        if (d.shot_camera[s] == c)
#pragma unroll
          for (int k = 0; k < 3; k++) v[k] += d.part[9 * (long)s + k];
    block_sum<3>(v, lds);
    __syncthreads();
    if (threadIdx.x == 0)
      for (int k = 0; k < 3; k++) {
        const int i = d.cam0 + 3 * c + k;
        double o;
        if (mode == 0) o = d.sc_red[i] * (v[k] + d.prior_diag[i] * y[i]) + d.D_red[i] / radius * x[i];
        else o = d.sc_red[i] * (-d.g_red[i] + v[k]);
        out[i] = o;
        if (mode == 0 && dot_part) dot[0] += x[i] * o;
      }
  }
  if (mode == 0 && dot_part) {
    block_sum<1>(dot, lds);
    if (threadIdx.x == 0) dot_part[blockIdx.x] = dot[0];
  }
}

// ---- PCG vector kernels (single block; device-resident scalars) --------------------------------
// scal: [0] rz  [1] pAp  [2] rz_new  [3] rr  [4] bb  [8..] misc reductions
__global__ void dot2_kernel(const double *a, const double *b, const double *c, const double *e, int n, double *o0, double *o1) {
  __shared__ double lds[32];
  double v[2] = {0, 0};
#pragma unroll 4
  for (int i = threadIdx.x; i < n; i += blockDim.x) {  // (unrolled: the loads of four steps in flight, the additions in order)
    v[0] += a[i] * b[i];
    if (c) v[1] += c[i] * e[i];
  }
  block_sum<2>(v, lds);
  if (threadIdx.x == 0) {
    *o0 = v[0];
    if (c) *o1 = v[1];
  }
}
// start of PCG in one launch: x = 0, r = b, p = z, and the two inner products dot2_kernel(r, z, b, b) would return (same order)
// (round 6: and y = sc p, the scaled direction the first mat-vec multiplies -- scale_vec_kernel's launch in front of every mat-vec is gone)
__global__ void pcg_init_kernel(const double *b, const double *z, double *x, double *r, double *p, int n, double *o_rz, double *o_bb, const double *sc, double *y) {
  __shared__ double lds[32];
  double v[2] = {0, 0};
#pragma unroll 4
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double bi = b[i], zi = z[i];
    x[i] = 0.0;
    r[i] = bi;
    p[i] = zi;
    y[i] = sc[i] * zi;
    v[0] += bi * zi;
    v[1] += bi * bi;
  }
  block_sum<2>(v, lds);
  if (threadIdx.x == 0) {
    *o_rz = v[0];
    *o_bb = v[1];
  }
}
// x += alpha p, r -= alpha Ap with alpha = rz / pAp, pAp = the sum of the mat-vec's shares dot_part[0 .. nparts) in workgroup order (every
// workgroup adds them the same way: the same alpha everywhere; dot_part == nullptr: pAp is in *o_pAp already).  Also leaves the block's share
// of r.r in rr_part[block] (summed by the host in block order when it polls: deterministic).
// y_out (if any) = sc x for the back-substitution that follows the last iteration (scale_vec_kernel's launch)
__global__ void __launch_bounds__(TPB) pcg_step1_kernel(double *x, double *r, const double *p, const double *Ap, int n, const double *rz, const double *dot_part, int nparts,
                                                         double *o_pAp, double *rr_part, const double *sc = nullptr, double *y_out = nullptr) {
  __shared__ double lds[32];
  __shared__ double s_alpha;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  {
    double pAp;
    if (dot_part) {
      double a[1] = {0.0};
      for (int q = threadIdx.x; q < nparts; q += TPB) a[0] += dot_part[q];
      block_sum<1>(a, lds);
      pAp = a[0];
    } else
      pAp = *o_pAp;
    if (threadIdx.x == 0) {
      s_alpha = pAp != 0.0 ? *rz / pAp : 0.0;
      if (blockIdx.x == 0 && dot_part) *o_pAp = pAp;
    }
    __syncthreads();
  }
  double v[1] = {0.0};
  if (i < n) {
    const double alpha = s_alpha;
    const double xi = x[i] + alpha * p[i];
    x[i] = xi;
    if (y_out) y_out[i] = sc[i] * xi;
    const double ri = r[i] - alpha * Ap[i];
    r[i] = ri;
    v[0] = ri * ri;
  }
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) rr_part[blockIdx.x] = v[0];
}
__global__ void precond_apply_kernel(Dev d, const double *r, double *z) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < d.S) {
    const double *Bi = d.Binv + 36 * (long)b, *rr = r + 6 * b;
    for (int i = 0; i < 6; i++) {
      double s = 0;
      for (int j = 0; j < 6; j++) s += Bi[6 * i + j] * rr[j];
      z[6 * b + i] = s;
    }
  } else if (b < d.S + d.NC) {
    const int c = b - d.S;
    const double *Bi = d.Binv + 36 * (long)d.S + 9 * c, *rr = r + d.cam0 + 3 * c;
    for (int i = 0; i < 3; i++) z[d.cam0 + 3 * c + i] = Bi[3 * i] * rr[0] + Bi[3 * i + 1] * rr[1] + Bi[3 * i + 2] * rr[2];
  }
}
// p = z + beta p with beta = rz_new / rz_old (the two live in alternating slots: no launch to move one into the other), and y = sc p for the
// next mat-vec
__global__ void pcg_step2_kernel(double *p, const double *z, int n, const double *rz_old, const double *rz_new, const double *sc, double *y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double beta = *rz_old != 0.0 ? *rz_new / *rz_old : 0.0;
  const double pi = z[i] + beta * p[i];
  p[i] = pi;
  y[i] = sc[i] * pi;
}

// prior part of the model change, and the candidate point x + delta with its norms
// out: [0] += prior model change ; [1] step^2 ; [2] x^2  (over variable blocks only)
// (round 6: out[0] starts from the sum of the back-substitution's npart workgroup shares of the observations' model change, partial[0 .. npart)
// in workgroup order -- finish_reduce_kernel's launch in front of this one is gone)
// (later in round 6: the points' shares of the step's norms -- candidate_points_kernel's part2[0 .. 2 npart2), launched in FRONT of this kernel now -- are
// summed here too, out2[c] = sum_i part2[2 i + c] as finish_reduce_kernel formed them, and the candidate's rotation blocks are written with its poses:
// finish_reduce_kernel's and shot_rot_kernel's launches behind this one are gone)
__global__ void __launch_bounds__(1024) candidate_kernel(Dev d, const double *y, const double *partial, long npart, double *out, const double *part2, long npart2,
                                                         double *out2, int do_rot) {
  __shared__ double lds[32];
  {
    double m[1] = {0.0};
    for (long i = threadIdx.x; i < npart; i += blockDim.x) m[0] += partial[i];
    block_sum<1>(m, lds);
    if (threadIdx.x == 0) out[0] = m[0];
    __syncthreads();
  }
  for (int c = 0; c < 2; c++) {
    double m[1] = {0.0};
    for (long i = threadIdx.x; i < npart2; i += blockDim.x) m[0] += part2[2 * i + c];
    block_sum<1>(m, lds);
    if (threadIdx.x == 0) out2[c] = m[0];
    __syncthreads();
  }
  double v[3] = {0, 0, 0};
  for (int c = threadIdx.x; c < d.NC; c += blockDim.x) {
    const double *q = d.cams + 3 * c;
    const bool fixed = d.cam_fixed[c];
    if (!fixed) {
      const double *pr = d.cam_prior + 3 * c, *sg = d.cam_sigma + 3 * c;
      const double wq[3] = {1.0 / fmax(sg[0], kEps), 1.0 / fmax(sg[1], kEps), 1.0 / fmax(sg[2], kEps)};
      const double e[3] = {(q[0] - pr[0]) * wq[0], (q[1] - pr[1]) * wq[1], log(q[2] / pr[2]) * wq[2]};
      const double j[3] = {wq[0], wq[1], wq[2] / q[2]};
      for (int k = 0; k < 3; k++) {
        const double m = j[k] * y[d.cam0 + 3 * c + k];
        v[0] -= m * (e[k] + 0.5 * m);
      }
    }
    for (int k = 0; k < 3; k++) {
      const double dl = fixed ? 0.0 : y[d.cam0 + 3 * c + k];
      d.cams_n[3 * c + k] = q[k] + dl;
      if (!fixed) {
        v[1] += dl * dl;
        v[2] += q[k] * q[k];
      }
    }
  }
  for (int s = threadIdx.x; s < d.S; s += blockDim.x) {
    const bool fixed = d.shot_fixed && d.shot_fixed[s];
    // the shot's step, pose and prior requested together (a load behind every test made this kernel six round trips per shot: 61 us)
    double ys[6], ps[6], g3[3] = {0, 0, 0}, sg = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      ys[k] = y[6 * s + k];
      ps[k] = d.poses[6 * s + k];
    }
    if (d.gps && d.gps_sigma) {
      sg = d.gps_sigma[s];
#pragma unroll
      for (int k = 0; k < 3; k++) g3[k] = d.gps[3 * s + k];
    }
    if (!fixed && sg > 0) {
      const double wq = 1.0 / sg;
      for (int k = 0; k < 3; k++) {
        const double m = wq * ys[3 + k];
        const double e = wq * (ps[3 + k] - g3[k]);
        v[0] -= m * (e + 0.5 * m);
      }
    }
    if (!fixed && d.up && d.up_sigma && d.up_sigma[s] > 0) {
      const double *ru = d.up_r + 3 * s, *Ju = d.up_J + 9 * s;
      for (int i = 0; i < 3; i++) {
        const double m = Ju[3 * i] * y[6 * s] + Ju[3 * i + 1] * y[6 * s + 1] + Ju[3 * i + 2] * y[6 * s + 2];
        v[0] -= m * (ru[i] + 0.5 * m);
      }
    }
    double pn[6];
    for (int k = 0; k < 6; k++) {
      const double dl = fixed ? 0.0 : ys[k];
      pn[k] = ps[k] + dl;
      d.poses_n[6 * s + k] = pn[k];
      if (!fixed) {
        v[1] += dl * dl;
        v[2] += ps[k] * ps[k];
      }
    }
    // shot_rot_kernel's work for the candidate (up_r / up_J above are arrays of the old point): while a thread has one shot -- 5 000 shots' blocks by
    // one workgroup were 67 us at configs[4] against the 6 us of the launch
    if (do_rot) rot_and_derivs(pn, d.shotR + 36 * (long)s, d.shotR + 36 * (long)s + 9);
  }
  block_sum<3>(v, lds);
  if (threadIdx.x == 0) {
    out[0] += v[0];
    out[1] = v[1];
    out[2] = v[2];
  }
}
__global__ void __launch_bounds__(TPB) candidate_points_kernel(Dev d, double *part) {
  __shared__ double lds[16];
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  double v[2] = {0, 0};
  if (i < 3L * d.P) {
    const bool fixed = d.point_fixed && d.point_fixed[i / 3];
    const double dl = fixed ? 0.0 : d.d_pt[i];
    d.pts_n[i] = d.pts[i] + dl;
    if (!fixed) {
      v[0] = dl * dl;
      v[1] = d.pts[i] * d.pts[i];
    }
  }
  block_sum<2>(v, lds);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = v[0];
    part[2 * blockIdx.x + 1] = v[1];
  }
}
__global__ void absmax_kernel(const double *a, long n, const double *b, long m, double *out) {
  // max |.| over two arrays; *out must be zeroed first.  Non-negative doubles order like their bit
  // patterns, so an integer atomicMax gives the exact (order independent) maximum.
  __shared__ double lds[32];
  double v = 0;
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long i = t0; i < n; i += stride) v = fmax(v, fabs(a[i]));
  for (long i = t0; i < m; i += stride) v = fmax(v, fabs(b[i]));
  for (int k = 32; k >= 1; k >>= 1) v = fmax(v, __shfl_xor(v, k));
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (unsigned w = 1; w < (blockDim.x + 63) / 64; w++) v = fmax(v, lds[w]);
    atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(v));
  }
}
// lm_diag_kernel and absmax_kernel(g_red, g_pt) in one launch (they read and write different arrays); *out must be zeroed first
__global__ void __launch_bounds__(256) lm_diag_absmax_kernel(Dev d, double *out) {
  __shared__ double lds[32];
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = d.nred, m = 3L * d.P;
  double v = 0;
  for (long i = t0; i < n; i += stride) {
    d.D_red[i] = fmin(fmax(d.diag_red[i] * d.sc_red[i] * d.sc_red[i], 1e-6), 1e32);
    v = fmax(v, fabs(d.g_red[i]));
  }
  for (long i = t0; i < m; i += stride) {
    const long p = i / 3;
    const int j = (int)(i - 3 * p);
    const int dg[3] = {0, 3, 5};
    d.D_pt[i] = fmin(fmax(d.Hpp[6 * p + dg[j]] * d.sc_pt[i] * d.sc_pt[i], 1e-6), 1e32);
    v = fmax(v, fabs(d.g_pt[i]));
  }
  for (int k = 32; k >= 1; k >>= 1) v = fmax(v, __shfl_xor(v, k));
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (unsigned w = 1; w < (blockDim.x + 63) / 64; w++) v = fmax(v, lds[w]);
    atomicMax((unsigned long long *)out, (unsigned long long)__double_as_longlong(v));
  }
}
__global__ void reproj_kernel(Dev d, double *out) {
  const long o = (long)blockIdx.x * TPB + threadIdx.x;
  if (o >= d.M) return;
  const int s = d.o_shot[o], p = d.o_point[o];
  const double *R = d.shotR + 36 * (long)s;
  double r[2];
  const int cmodel = d.cam_model ? d.cam_model[d.shot_camera[s]] : 0;
  project_obs<false>(cmodel, d.pts + 3 * (long)p, d.poses + 6 * (long)s, R, R + 9,
                     cmodel >= 2 ? d.cam_ext + 16 * d.shot_camera[s] : d.cams + 3 * d.shot_camera[s], d.o_x[o], d.o_y[o],
                     1.0, r, nullptr, nullptr, nullptr);
  out[2 * o] = r[0];
  out[2 * o + 1] = r[1];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// ---- exact elimination of the camera border of the preconditioner -----------------------------------
// With the exact band A (shot block of the reduced system) the only thing block Jacobi on the camera rows
// leaves to CG is the coupling B to the few shared intrinsics.  For nb = 3 * cameras <= 6 border unknowns the
// border is eliminated exactly instead:  W = A^-1 B (nb cyclic-reduction solves),  Sigma = C - B^T W,
//   z_s = A^-1 r_s,   z_c = Sigma^-1 (r_c - B^T z_s),   z_s -= W z_c,
// which makes the preconditioner the reduced matrix itself: CG converges in one or two iterations.
__global__ void unit_vec_kernel(double *x, int n, int j) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = i == j ? 1.0 : 0.0;
}
// out[i * nb + j] = Bc_i . W_j   (one block per entry)
__global__ void border_dots_kernel(const double *Bc, const double *W, int nb, int n, double *out) {
  __shared__ double lds[32];
  const int i = blockIdx.x / nb, j = blockIdx.x % nb;
  double v[1] = {0.0};
#pragma unroll 4
  for (int t = threadIdx.x; t < n; t += blockDim.x) v[0] += Bc[(long)i * n + t] * W[(long)j * n + t];
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) out[blockIdx.x] = v[0];
}
// z_c = SigInv (r_c - B^T z_s)   (single block)
__global__ void border_rhs_kernel(const double *Bc, const double *SigInv, const double *r, double *z, int nb, int n, int cam0) {
  __shared__ double lds[32];
  __shared__ double y[8];
  for (int i = 0; i < nb; i++) {  // (all rows in one pass over z_s with six accumulators was measured in round 4: 37 against 33 us)
    double v[1] = {0.0};
    for (int t = threadIdx.x; t < n; t += blockDim.x) v[0] += Bc[(long)i * n + t] * z[t];
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) y[i] = r[cam0 + i] - v[0];
    __syncthreads();
  }
  if (threadIdx.x < nb) {
    double acc = 0.0;
    for (int j = 0; j < nb; j++) acc += SigInv[threadIdx.x * nb + j] * y[j];
    z[cam0 + threadIdx.x] = acc;
  }
}
// ... the same in two launches for long rows (round 6): row i of B^T z_s by workgroup i -- the same threads add the same entries in the same
// order as in the single block, so y has the same bits --, then z_c = SigInv y.  The single block walked its nb rows one after the other:
// 129 us for the sixteen border rows of a Brown camera with a GPS bias at configs[4] (6 S = 30 000 entries each), 12 us this way.
__global__ void __launch_bounds__(1024) border_rhs_dots_kernel(const double *Bc, const double *r, const double *z, double *y, int n, int cam0) {
  __shared__ double lds[32];
  const int i = blockIdx.x;
  double v[1] = {0.0};
  for (int t = threadIdx.x; t < n; t += blockDim.x) v[0] += Bc[(long)i * n + t] * z[t];
  block_sum<1>(v, lds);
  if (threadIdx.x == 0) y[i] = r[cam0 + i] - v[0];
}
__global__ void border_rhs_apply_kernel(const double *SigInv, const double *y, double *z, int nb, int cam0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  double acc = 0.0;
  for (int j = 0; j < nb; j++) acc += SigInv[i * nb + j] * y[j];
  z[cam0 + i] = acc;
}
// z_s -= W z_c
__global__ void border_update_kernel(const double *W, double *z, int nb, int n, int cam0) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  double acc = 0.0;
  for (int j = 0; j < nb; j++) acc += W[(long)j * n + t] * z[cam0 + j];
  z[t] -= acc;
}


// ---- exact camera border: every column of B = S [0; I] in one pass ------------------------------------------------------------
// The columns of the reduced matrix that belong to the camera unknowns used to come from 3 NC mat-vecs with unit vectors (two
// streaming passes over all observations each).  A unit vector has no shot part and touches an observation through its own
// camera's three columns only, so all NB = 3 NC columns are formed together: pass A reads Jp and Jk once (t_o is 2 x 3 on the
// observation's camera columns, u_p = sum Jp^T t and v_p = Hhat u_p are 3 x NB per point, w_o = t_o - Jp v_p is 2 x NB), pass B reads
// Jc and Jk once.  Same summation orders as schur_point_coop_kernel<0> / schur_shot_kernel column by column.
// Sigma^-1 = (C - B^T W)^-1 by Gauss-Jordan with partial pivoting (nb <= 6), one lane; *status = 1 when Sigma is singular / not finite
__global__ void border_sigma_kernel(const double *Cm, const double *dots, double *SigInv, int nb, int *status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double Sg[36], Inv[36];
  for (int i = 0; i < nb * nb; i++) {
    Sg[i] = Cm[i] - dots[i];
    Inv[i] = 0.0;
  }
  for (int i = 0; i < nb; i++) Inv[i * nb + i] = 1.0;
  int bad = 0;
  for (int c = 0; c < nb && !bad; c++) {
    int piv = c;
    for (int r2 = c + 1; r2 < nb; r2++)
      if (fabs(Sg[r2 * nb + c]) > fabs(Sg[piv * nb + c])) piv = r2;
    const double pv = Sg[piv * nb + c];
    if (!(fabs(pv) > 0) || !(fabs(pv) < 1.0 / 0.0)) { bad = 1; break; }
    for (int q = 0; q < nb; q++) {
      const double t0 = Sg[c * nb + q], t1 = Inv[c * nb + q];
      Sg[c * nb + q] = Sg[piv * nb + q];
      Sg[piv * nb + q] = t0;
      Inv[c * nb + q] = Inv[piv * nb + q];
      Inv[piv * nb + q] = t1;
    }
    const double ip = 1.0 / Sg[c * nb + c];
    for (int q = 0; q < nb; q++) { Sg[c * nb + q] *= ip; Inv[c * nb + q] *= ip; }
    for (int r2 = 0; r2 < nb; r2++) {
      if (r2 == c) continue;
      const double f = Sg[r2 * nb + c];
      for (int q = 0; q < nb; q++) { Sg[r2 * nb + q] -= f * Sg[c * nb + q]; Inv[r2 * nb + q] -= f * Inv[c * nb + q]; }
    }
  }
  *status = bad;
  for (int i = 0; i < nb * nb; i++) SigInv[i] = bad ? 0.0 : Inv[i];
}
template <int NB>
__global__ void __launch_bounds__(kCoopObs) border_point_kernel(Dev d, double *wB) {
  constexpr int NC_ = NB / 3;
  __shared__ double gsum[kCoopObs * 9];
  __shared__ int gcam[kCoopObs];
  __shared__ double vpt[kCoopObs * 3 * NB];
  const int tid = threadIdx.x;
  const int p0 = d.wg_pt[blockIdx.x], p1 = d.wg_pt[blockIdx.x + 1];
  const long o0 = d.pt_off[p0], o1 = d.pt_off[p1];
  const int nobs = (int)(o1 - o0);
  if (nobs <= kCoopObs) {
    double t0[3] = {0, 0, 0}, t1[3] = {0, 0, 0}, jp[6] = {0, 0, 0, 0, 0, 0};
    int pl = 0, cam = 0;
    const long o = o0 + tid;
    if (tid < nobs) {
      pl = d.o_point[o] - p0;
      const int so = d.o_shot[o];
      cam = d.shot_camera[so];
      const double *sck = d.sc_red + d.cam0 + 3 * cam;
      double jk[6];
      row_jk(d, o, so, jk);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        t0[k] = jk[k] * sck[k];
        t1[k] = jk[3 + k] * sck[k];
      }
#pragma unroll
      for (int j = 0; j < 6; j++) jp[j] = JL(o, R_JP + j);
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int j = 0; j < 3; j++) gsum[9 * tid + 3 * k + j] = jp[j] * t0[k] + jp[3 + j] * t1[k];
      gcam[tid] = cam;
    }
    __syncthreads();
    if (tid < p1 - p0) {
      const int p = p0 + tid;
      double u[NB][3];
#pragma unroll
      for (int c = 0; c < NB; c++) u[c][0] = u[c][1] = u[c][2] = 0.0;
      const int a = (int)(d.pt_off[p] - o0), b = (int)(d.pt_off[p + 1] - o0);
      for (int k = a; k < b; k++) {
        const int cc = gcam[k];
#pragma unroll
        for (int c = 0; c < NC_; c++)
          if (c == cc)
#pragma unroll
            for (int q = 0; q < 3; q++)
#pragma unroll
              for (int j = 0; j < 3; j++) u[3 * c + q][j] += gsum[9 * k + 3 * q + j];
      }
      const double *Hh = d.Hhat + 6 * (long)p;
#pragma unroll
      for (int c = 0; c < NB; c++) {
        vpt[(3 * tid + 0) * NB + c] = Hh[0] * u[c][0] + Hh[1] * u[c][1] + Hh[2] * u[c][2];
        vpt[(3 * tid + 1) * NB + c] = Hh[1] * u[c][0] + Hh[3] * u[c][1] + Hh[4] * u[c][2];
        vpt[(3 * tid + 2) * NB + c] = Hh[2] * u[c][0] + Hh[4] * u[c][1] + Hh[5] * u[c][2];
      }
    }
    __syncthreads();
    if (tid < nobs) {
      double2 *dst = reinterpret_cast<double2 *>(wB + 2L * NB * o);
#pragma unroll
      for (int c = 0; c < NB; c++) {
        const double v0 = vpt[(3 * pl + 0) * NB + c], v1 = vpt[(3 * pl + 1) * NB + c], v2 = vpt[(3 * pl + 2) * NB + c];
        const bool own = (c / 3) == cam;
        st_stream2<2>(reinterpret_cast<double *>(dst + c), (own ? t0[c % 3] : 0.0) - (jp[0] * v0 + jp[1] * v1 + jp[2] * v2),
                      (own ? t1[c % 3] : 0.0) - (jp[3] * v0 + jp[4] * v1 + jp[5] * v2));
      }
    }
    return;
  }
  // one long track: strided over the workgroup, fixed-order tree reduction (through vpt)
  double u[NB][3];
#pragma unroll
  for (int c = 0; c < NB; c++) u[c][0] = u[c][1] = u[c][2] = 0.0;
  for (long o = o0 + tid; o < o1; o += kCoopObs) {
    const int so = d.o_shot[o], cam = d.shot_camera[so];
    const double *sck = d.sc_red + d.cam0 + 3 * cam;
    double jk[6];
    row_jk(d, o, so, jk);
#pragma unroll
    for (int c = 0; c < NC_; c++)
      if (c == cam)
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const double t0 = jk[k] * sck[k], t1 = jk[3 + k] * sck[k];
#pragma unroll
          for (int j = 0; j < 3; j++) u[3 * c + k][j] += JA(o, R_JP + j) * t0 + JA(o, R_JP + 3 + j) * t1;
        }
  }
#pragma unroll
  for (int c = 0; c < NB; c++)
#pragma unroll
    for (int j = 0; j < 3; j++) vpt[(3 * tid + j) * NB + c] = u[c][j];
  __syncthreads();
  for (int h = kCoopObs / 2; h >= 1; h >>= 1) {
    if (tid < h)
      for (int q = 0; q < 3 * NB; q++) vpt[3 * NB * tid + q] += vpt[3 * NB * (tid + h) + q];
    __syncthreads();
  }
  const double *Hh = d.Hhat + 6 * (long)p0;
  double v[NB][3];
#pragma unroll
  for (int c = 0; c < NB; c++) {
    const double u0 = vpt[0 * NB + c], u1 = vpt[1 * NB + c], u2 = vpt[2 * NB + c];
    v[c][0] = Hh[0] * u0 + Hh[1] * u1 + Hh[2] * u2;
    v[c][1] = Hh[1] * u0 + Hh[3] * u1 + Hh[4] * u2;
    v[c][2] = Hh[2] * u0 + Hh[4] * u1 + Hh[5] * u2;
  }
  for (long o = o0 + tid; o < o1; o += kCoopObs) {
    const int so = d.o_shot[o], cam = d.shot_camera[so];
    const double *sck = d.sc_red + d.cam0 + 3 * cam;
    double jk[6];
    row_jk(d, o, so, jk);
#pragma unroll
    for (int c = 0; c < NB; c++) {
      const bool own = (c / 3) == cam;
      const double t0 = own ? jk[c % 3] * sck[c % 3] : 0.0, t1 = own ? jk[3 + c % 3] * sck[c % 3] : 0.0;
      wB[(2L * NB * o) + 2 * c] = t0 - (JA(o, R_JP) * v[c][0] + JA(o, R_JP + 1) * v[c][1] + JA(o, R_JP + 2) * v[c][2]);
      wB[(2L * NB * o) + 2 * c + 1] = t1 - (JA(o, R_JP + 3) * v[c][0] + JA(o, R_JP + 4) * v[c][1] + JA(o, R_JP + 5) * v[c][2]);
    }
  }
}

// pass B: wavefront per shot.  Bc[c][6 s + j] = sc (sum Jc^T w_c) (the finish of the mat-vec on shot rows: no diagonal terms for a
// camera unit vector), partB[s][3 x NB] = sum Jk^T w_c
template <int NB, int W>
__global__ void __launch_bounds__(64 * W) border_shot_kernel(Dev d, const double *wB, double *Bc, double *partB) {
  __shared__ double red[W > 1 ? W * 9 : 1];
  const int s = (int)xcd_contiguous(blockIdx.x, gridDim.x), lane = threadIdx.x;
  double v[NB][9];
#pragma unroll
  for (int c = 0; c < NB; c++)
#pragma unroll
    for (int i = 0; i < 9; i++) v[c][i] = 0.0;
  const ShotFrame f = shot_frame(d, s);
  for (long k = d.shot_off[s] + lane; k < d.shot_off[s + 1]; k += 64 * W) {
    const long o = d.shot_obs[k];
    const double2 *src = reinterpret_cast<const double2 *>(wB + 2L * NB * o);
    double jc0[6], jc1[6], jk0[3], jk1[3];
    {
      double rr[2], jp[6], jc[12], jk[6];
      sm_row(d, f, k, d.sm_wt[k], rr, jp, jc, jk);
#pragma unroll
      for (int j = 0; j < 6; j++) {
        jc0[j] = jc[j];
        jc1[j] = jc[6 + j];
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        jk0[j] = jk[j];
        jk1[j] = jk[3 + j];
      }
    }
#pragma unroll
    for (int c = 0; c < NB; c++) {
      const double2 wv = src[c];
#pragma unroll
      for (int j = 0; j < 6; j++) v[c][j] += jc0[j] * wv.x + jc1[j] * wv.y;
#pragma unroll
      for (int j = 0; j < 3; j++) v[c][6 + j] += jk0[j] * wv.x + jk1[j] * wv.y;
    }
  }
#pragma unroll
  for (int c = 0; c < NB; c++) {  // (column by column through the same nine slots of LDS)
    shot_sum<9, W>(v[c], red);
    if (W > 1) __syncthreads();
  }
  if (lane == 0) {
    const long n6 = 6L * d.S;
#pragma unroll
    for (int c = 0; c < NB; c++) {
      for (int j = 0; j < 6; j++) Bc[c * n6 + 6 * s + j] = d.sc_red[6 * s + j] * v[c][j];
      for (int j = 0; j < 3; j++) partB[(long)s * 3 * NB + 3 * c + j] = v[c][6 + j];
    }
  }
}

// camera rows of the columns: Cm[(3 cam + k) * NB + c] = sc_i (sum over the camera's shots of partB + prior_diag_i y_i) + D_i / radius [i == c],
// y = sc e_c  (schur_finish_kernel, mode 0, on the camera rows); one block per camera, fixed order
template <int NB>
__global__ void __launch_bounds__(TPB) border_cam_kernel(Dev d, const double *partB, double *Cm, double radius) {
  __shared__ double lds[4 * 3 * NB];
  const int cam = blockIdx.x;
  double v[3 * NB];
#pragma unroll
  for (int i = 0; i < 3 * NB; i++) v[i] = 0.0;
  for (int s = threadIdx.x; s < d.S; s += TPB)
    if (d.shot_camera[s] == cam)
#pragma unroll
      for (int i = 0; i < 3 * NB; i++) v[i] += partB[(long)s * 3 * NB + i];
  block_sum<3 * NB>(v, lds);
  if (threadIdx.x == 0)
    for (int c = 0; c < NB; c++)
      for (int k = 0; k < 3; k++) {
        const int i = d.cam0 + 3 * cam + k;
        const bool diag = (3 * cam + k) == c;
        const double y = diag ? d.sc_red[i] : 0.0;
        Cm[(3 * cam + k) * NB + c] = d.sc_red[i] * (v[3 * c + k] + d.prior_diag[i] * y) + (diag ? d.D_red[i] / radius : 0.0);
      }
}

// ---- setup on the device: the point-major order and the shot-major index lists (two stable radix sorts + lower bounds) ----
__global__ void iota_int_kernel(int *a, long n) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) a[i] = (int)i;
}
__global__ void gather_byte_kernel(const int *idx, const unsigned char *src, long n, unsigned char *dst) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// generic mode: an instance's shot-major segment starts where its first view's does
__global__ void gen_shot_off_kernel(const long *view_off, const int *inst_view0, int S, long *shot_off) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i <= S) shot_off[i] = view_off[inst_view0[i]];
}
__global__ void gather_int_kernel(const int *idx, const int *src, long n, int *dst) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// off[v] = first position k with sorted[k] >= v, v = 0 .. nv (off[nv] = n)
__global__ void lower_bound_kernel(const int *sorted, long n, int nv, long *off) {
  const long v = (long)blockIdx.x * TPB + threadIdx.x;
  if (v > nv) return;
  long lo = 0, hi = n;
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (sorted[mid] < (int)v)
      lo = mid + 1;
    else
      hi = mid;
  }
  off[v] = lo;
}
// largest (max shot - min shot) over the tracks: the block half-bandwidth of the shot-shot coupling
__global__ void track_width_kernel(const long *pt_off, const int *o_shot, int np, int *out) {
  const int p = blockIdx.x * TPB + threadIdx.x;
  if (p >= np) return;
  int mn = 1 << 30, mx = -1;
  for (long k = pt_off[p]; k < pt_off[p + 1]; k++) {
    const int s = o_shot[k];
    mn = min(mn, s);
    mx = max(mx, s);
  }
  if (mx >= 0) atomicMax(out, mx - mn);
}

// first shot of every track (S for a point nobody observes: sorts behind every slab's reach) and whether some track sees a shot twice
__global__ void track_first_kernel(const long *pt_off, const int *o_shot, int np, int S, int *first, int *dup) {
  const int p = blockIdx.x * TPB + threadIdx.x;
  if (p >= np) return;
  int mn = 1 << 30, mx = -1;
  const long a = pt_off[p], b = pt_off[p + 1];
  for (long k = a; k < b; k++) {
    const int s = o_shot[k];
    mn = min(mn, s);
    mx = max(mx, s);
  }
  first[p] = mx >= 0 ? mn : S;
  if (b - a > 64) {  // such a track is wider than any band the window kernel takes, or it repeats a shot
    atomicOr(dup, 1);
    return;
  }
  for (long k = a + 1; k < b; k++)
    for (long k2 = a; k2 < k; k2++)
      if (o_shot[k] == o_shot[k2]) {
        atomicOr(dup, 1);
        return;
      }
}
// the sorted points' tracks as the matrix-core band assembly reads them: first observation, the observation at every shot offset from
// the first shot (tracks of at most 16 shots in a window of 16: checked by the caller), the last offset
__global__ void sorted_tracks_kernel(const int *pts, const int *first, const long *pt_off, const int *o_shot, int np, int *o0, int *last,
                                     unsigned char *pos) {
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k >= np) return;
  const int p = pts[k], a = first[k];
  const long b = pt_off[p], e = pt_off[p + 1];
  o0[k] = (int)b;
  unsigned int w[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  int mx = -1;
  for (long q = b; q < e; q++) {
    const int sh = o_shot[q] - a;
    if (sh >= 0 && sh < 16) {
      w[sh >> 2] = (w[sh >> 2] & ~(0xffu << (8 * (sh & 3)))) | ((unsigned int)(q - b) << (8 * (sh & 3)));
      mx = max(mx, sh);
    }
  }
  last[k] = mx;
  reinterpret_cast<uint4 *>(pos)[k] = make_uint4(w[0], w[1], w[2], w[3]);
}
// off[v] = first position k with sorted[k] >= v * stride, v = 0 .. nv
__global__ void slab_bound_kernel(const int *sorted, int n, int stride, int nv, int *off) {
  const int v = blockIdx.x * TPB + threadIdx.x;
  if (v > nv) return;
  const long key = (long)v * stride;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((long)sorted[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  off[v] = lo;
}

// ---- setup helpers: the observation arrays are permuted on the device (host only builds the index lists) ----
__global__ void gather_pm_kernel(const int *perm, const double *raw_xy, const double *raw_sigma, long M, double *o_x, double *o_y,
                                 double *o_sigma) {
  const long k = (long)blockIdx.x * TPB + threadIdx.x;
  if (k >= M) return;
  const long o = perm[k];
  o_x[k] = raw_xy[2 * o];
  o_y[k] = raw_xy[2 * o + 1];
  o_sigma[k] = raw_sigma[o];
}
__global__ void gather_sm_kernel(const int *shot_obs, const int *o_shot, const int *o_point, const double *o_x, const double *o_y,
                                 const double *o_sigma, long M, int *s_shot, int *s_point, double *s_x, double *s_y, double *s_sigma) {
  const long k = (long)blockIdx.x * TPB + threadIdx.x;
  if (k >= M) return;
  const long o = shot_obs[k];
  s_shot[k] = o_shot[o];
  s_point[k] = o_point[o];
  s_x[k] = o_x[o];
  s_y[k] = o_y[o];
  s_sigma[k] = o_sigma[o];
}
__global__ void unpermute3_kernel(const int *perm, const double *in, long M, double *out) {
  const long k = (long)blockIdx.x * TPB + threadIdx.x;
  if (k >= M) return;
  const long o = perm[k];
  out[3 * o] = in[3 * k];
  out[3 * o + 1] = in[3 * k + 1];
  out[3 * o + 2] = in[3 * k + 2];
}
__global__ void unpermute2_kernel(const int *perm, const double *in, long M, double *out) {
  const long k = (long)blockIdx.x * TPB + threadIdx.x;
  if (k >= M) return;
  const long o = perm[k];
  out[2 * o] = in[2 * k];
  out[2 * o + 1] = in[2 * k + 1];
}


// ---- batched fp64 GEMM on the matrix cores (round 5: replaces rocblas_dgemm_strided_batched in the dense-cluster cyclic reduction) ----
// C = alpha op(A) op(B) + beta C, column-major, `batch` problems a stride apart (grid z).  A workgroup of four wavefronts owns a 64 x 64
// tile of C, a wavefront a 32 x 32 quarter = 2 x 2 accumulators of v_mfma_f64_16x16x4_f64 (lane l feeds A[l % 16][l / 16] and
// B[l / 16][l % 16]; register r of lane l holds C[4 r + l / 16][l % 16]).  K advances 16 at a time through LDS.  A panel is stored in
// the orientation its GLOBAL layout is contiguous in, so that the wavefront's stores are consecutive addresses:
//   contiguous along the tile's rows / columns (A not transposed, B transposed):  [k][i], row stride 80 doubles -- the sixteen lanes of
//     one k read 32 consecutive banks and the next k starts 32 banks on;
//   contiguous along k (A transposed, B not transposed):                          [i][k], row stride 18 doubles -- lane i of an operand
//     read sits 36 banks after lane i - 1 (distinct multiples of four), the next k two banks on.
// (The first version stored every panel [k][i]: the k-contiguous ones went in with a stride of 80 doubles between lanes -- two bank groups
// for 64 lanes -- and the LDS stores, not the matrix pipe, set the pace: 2.2 x rocBLAS' time on the block survey.)  Two LDS buffers and
// two register sets: while the matrix instructions of step s run on one buffer, the panels of step s + 1 are written to the other and
// the global loads of step s + 3 are issued -- one barrier per step; 38 KB of LDS keeps four workgroups on a CU.  Edges (m, n, k not
// multiples of the tile) are zero-filled on the way in and masked on the way out.  The C tile of an updating product (beta != 0) is
// requested before the K loop.
constexpr int kGemmKs = 16, kGemmLdM = 80, kGemmLdK = kGemmKs + 2;
constexpr int kGemmPanel = 64 * kGemmLdK > kGemmKs * kGemmLdM ? 64 * kGemmLdK : kGemmKs * kGemmLdM;  // doubles of one panel in either orientation
constexpr size_t kGemmLds = (size_t)4 * kGemmPanel * sizeof(double);                                  // A and B panels, twice
typedef double gemm_v4d __attribute__((ext_vector_type(4)));
// One launch carries up to three INDEPENDENT products (grid z = the sum of their batches): most launches of this path are small -- a
// level's few clusters x a panel's tiles --, and products that do not depend on each other (G = D^-1 E and H = D^-1 E_r^T; the trailing
// update of a panel and the panel's own column block) fill the chip together instead of one after the other.  skip_j0 .. skip_j1: columns
// of C this product leaves alone (the trailing update of the blocked inversion skips the panel's columns, which its sibling writes).
struct GemmProb {
  int ta, tb, m, n, k, lda, ldb, ldc, batch, skip_j0, skip_j1;
  double alpha, beta;
  const double *A, *B;
  double *C;
  long sa, sb, sc;
};
struct GemmList {
  int np;
  GemmProb p[3];
};
__global__ void __launch_bounds__(256, 4) dgemm_mfma_kernel(GemmList L) {
  extern __shared__ __attribute__((aligned(16))) double gemm_lds[];  // [buffer][A | B][kGemmPanel]
  int which = 0, zq = blockIdx.z;
  while (which + 1 < L.np && zq >= L.p[which].batch) {
    zq -= L.p[which].batch;
    which++;
  }
  const GemmProb &P = L.p[which];
  const int m = P.m, n = P.n, k = P.k, lda = P.lda, ldb = P.ldb, ldc = P.ldc;
  const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
  if (i0 >= m || j0 >= n) return;  // (the grid is sized for the largest product of the launch)
  const double *A = P.A + (long)zq * P.sa, *B = P.B + (long)zq * P.sb;
  double *C = P.C + (long)zq * P.sc;
  const double alpha = P.alpha, beta = P.beta;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wi = 32 * (wave & 1), wj = 32 * (wave >> 1);
  const int li = lane & 15, lk = lane >> 4;
  const bool TA = P.ta != 0, TB = P.tb != 0;
  const bool AK = TA, BK = !TB;  // the panel is contiguous along k in global memory
  // this thread's four elements of each panel per step: (k inside the panel, row / column inside the tile)
  auto pos = [&](bool kmajor, int q, int &x, int &kk) {
    x = kmajor ? (tid >> 4) + 16 * q : (tid & 63);
    kk = kmajor ? (tid & 15) : (tid >> 6) + 4 * q;
  };
  auto fetch = [&](int k0, double (&ra)[4], double (&rb)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int ia, ka, jb, kb;
      pos(AK, q, ia, ka);
      pos(BK, q, jb, kb);
      const bool oa = i0 + ia < m && k0 + ka < k, ob = j0 + jb < n && k0 + kb < k;
      const long xa = !TA ? (long)(i0 + ia) + (long)(k0 + ka) * lda : (long)(k0 + ka) + (long)(i0 + ia) * lda;
      const long xb = !TB ? (long)(k0 + kb) + (long)(j0 + jb) * ldb : (long)(j0 + jb) + (long)(k0 + kb) * ldb;
      ra[q] = oa ? A[xa] : 0.0;
      rb[q] = ob ? B[xb] : 0.0;
    }
  };
  auto stage = [&](int buf, const double (&ra)[4], const double (&rb)[4]) {
    double *As = gemm_lds + (size_t)buf * 2 * kGemmPanel, *Bs = As + kGemmPanel;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int ia, ka, jb, kb;
      pos(AK, q, ia, ka);
      pos(BK, q, jb, kb);
      As[AK ? ia * kGemmLdK + ka : ka * kGemmLdM + ia] = ra[q];
      Bs[BK ? jb * kGemmLdK + kb : kb * kGemmLdM + jb] = rb[q];
    }
  };
  // The accumulators start from (beta / alpha) C -- alpha (op(A) op(B) + (beta / alpha) C) is the product asked for, exactly so for the
  // +-1 and 0 this path passes: the C tile of an updating product is requested before anything else and needs no registers of its own
  gemm_v4d acc[2][2];
  const double cscale = beta == 0.0 ? 0.0 : beta / alpha;
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = i0 + wi + 16 * a + 4 * r + lk, j = j0 + wj + 16 * b + li;
        acc[a][b][r] = (beta != 0.0 && i < m && j < n) ? cscale * C[(long)i + (long)j * ldc] : 0.0;
      }
  // operand addresses inside a panel: base + kk * step
  const int aBase = AK ? (wi + li) * kGemmLdK + lk : lk * kGemmLdM + wi + li, aTile = AK ? 16 * kGemmLdK : 16, aStep = AK ? 1 : kGemmLdM;
  const int bBase = BK ? (wj + li) * kGemmLdK + lk : lk * kGemmLdM + wj + li, bTile = BK ? 16 * kGemmLdK : 16, bStep = BK ? 1 : kGemmLdM;
  auto compute = [&](int buf) {
    const double *As = gemm_lds + (size_t)buf * 2 * kGemmPanel, *Bs = As + kGemmPanel;
#pragma unroll
    for (int kk = 0; kk < kGemmKs; kk += 4) {
      double va[2], vb[2];
#pragma unroll
      for (int a = 0; a < 2; a++) va[a] = As[aBase + a * aTile + kk * aStep];
#pragma unroll
      for (int b = 0; b < 2; b++) vb[b] = Bs[bBase + b * bTile + kk * bStep];
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[a], vb[b], acc[a][b], 0, 0, 0);
    }
  };
  const int nsteps = (k + kGemmKs - 1) / kGemmKs;
  double fa0[4], fb0[4], fa1[4], fb1[4];
  fetch(0, fa0, fb0);
  fetch(kGemmKs, fa1, fb1);  // step 1
  stage(0, fa0, fb0);
  fetch(2 * kGemmKs, fa0, fb0);  // step 2
  __syncthreads();
  for (int s = 0; s < nsteps; s += 2) {
    compute(0);  // step s
    if (s + 1 < nsteps) stage(1, fa1, fb1);
    fetch((s + 3) * kGemmKs, fa1, fb1);
    __syncthreads();
    if (s + 1 < nsteps) {
      compute(1);  // step s + 1
      if (s + 2 < nsteps) stage(0, fa0, fb0);
      fetch((s + 4) * kGemmKs, fa0, fb0);
      __syncthreads();
    }
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = i0 + wi + 16 * a + 4 * r + lk, j = j0 + wj + 16 * b + li;
        if (i < m && j < n && !(j >= P.skip_j0 && j < P.skip_j1)) C[(long)i + (long)j * ldc] = alpha * acc[a][b][r];
      }
}

#include "ba_generic.inc"

// the LM loop's scalars, written by the device straight into pinned host memory, a sequence number last: the host waits for the number
// instead of for a copy and a stream synchronisation
__global__ void __launch_bounds__(64) publish_kernel(const double *a, int na, double *ha, const int *b, int nb, int *hb, const double *c, int nc, double *hc, int *hseq,
                                                      int seq) {
  const int tid = threadIdx.x;
  for (int t = tid; t < na; t += 64) ha[t] = a[t];
  for (int t = tid; t < nb; t += 64) hb[t] = b[t];
  for (int t = tid; t < nc; t += 64) hc[t] = c[t];
  __threadfence_system();
  __builtin_amdgcn_wave_barrier();
  if (tid == 0) __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Device memory of one solve: slabs from the context's block cache, handed out by a bump pointer.  A solve makes ~100 arrays; as
// hipMalloc / hipFree pairs they were 3 ms of a local bundle adjustment's set-up and tear-down (5 ms of run) -- now the slabs of the last
// call of similar size are reused and nothing is freed.  512 bytes are left between arrays.  (The memory is NOT fresh: whatever the solver
// reads it has written in this call -- the host emulation poisons every slab it hands out, tests/native.)
struct Arena {
  std::vector<OsfmPoolBuf *> slabs;
  osfm_ctx *ctx = nullptr;  // set: slabs come from / go back to the context's cache (the caller holds the context lock)
  char *cur = nullptr;
  size_t left = 0, next_slab = (size_t)32 << 20;
  ~Arena() {
    for (OsfmPoolBuf *b : slabs) delete b;  // (OsfmPoolBuf returns the block to the cache, after a device sync when the call failed)
  }
  template <typename T>
  T *alloc(size_t n, hipError_t &e) {
    if (e != hipSuccess) return nullptr;
    const size_t bytes = (((n ? n : 1) * sizeof(T) + 255) / 256) * 256 + 512;
    if (bytes > left) {
      OsfmPoolBuf *b = new (std::nothrow) OsfmPoolBuf();
      if (!b) {
        e = hipErrorOutOfMemory;
        return nullptr;
      }
      const size_t want = std::max(bytes, next_slab);
      e = ctx ? b->alloc(ctx, want) : b->alloc(want);
      if (e != hipSuccess && want > bytes) {  // a slab of the growing size does not fit any more: exactly what is asked for
        (void)hipGetLastError();
        e = ctx ? b->alloc(ctx, bytes) : b->alloc(bytes);
      }
      if (e != hipSuccess) {
        b->p = nullptr;
        delete b;
        return nullptr;
      }
#ifdef OSFM_HIPEMU
      memset(b->p, 0xFF, b->bytes);  // a reused slab must not look initialised
#endif
      slabs.push_back(b);
      cur = (char *)b->p;
      left = b->bytes;
      next_slab = std::min(next_slab * 2, (size_t)2 << 30);
    }
    T *p = (T *)cur;
    cur += bytes;
    left -= bytes;
    return p;
  }
  // Small host arrays go through pinned staging memory of the context and an asynchronous copy on the solve's stream: a hipMemcpy from
  // pageable memory is a synchronous round trip of 15 - 30 us whatever the size, and a solve uploads twenty to forty-five of them
  // (0.3 ms of a local bundle adjustment's 0.5 ms set-up).  Large arrays (the observations) keep the direct copy.
  hipStream_t st = nullptr;
  size_t staged = 0;
  static constexpr size_t kStageBytes = (size_t)4 << 20, kStageMax = (size_t)256 << 10;
  template <typename T>
  T *upload(const T *h, size_t n, hipError_t &e) {
    T *p = alloc<T>(n, e);
    if (e != hipSuccess || n == 0) return p;
    const size_t bytes = n * sizeof(T), padded = (bytes + 63) / 64 * 64;
    if (ctx && st && bytes <= kStageMax && staged + padded <= kStageBytes) {
      if (!ctx->h_stage && hipHostMalloc(&ctx->h_stage, kStageBytes, hipHostMallocDefault) != hipSuccess) {
        ctx->h_stage = nullptr;
        (void)hipGetLastError();
      }
      if (ctx->h_stage) {
        char *dst = (char *)ctx->h_stage + staged;
        memcpy(dst, h, bytes);
        staged += padded;
        e = hipMemcpyAsync(p, dst, bytes, hipMemcpyHostToDevice, st);
        return p;
      }
    }
    e = hipMemcpy(p, h, bytes, hipMemcpyHostToDevice);
    return p;
  }
};

inline int nblk(long n, int t = TPB) { return (int)std::max<long>(1, (n + t - 1) / t); }  // (never an empty grid: the kernels test their index)

struct Solver {
  osfm_ctx *ctx;
  Dev d;
  hipStream_t st;
  int loss;
  double loss_a;
  // the LM loop's scalars come back through PINNED host memory (kept by the context): a D2H copy into pageable memory is staged and
  // synchronous in the runtime, ~25 us each; into pinned memory the copies of one round trip are queued back to back
  double *hscal = nullptr;  // 32 doubles
  int *hstat = nullptr;     // 4 ints
  double *hrr = nullptr;    // the blocks' shares of r.r (nbr doubles)
  int pinned(int nbr) {
    const size_t need = (size_t)(32 + 2 + nbr + 8) * sizeof(double);
    if (ctx->h_pinned_bytes < need) {
      if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
      ctx->h_pinned = nullptr;
      ctx->h_pinned_bytes = 0;
      OSFM_HIP(hipHostMalloc(&ctx->h_pinned, std::max<size_t>(need, 65536), hipHostMallocDefault));
      ctx->h_pinned_bytes = std::max<size_t>(need, 65536);
    }
    hscal = (double *)ctx->h_pinned;
    hstat = (int *)(hscal + 32);
    hrr = hscal + 34;
    hseq = (int *)(hscal + 34 + nbr);
    nbr_ = nbr;
    for (int i = 0; i < 34; i++) hscal[i] = 0.0;
    *hseq = 0;
    seq = 0;
    // (measured on the local problem: 4.89 -> 4.63 - 4.80 ms of run for ten iterations, 4.24 -> 4.20 ms per LM iteration at configs[4]; OSFM_BA_NO_SPIN
    //  keeps the copy + stream synchronisation)
    spin = getenv("OSFM_BA_NO_SPIN") == nullptr;
    if (const char *su = getenv("OSFM_BA_SPIN_US")) spin_us = atof(su);
    if (spin) OSFM_HIP(hipHostGetDevicePointer(&dev_pinned, ctx->h_pinned, 0));
    return OSFM_OK;
  }
  // one host round trip: a (na doubles) -> hscal + ha_off, b (nb ints) -> hstat + hb_off, c (nc doubles) -> hrr
  int *hseq = nullptr;
  int seq = 0, nbr_ = 0;
  bool spin = false;
  void *dev_pinned = nullptr;
  int fetch(const double *a, int na, int ha_off, const int *b = nullptr, int nb = 0, int hb_off = 0, const double *c = nullptr, int nc = 0) {
    if (!spin) {
      if (na) OSFM_HIP(hipMemcpyAsync(hscal + ha_off, a, (size_t)na * sizeof(double), hipMemcpyDeviceToHost, st));
      if (nb) OSFM_HIP(hipMemcpyAsync(hstat + hb_off, b, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, st));
      if (nc) OSFM_HIP(hipMemcpyAsync(hrr, c, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, st));
      OSFM_HIP(hipStreamSynchronize(st));
      return OSFM_OK;
    }
    double *dh = (double *)dev_pinned;
    const int want = ++seq;
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, st, a, na, dh + ha_off, b, nb, (int *)(dh + 32) + hb_off, c, nc, dh + 34, (int *)(dh + 34 + nbr_), want);
    OSFM_HIP(hipGetLastError());  // (of this launch or of any launch since the last round trip)
    // The host spins on the sequence number for a bounded time (spin_us, default 2 ms: a round trip of the LM loop is 20 us - 1 ms of queued
    // kernels), then hands the wait to the runtime (hipStreamSynchronize blocks without burning the core): long queues -- the first
    // evaluation of a cold context, a wide-band factorisation -- and OpenSfM's multi-process stages do not hold N cores at 100 %.
    const auto t0 = std::chrono::steady_clock::now();
    for (long it = 0; __atomic_load_n(hseq, __ATOMIC_ACQUIRE) != want; it++) {
      if ((it & 0x3FF) == 0x3FF && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 > spin_us) {
        OSFM_HIP(hipStreamSynchronize(st));  // a failed launch or a fault shows up here
        OSFM_REQUIRE(__atomic_load_n(hseq, __ATOMIC_ACQUIRE) == want, OSFM_E_HIP, "the device never published round trip %d", want);
      }
    }
    return OSFM_OK;
  }
  double spin_us = 2000.0;

  void rot(const double *poses) { hipLaunchKernelGGL(shot_rot_kernel, dim3(nblk(d.S, 64)), dim3(64), 0, st, d, poses); }

  // ---- generic mode (kernels: ba_generic.inc) ----
  int *g_cols = nullptr, *g_col_pos = nullptr;  // generic exact border: the border columns some view holds, and their positions (-1: none)
  int g_ncols = 0;
  double *g_wB = nullptr, *g_vpartB = nullptr;  // rows x ncols x NR, views x ncols x KW
  // five columns per launch of gen_border_shot_kernel for up to nine border slots (a Brown camera: two launches instead of three; measured 7.46
  // against 7.96 ms per LM iteration at configs[4] although the kernel then runs one wave per SIMD); OSFM_BA_BORDER_CH3 = three per launch
  bool gen_border_ch5 = getenv("OSFM_BA_BORDER_CH3") == nullptr;
  // round 6: all nine columns of a Brown camera in ONE launch (the rows are recomputed once, a record of wB is read once: 5.31 -> 5.16 ms per LM iteration
  // at configs[4]; OSFM_BA_BORDER_CH5 keeps the two launches of five and four columns)
  bool gen_border_ch9 = getenv("OSFM_BA_BORDER_CH5") == nullptr;
  int gen_uniform_model = -1;  // every camera has this projection type (the evaluation kernel is specialised for the common ones), -1: mixed
  bool gen_compact = false;    // ... and the rows keep (Xc, wt) instead of their border slots (gen_eval_kernel's COMPACT layout)
  bool have_bpri = false;  // a prior couples an instance with a free border block (position prior with a free bias, up vector / compass with a free rig camera)
#define OSFM_GEN_KW(NRV, MV, KERNEL, grid, block, stream, ...)                                                \
  do {                                                                                                        \
    if (d.g.KW <= 4) hipLaunchKernelGGL((KERNEL<NRV, 4, MV>), grid, block, 0, stream, __VA_ARGS__);            \
    else if (d.g.KW <= 9) hipLaunchKernelGGL((KERNEL<NRV, 9, MV>), grid, block, 0, stream, __VA_ARGS__);       \
    else if (d.g.KW <= 16) hipLaunchKernelGGL((KERNEL<NRV, 16, MV>), grid, block, 0, stream, __VA_ARGS__);     \
    else hipLaunchKernelGGL((KERNEL<NRV, kGenMaxKW, MV>), grid, block, 0, stream, __VA_ARGS__);                \
  } while (0)
  // the per-instance kernels recompute their rows (gen_sm_row): specialised, like the evaluation kernel, for the projection type every camera of the
  // problem has when that is one of the common three
#define OSFM_GEN_NR_KW(KERNEL, grid, block, stream, ...)                                                                                \
  do {                                                                                                                                  \
    if (d.g.NRr == 3) OSFM_GEN_KW(3, -1, KERNEL, grid, block, stream, __VA_ARGS__);                                                       \
    else if (gen_uniform_model == OSFM_CAMERA_BROWN) OSFM_GEN_KW(2, OSFM_CAMERA_BROWN, KERNEL, grid, block, stream, __VA_ARGS__);         \
    else if (gen_uniform_model == OSFM_CAMERA_FISHEYE_OPENCV) OSFM_GEN_KW(2, OSFM_CAMERA_FISHEYE_OPENCV, KERNEL, grid, block, stream, __VA_ARGS__); \
    else if (gen_uniform_model == OSFM_CAMERA_PERSPECTIVE) OSFM_GEN_KW(2, OSFM_CAMERA_PERSPECTIVE, KERNEL, grid, block, stream, __VA_ARGS__); \
    else OSFM_GEN_KW(2, -1, KERNEL, grid, block, stream, __VA_ARGS__);                                                                    \
  } while (0)
  int gen_nprior() const { return d.NC + d.g.NRC + d.S + 4 * d.g.NV; }
  // dynamic LDS of gen_prior_kernel: the workgroup's copy of Cpri and of the border gradient in mode 1, while the border is narrow enough for it
  // (beyond kGenPriorLdsMaxNB the kernel adds to the global arrays directly); modes 0 and 2 use none
  size_t gen_prior_lds(int mode) const {
    return (mode == 1 && d.g.NB <= kGenPriorLdsMaxNB) ? (size_t)((d.g.NB * d.g.NB + d.g.NB) * gen_prior_copies(d.g.NB) + 2) * sizeof(double) : 0;
  }
  // cost (with priors) at the given parameters into scal[8] (sum of squares of the reprojections into scal[9]); jac: the Jacobian rows
  // and the prior blocks as well
  void gen_eval_enqueue(const double *cam, const double *bias, const double *rcp, const double *poses, const double *pts, bool jac) {
    rot(poses);
    hipLaunchKernelGGL(gen_rc_rot_kernel, dim3(nblk(d.g.NRC, 64)), dim3(64), 0, st, d, rcp);
    const int nb = nblk(d.M);
    if (jac) {
      (void)hipMemsetAsync(d.g.PI, 0, (size_t)36 * d.S * sizeof(double), st);
      (void)hipMemsetAsync(d.g.Bpri, 0, (size_t)std::max(1, d.g.NB) * 6 * d.S * sizeof(double), st);
      (void)hipMemsetAsync(d.g.Cpri, 0, (size_t)std::max(1, d.g.NB * d.g.NB) * sizeof(double), st);
      (void)hipMemsetAsync(d.g.gpri, 0, (size_t)d.nred * sizeof(double), st);
    }
#define OSFM_GEN_EVAL(NRV, MODELV)                                                                                                           \
  do {                                                                                                                                       \
    if (jac && gen_compact && MODELV >= 0)                                                                                                   \
      hipLaunchKernelGGL((gen_eval_kernel<NRV, true, MODELV, (MODELV >= 0)>), dim3(nb), dim3(TPB), 0, st, d, cam, rcp, poses, pts, loss, loss_a); \
    else if (jac)                                                                                                                            \
      hipLaunchKernelGGL((gen_eval_kernel<NRV, true, MODELV>), dim3(nb), dim3(TPB), 0, st, d, cam, rcp, poses, pts, loss, loss_a);            \
    else                                                                                                                                     \
      hipLaunchKernelGGL((gen_eval_kernel<NRV, false, MODELV>), dim3(nb), dim3(TPB), 0, st, d, cam, rcp, poses, pts, loss, loss_a);           \
  } while (0)
    if (d.M > 0) {
      if (d.g.NRr == 3) OSFM_GEN_EVAL(3, -1);
      else if (gen_uniform_model == OSFM_CAMERA_BROWN) OSFM_GEN_EVAL(2, OSFM_CAMERA_BROWN);
      else if (gen_uniform_model == OSFM_CAMERA_FISHEYE_OPENCV) OSFM_GEN_EVAL(2, OSFM_CAMERA_FISHEYE_OPENCV);
      else if (gen_uniform_model == OSFM_CAMERA_PERSPECTIVE) OSFM_GEN_EVAL(2, OSFM_CAMERA_PERSPECTIVE);
      else OSFM_GEN_EVAL(2, -1);
    }
    hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(1024), 0, st, d.partial, (long)(d.M > 0 ? nb : 0), 2, d.scal + 8);
    hipLaunchKernelGGL(gen_prior_kernel, dim3(nblk(gen_nprior(), kGenPriorTPB)), dim3(kGenPriorTPB), gen_prior_lds(jac ? 1 : 0), st, d, cam, bias, rcp, poses, jac ? 1 : 0,
                       (const double *)nullptr, d.scal + 8);
    if (d.g.pt_prior_sigma && d.P > 0) hipLaunchKernelGGL(gen_point_prior_kernel, dim3(nblk(d.P)), dim3(TPB), 0, st, d, pts, 0, d.scal + 8);
  }
  void gen_gradients() {
    if (d.M > 0) {
      if (d.g.NRr == 3) hipLaunchKernelGGL(gen_point_grad_kernel<3>, dim3(d.nwg), dim3(kCoopObs), 0, st, d);
      else hipLaunchKernelGGL(gen_point_grad_kernel<2>, dim3(d.nwg), dim3(kCoopObs), 0, st, d);
    } else if (d.P > 0)
      hipLaunchKernelGGL(gen_point_grad_empty_kernel, dim3(nblk(d.P)), dim3(TPB), 0, st, d);
    OSFM_GEN_NR_KW(gen_shot_grad_kernel, dim3(d.S), dim3(64), st, d, loss, loss_a);
    if (d.g.NB > 0) {
      hipLaunchKernelGGL(gen_border_reduce_kernel, dim3(d.g.NB), dim3(256), 0, st, d, 2 * d.g.KW, 0, 1);
      hipLaunchKernelGGL(gen_border_reduce_kernel, dim3(d.g.NB), dim3(256), 0, st, d, 2 * d.g.KW, d.g.KW, 2);
    }
  }
  // pass A over two-row observations, mode 0 (mat-vec) / 1 (right-hand side) / 2 (back-substitution): the COMPACT rows' readers are specialised per projection type
  template <int MODE>
  void gen_schur_point2_mode(hipStream_t sq) {
    const dim3 grid(d.nwg), block(kCoopObs);
    if constexpr (MODE != 1) {  // (mode 1, the right-hand side, does not read the border slots)
      if (gen_compact && gen_uniform_model == OSFM_CAMERA_BROWN) {
        hipLaunchKernelGGL((gen_schur_point_kernel<2, MODE, OSFM_CAMERA_BROWN, true>), grid, block, 0, sq, d, (const double *)d.y);
        return;
      }
      if (gen_compact && gen_uniform_model == OSFM_CAMERA_FISHEYE_OPENCV) {
        hipLaunchKernelGGL((gen_schur_point_kernel<2, MODE, OSFM_CAMERA_FISHEYE_OPENCV, true>), grid, block, 0, sq, d, (const double *)d.y);
        return;
      }
      if (gen_compact && gen_uniform_model == OSFM_CAMERA_PERSPECTIVE) {
        hipLaunchKernelGGL((gen_schur_point_kernel<2, MODE, OSFM_CAMERA_PERSPECTIVE, true>), grid, block, 0, sq, d, (const double *)d.y);
        return;
      }
    }
    hipLaunchKernelGGL((gen_schur_point_kernel<2, MODE>), grid, block, 0, sq, d, (const double *)d.y);
  }
  void gen_schur_point2(int mode, hipStream_t sq) {
    if (mode == 0) gen_schur_point2_mode<0>(sq);
    else if (mode == 1) gen_schur_point2_mode<1>(sq);
    else gen_schur_point2_mode<2>(sq);
  }
  // the observation rows' share of J^T (I - Jp Hhat Jp^T) J y (mode 0, y = d.y) or of the right-hand side (mode 1) into zc, on stream sq
  void gen_rows_apply(int mode, hipStream_t sq) {
    if (d.M <= 0) return;
    if (mode == 0 && d.g.KW > 0) hipLaunchKernelGGL(gen_view_gather_kernel, dim3(nblk((long)d.g.NV * d.g.KW)), dim3(TPB), 0, sq, d, (const double *)d.y);
    if (d.g.NRr == 3) {
      if (mode == 0) hipLaunchKernelGGL((gen_schur_point_kernel<3, 0>), dim3(d.nwg), dim3(kCoopObs), 0, sq, d, (const double *)d.y);
      else hipLaunchKernelGGL((gen_schur_point_kernel<3, 1>), dim3(d.nwg), dim3(kCoopObs), 0, sq, d, (const double *)d.y);
    } else {
      gen_schur_point2(mode, sq);
    }
    OSFM_GEN_NR_KW(gen_schur_shot_kernel, dim3(d.S), dim3(64), sq, d);
    if (d.g.NB > 0) hipLaunchKernelGGL(gen_border_reduce_kernel, dim3(d.g.NB), dim3(256), 0, sq, d, 2 * d.g.KW, 0, 0);
  }
  // every column of the border in one pass over the observations (kernels at the end of ba_generic.inc) into Bc / dCm, on stream sq
  template <int NRV, int KWT, int CH, int MV>
  void gen_border_chunks(hipStream_t sq) {
    for (int c0 = 0; c0 < g_ncols; c0 += CH)
      hipLaunchKernelGGL((gen_border_shot_kernel<NRV, KWT, CH, MV>), dim3(d.S), dim3(64), 0, sq, d, (const int *)g_cols, g_ncols, c0, (const double *)g_wB, Bc, g_vpartB);
  }
  template <int NRV, int MV>
  void gen_border_columns_nr(hipStream_t sq) {
    if (d.M > 0 && g_ncols > 0) {
      if (gen_compact && MV >= 0)
        hipLaunchKernelGGL((gen_border_point_kernel<NRV, MV, (MV >= 0)>), dim3(d.nwg), dim3(kCoopObs), 0, sq, d, (const int *)g_cols, g_ncols, g_wB);
      else
        hipLaunchKernelGGL((gen_border_point_kernel<NRV>), dim3(d.nwg), dim3(kCoopObs), 0, sq, d, (const int *)g_cols, g_ncols, g_wB);
      if (d.g.KW <= 4) gen_border_chunks<NRV, 4, 4, MV>(sq);
      else if (d.g.KW <= 9 && g_ncols > 5 && gen_border_ch9) gen_border_chunks<NRV, 9, 9, MV>(sq);
      else if (d.g.KW <= 9 && gen_border_ch5) gen_border_chunks<NRV, 9, 5, MV>(sq);
      else if (d.g.KW <= 9) gen_border_chunks<NRV, 9, 3, MV>(sq);
      else if (d.g.KW <= 16) gen_border_chunks<NRV, 16, 2, MV>(sq);
      else gen_border_chunks<NRV, kGenMaxKW, 2, MV>(sq);
    }
  }
  void gen_border_columns(double radius, hipStream_t sq) {
    if (d.g.NRr == 3) gen_border_columns_nr<3, -1>(sq);
    else if (gen_uniform_model == OSFM_CAMERA_BROWN) gen_border_columns_nr<2, OSFM_CAMERA_BROWN>(sq);
    else if (gen_uniform_model == OSFM_CAMERA_FISHEYE_OPENCV) gen_border_columns_nr<2, OSFM_CAMERA_FISHEYE_OPENCV>(sq);
    else if (gen_uniform_model == OSFM_CAMERA_PERSPECTIVE) gen_border_columns_nr<2, OSFM_CAMERA_PERSPECTIVE>(sq);
    else gen_border_columns_nr<2, -1>(sq);
    const int NB = d.g.NB;
    hipLaunchKernelGGL(gen_border_finish_kernel, dim3(NB * NB + nblk((long)NB * 6 * d.S)), dim3(256), 0, sq, d, (const int *)g_col_pos, std::max(1, g_ncols),
                       (const double *)g_vpartB, Bc, dCm, radius, have_bpri ? 1 : 0);
  }
  // y_ready: d.y = sc x is there already (PCG's vector kernels leave it); dot_part: the shares of x . out as well (PCG's p . Ap)
  void gen_matvec(const double *x, double *out, double radius, hipStream_t sq, bool y_ready = false, double *dot_part = nullptr) {
    if (!y_ready) hipLaunchKernelGGL(scale_vec_kernel, dim3(nblk(d.nred)), dim3(TPB), 0, sq, d.sc_red, x, d.y, d.nred);
    gen_rows_apply(0, sq);
    if (have_bpri && d.g.NB > 0) hipLaunchKernelGGL(gen_bpri_dot_kernel, dim3(d.g.NB, kBpriSlices), dim3(256), 0, sq, d, (const double *)d.y);
    hipLaunchKernelGGL(gen_schur_finish_kernel, dim3(nblk(d.nred)), dim3(TPB), 0, sq, d, x, (const double *)d.y, out, radius, 0, d.M > 0 ? 1 : 0,
                       have_bpri ? 1 : 0, dot_part);
  }

  // cost (with priors) at (cams, poses, pts) into scal[8] (and the sum of squares into scal[9]); optionally builds the Jacobian
  // rot_done: shotR holds the rotation blocks of `poses` already (candidate_kernel writes them with the candidate)
  void eval_enqueue(const double *cams, const double *poses, const double *pts, bool jac, bool rot_done = false) {
    if (d.gen) {  // the candidate lives in the *_n arrays of every block
      const bool cand = poses == d.poses_n;
      return gen_eval_enqueue(cand ? d.g.cam_n : d.g.cam, cand ? d.g.bias_n : d.g.bias, cand ? d.g.rc_n : d.g.rc, poses, pts, jac);
    }
    if (!rot_done) rot(poses);
    if (jac) hipLaunchKernelGGL(eval_kernel<true>, dim3(d.nwg), dim3(kCoopObs), 0, st, d, cams, poses, pts, loss, loss_a);  // rows, point blocks, cost
    else hipLaunchKernelGGL(eval_kernel<false>, dim3(d.nwg), dim3(kCoopObs), 0, st, d, cams, poses, pts, loss, loss_a);
    hipLaunchKernelGGL(prior_cost_kernel, dim3(1), dim3(1024), 0, st, d, cams, poses, (const double *)d.partial, (long)d.nwg, d.scal + 8, d.scal + 10, jac ? 1 : 0);
  }
  int eval(const double *cams, const double *poses, const double *pts, bool jac, double *cost, double *sumsq) {
    eval_enqueue(cams, poses, pts, jac);
    OSFM_HIP(hipGetLastError());  // a launch the runtime refused (LDS, grid) must not come back as a cost of stale numbers
    {
      const int rcf = fetch(d.scal + 8, 2, 0);
      if (rcf != OSFM_OK) return rcf;
    }
    *cost = hscal[0];
    if (sumsq) *sumsq = hscal[1];
    return OSFM_OK;
  }
  void gradients() {
    if (d.gen) return gen_gradients();
    if (shot_waves() > 1) hipLaunchKernelGGL(shot_grad_kernel<kShotWavesSmall>, dim3(d.S), dim3(64 * kShotWavesSmall), 0, st, d, (const double *)d.poses, loss, loss_a);
    else hipLaunchKernelGGL(shot_grad_kernel<1>, dim3(d.S), dim3(64), 0, st, d, (const double *)d.poses, loss, loss_a);  // (the points' blocks: eval_kernel<true>)
    if (!cams_inert) hipLaunchKernelGGL(cam_reduce_kernel, dim3(d.NC), dim3(kCamRedT), 0, st, d, 9, (const double *)d.cams);
    else hipLaunchKernelGGL(cam_grad_kernel, dim3(nblk(d.NC, 64)), dim3(64), 0, st, d, d.cams);
  }
  bool use_band = false, use_ctri = false, use_bcr = false, use_border = false;
  // every camera constant (local / pose-only bundle adjustment): the camera rows have zero scale, zero gradient and zero right-hand side,
  // so the per-camera sums over the shots (cam_reduce_kernel, 4 launches of ~10 us per LM iteration of a 48-shot problem) are skipped and
  // camred stays at the zeros it is given at setup
  bool cams_inert = false;
  // second stream: the camera-border columns and the right-hand side only need the Jacobian and Hhat, so they run underneath the
  // cyclic-reduction factorisation (a latency chain of ~22 small launches that leaves most CUs idle)
  hipStream_t st2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  double *Bc = nullptr, *Wb = nullptr, *SigInv = nullptr, *dots = nullptr;  // border elimination (nb x 6S, nb x 6S, nb x nb, nb x nb)
  double *yb = nullptr;                                                     // ... r_c - B^T z_s (nb) between the two launches of the long-row path
  static constexpr int kBorderRhsSplit = 6 * 512;                           // rows of at least this many entries take the two-launch path
  double *wB = nullptr, *partB = nullptr, *dCm = nullptr;                     // its columns in one pass: w (2 nb per observation), camera partials, C
  // z_q = A^-1 r_q for nrhs right-hand sides (strides in doubles) in one walk of the levels
  bool use_sband = false;   // the band is factorised by one workgroup (sband_factor_kernel): few shots
  bool sband_solved = false;  // ... and this iteration's factorisation launch carried the solve of d.b and the start of PCG with it
  double *sbL = nullptr;    // ... its factor, in the band's layout
  void bcr_solve_set(const RhsSet &rs) {
    if (use_sband) {
      hipLaunchKernelGGL(sband_solve_kernel, dim3((rs.nrhs + kSbSolveWaves - 1) / kSbSolveWaves), dim3(64 * kSbSolveWaves), sband_solve_lds(d.S, d.bw), st, d,
                         (const double *)sbL, rs, SbFuse{0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});
      return;
    }
    const int N = d.ncl;
    const unsigned q = (unsigned)rs.nrhs;
    auto ne = [N](int s) { return (N + 2 * s - 1) / (2 * s); };
    int st0 = 2;  // from here on at most kMidWaves clusters are left on a level: one workgroup walks those levels down and up again
    while (ne(st0) > kMidWaves) st0 *= 2;
    hipLaunchKernelGGL(bcr_down_kernel, dim3(ne(1), q), dim3(64), 0, st, d, 1, rs, 1);
    for (int s = 2; s < st0; s *= 2) hipLaunchKernelGGL(bcr_down_kernel, dim3(ne(s), q), dim3(64), 0, st, d, s, rs, 0);
    hipLaunchKernelGGL(bcr_mid_kernel, dim3(1, q), dim3(64 * kMidWaves), 0, st, d, st0);
    for (int s = st0 / 2; s >= 2; s /= 2) hipLaunchKernelGGL(bcr_up_kernel, dim3(ne(s), q), dim3(64), 0, st, d, s, rs, 0);
    hipLaunchKernelGGL(bcr_up_kernel, dim3(std::max(ne(1), rs.cam_q >= 0 ? (d.NC + 63) / 64 : 0), q), dim3(64), 0, st, d, 1, rs, 1);
  }
  bool use_wide = false;
  // the same for the wide band: right-hand sides side by side (at most 4 per walk) through L y = b, z = D^-1 y, L^T x = z
  template <int NR>
  void wide_walk(const RhsSet &rs) {
    const int nrows = d.wNB * kWB;
    hipLaunchKernelGGL(wide_load_kernel<NR>, dim3(nblk(nrows)), dim3(TPB), 0, st, d, rs);
    for (int J = 0; J + 1 < d.wNB; J++)
      hipLaunchKernelGGL((wide_push_kernel<NR, false>), dim3(std::min(d.wWb, d.wNB - 1 - J)), dim3(kWB * kWKS), 0, st, d, J);
    hipLaunchKernelGGL(wide_diag_kernel<NR>, dim3(d.wNB), dim3(256), 0, st, d);
    for (int J = d.wNB - 1; J >= 1; J--) hipLaunchKernelGGL((wide_push_kernel<NR, true>), dim3(std::min(d.wWb, J)), dim3(kWB * kWKS), 0, st, d, J);
    hipLaunchKernelGGL(wide_store_kernel<NR>, dim3(nblk(std::max<long>(6L * d.S, d.NC))), dim3(TPB), 0, st, d, rs);
  }
  // ---- dense-cluster cyclic reduction of the wide band (kernels: dbcr_*) ----
  // C = alpha op(A) op(B) + beta C for `batch` column-major problems a stride apart (dgemm_mfma_kernel); gemm_launch sends up to three
  // independent products in one launch
  static GemmProb gemm_prob(bool ta, bool tb, int m, int n, int k, double alpha, const double *A, int lda, long sa, const double *B, int ldb, long sb, double beta,
                            double *C, int ldc, long sc, int batch, int skip_j0 = 0, int skip_j1 = 0) {
    GemmProb p;
    p.ta = ta; p.tb = tb; p.m = m; p.n = n; p.k = k; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.batch = batch; p.skip_j0 = skip_j0; p.skip_j1 = skip_j1;
    p.alpha = alpha; p.beta = beta; p.A = A; p.B = B; p.C = C; p.sa = sa; p.sb = sb; p.sc = sc;
    return p;
  }
  void gemm_launch(std::initializer_list<GemmProb> probs) {
    GemmList L;
    L.np = 0;
    int mx = 0, nx = 0, z = 0;
    for (const GemmProb &p : probs) {
      if (p.m <= 0 || p.n <= 0 || p.batch <= 0) continue;
      L.p[L.np++] = p;
      mx = std::max(mx, p.m);
      nx = std::max(nx, p.n);
      z += p.batch;
    }
    if (L.np == 0) return;
    hipLaunchKernelGGL(dgemm_mfma_kernel, dim3((unsigned)((mx + 63) / 64), (unsigned)((nx + 63) / 64), (unsigned)z), dim3(256), kGemmLds, st, L);
  }
  void dgemm_sb(bool ta, bool tb, int m, int n, int k, double alpha, const double *A, int lda, long sa, const double *B, int ldb, long sb, double beta, double *C,
                int ldc, long sc, int batch) {
    gemm_launch({gemm_prob(ta, tb, m, n, k, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, batch)});
  }
  // A_k <- A_k^-1 for `batch` SPD qm x qm blocks `strideA` apart: blocked Gauss-Jordan, panels of qT columns.  Per panel J:
  //   P = A_JJ^-1 (LDS), R = A_J,: and C = A_:,J copied (C's pivot rows zeroed);  Rn = P R;  A -= C Rn;  A_:,J = -C P;  A_J,: = Rn, A_JJ = P
  // look-ahead: stream st3 runs the pivot chain (dgj_pivot_ahead_kernel) beside the products of the panel before
  hipStream_t st3 = nullptr;
  hipEvent_t ev_rn[2] = {nullptr, nullptr}, ev_p[2] = {nullptr, nullptr};
  int dbcr_invert_batch_ahead(double *A, long strideA, int batch, int *d_status) {
    const int m = d.qm, T = d.qT;
    const double one = 1.0, neg = -1.0, zero = 0.0;
    const long sP = (long)T * T, sR = (long)T * m;
    const size_t lds = (size_t)2 * kWB * kWLd * sizeof(double);
    double *Pb[2] = {d.qP, d.qP2};
    hipLaunchKernelGGL(dgj_pivot_ahead_kernel, dim3(1, batch), dim3(256), lds, st, A, strideA, m, T, 1, 0, 0, std::min(T, m), (const double *)nullptr,
                       (const double *)nullptr, (const double *)nullptr, Pb[0], d_status);
    int J = 0;
    for (int j0 = 0; j0 < m; j0 += T, J++) {
      const int w = std::min(T, m - j0), jn = j0 + T, wn = jn < m ? std::min(T, m - jn) : 0;
      double *P = Pb[J & 1];
      const int ncopy = (int)std::min<long>(64, ((long)w * m + 255) / 256);
      hipLaunchKernelGGL(dgj_copy_kernel, dim3(ncopy, batch), dim3(256), 0, st, (const double *)A, strideA, m, T, j0, w, jn, wn, d.qR, d.qC, d.qBn);
      dgemm_sb(false, false, w, m, w, one, P, T, sP, d.qR, T, sR, zero, d.qRn, T, sR, batch);
      if (wn > 0) {
        OSFM_HIP(hipEventRecord(ev_rn[J & 1], st));
        OSFM_HIP(hipStreamWaitEvent(st3, ev_rn[J & 1], 0));
        hipLaunchKernelGGL(dgj_pivot_ahead_kernel, dim3(1, batch), dim3(256), lds, st3, (const double *)A, strideA, m, T, 0, w, jn, wn, (const double *)d.qC,
                           (const double *)d.qRn, (const double *)d.qBn, Pb[(J + 1) & 1], d_status);
        OSFM_HIP(hipEventRecord(ev_p[(J + 1) & 1], st3));
      }
      gemm_launch({gemm_prob(false, false, m, m, w, neg, d.qC, m, sR, d.qRn, T, sR, one, A, m, strideA, batch, j0, j0 + w),
                   gemm_prob(false, false, m, w, w, neg, d.qC, m, sR, P, T, sP, zero, A + (long)j0 * m, m, strideA, batch)});
      hipLaunchKernelGGL(dgj_scatter_kernel, dim3((unsigned)(((long)w * m + 255) / 256), batch), dim3(256), 0, st, A, strideA, m, T, j0, w, (const double *)P,
                         (const double *)d.qRn);
      if (wn > 0) OSFM_HIP(hipStreamWaitEvent(st, ev_p[(J + 1) & 1], 0));  // before the next panel's copies overwrite what the pivot kernel reads
    }
    return OSFM_OK;
  }
  int dbcr_invert_batch(double *A, long strideA, int batch, int *d_status) {
    if (st3) return dbcr_invert_batch_ahead(A, strideA, batch, d_status);
    const int m = d.qm, T = d.qT;
    const double one = 1.0, neg = -1.0, zero = 0.0;
    const long sP = (long)T * T, sR = (long)T * m;
    for (int j0 = 0; j0 < m; j0 += T) {
      const int w = std::min(T, m - j0);
      const int ncopy = (int)std::min<long>(64, ((long)w * m + 255) / 256);
      hipLaunchKernelGGL(dgj_pivot_kernel, dim3(1 + ncopy, batch), dim3(256), (size_t)(kWB * kWLd + kGj16Scratch) * sizeof(double), st, A, strideA, m, T, j0, w, d.qP, d.qR,
                         d.qC, d_status);
      dgemm_sb(false, false, w, m, w, one, d.qP, T, sP, d.qR, T, sR, zero, d.qRn, T, sR, batch);
      // the trailing update (which leaves the panel's columns alone) and the panel's own column block, -C P, in one launch
      gemm_launch({gemm_prob(false, false, m, m, w, neg, d.qC, m, sR, d.qRn, T, sR, one, A, m, strideA, batch, j0, j0 + w),
                   gemm_prob(false, false, m, w, w, neg, d.qC, m, sR, d.qP, T, sP, zero, A + (long)j0 * m, m, strideA, batch)});
      hipLaunchKernelGGL(dgj_scatter_kernel, dim3((unsigned)(((long)w * m + 255) / 256), batch), dim3(256), 0, st, A, strideA, m, T, j0, w, d.qP, d.qRn);
    }
    return OSFM_OK;
  }
  int dbcr_factor(int *d_status) {
    const int m = d.qm, N = d.qN;
    const long m2 = (long)m * m;
    const double neg = -1.0, one = 1.0, zero = 0.0;
    {
      static OsfmPerDeviceOnce once;
      const int rca = once.run(ctx->device, []() -> int {
        OSFM_HIP(hipFuncSetAttribute((const void *)dgj_pivot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        OSFM_HIP(hipFuncSetAttribute((const void *)dgj_pivot_ahead_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return OSFM_OK;
      });
      if (rca != OSFM_OK) return rca;
    }
    hipLaunchKernelGGL(dbcr_build_kernel, dim3((unsigned)((m2 + 255) / 256), (unsigned)N), dim3(256), 0, st, d, d_status);
    const unsigned tl = (unsigned)((m + 31) / 32);
    int cur = 0;
    for (int s = 1; s < N; s *= 2) {
      const int ne = (N - s - 1) / (2 * s) + 1;  // clusters (2k + 1) s < N
      const int nR = (N - 1) / (2 * s);          // ... that have a right neighbour (2k + 2) s < N
      const long sk = 2L * s * m2;               // from one cluster of the level to the next, in m x m blocks of one array
      double *Di = d.qD + (long)s * m2, *Ei = d.qE[cur] + (long)s * m2, *Er = d.qE[cur] + 2L * s * m2, *Xi = d.qX + (long)s * 2 * m2;
      const int rci = dbcr_invert_batch(Di, sk, ne, d_status);
      if (rci != OSFM_OK) return rci;
      // G_i = D_i^-1 E_i,  H_i = D_i^-1 E_r^T
      gemm_launch({gemm_prob(false, false, m, m, m, one, Di, m, sk, Ei, m, sk, zero, Xi, m, 2 * sk, ne),
                   gemm_prob(false, true, m, m, m, one, Di, m, sk, Er, m, sk, zero, Xi + m2, m, 2 * sk, nR)});
      hipLaunchKernelGGL(dbcr_transpose_kernel, dim3(tl, tl, 2 * ne), dim3(256), 0, st, Xi, 2 * sk, m2, d.qXt + (long)s * 2 * m2, 2 * sk, m2, m, 2);
      // D_{i+s} -= E_r H_i and E_{i+s} <- -E_r G_i (into the other buffer: E_r is an operand) together; D_{i-s} -= E_i^T G_i in a launch
      // of its own: a surviving cluster is the right neighbour of one eliminated cluster and the left neighbour of the next, so the two
      // updates of its D must not run side by side
      gemm_launch({gemm_prob(false, false, m, m, m, neg, Er, m, sk, Xi + m2, m, 2 * sk, one, d.qD + 2L * s * m2, m, sk, nR),
                   gemm_prob(false, false, m, m, m, neg, Er, m, sk, Xi, m, 2 * sk, zero, d.qE[1 - cur] + 2L * s * m2, m, sk, nR)});
      dgemm_sb(true, false, m, m, m, neg, Ei, m, sk, Xi, m, 2 * sk, one, d.qD, m, sk, ne);
      cur ^= 1;
    }
    const int rcr = dbcr_invert_batch(d.qD, m2, 1, d_status);  // the last cluster standing
    if (rcr != OSFM_OK) return rcr;
    OSFM_HIP(hipGetLastError());
    return OSFM_OK;
  }
  template <int NR>
  void dbcr_walk(const RhsSet &rs) {
    const int N = d.qN, m = d.qm;
    const unsigned gx = (unsigned)((m + 3) / 4);
    hipLaunchKernelGGL(dbcr_load_kernel<NR>, dim3(nblk((long)N * m)), dim3(TPB), 0, st, d, rs);
    int top = 1;
    for (int s = 1; s < N; s *= 2) {
      hipLaunchKernelGGL(dbcr_sweep_kernel<NR>, dim3(gx, (N + 2 * s - 1) / (2 * s)), dim3(256), 0, st, d, s, 0);
      top = s;
    }
    hipLaunchKernelGGL(dbcr_sweep_kernel<NR>, dim3(gx, 1), dim3(256), 0, st, d, 1, 2);
    if (N > 1)
      for (int s = top; s >= 1; s /= 2) hipLaunchKernelGGL(dbcr_sweep_kernel<NR>, dim3(gx, (N - s - 1) / (2 * s) + 1), dim3(256), 0, st, d, s, 1);
    hipLaunchKernelGGL(dbcr_store_kernel<NR>, dim3(nblk(std::max<long>(6L * d.S, d.NC))), dim3(TPB), 0, st, d, rs);
  }
  template <int NR>
  void wide_or_dense_walk(const RhsSet &rs) {
    if (d.qN > 0) dbcr_walk<NR>(rs);
    else wide_walk<NR>(rs);
  }
  void wide_solve_set(const RhsSet &rs) {
    if (rs.nrhs == 1) return wide_or_dense_walk<1>(rs);
    for (int q0 = 0; q0 < rs.nrhs; q0 += 4) {
      RhsSet c = rs;
      c.r = rs.r + q0 * rs.r_stride;
      c.z = rs.z + q0 * rs.z_stride;
      c.nrhs = std::min(4, rs.nrhs - q0);
      c.qx = (rs.qx >= q0 && rs.qx < q0 + c.nrhs) ? rs.qx - q0 : -1;
      c.cam_q = (rs.cam_q >= q0 && rs.cam_q < q0 + c.nrhs) ? rs.cam_q - q0 : -1;
      wide_or_dense_walk<4>(c);
    }
  }
  void exact_solve_set(const RhsSet &rs) {
    if (use_wide) wide_solve_set(rs);
    else bcr_solve_set(rs);
  }
  // z = M^-1 r; solved: the exact band solve of r is already in z (it went through the walk of the camera border's columns)
  double cur_radius = 0.0;  // generic mode: the border rows of the fallback preconditioners are the scaled diagonal at this radius
  void precond(const double *r, double *z, bool solved = false) {
    const RhsSet one{r, 0, z, 0, 1, nullptr, nullptr, -1, d.gen ? -1 : 0};
    if ((use_bcr || use_wide) && use_border) {
      const int nb = d.gen ? d.g.NB : 3 * d.NC, n = 6 * d.S;
      if (!solved) exact_solve_set(one);
      if (n >= kBorderRhsSplit) {  // long rows: a workgroup per row, then the nb x nb product (same bits as the single block)
        hipLaunchKernelGGL(border_rhs_dots_kernel, dim3(nb), dim3(1024), 0, st, Bc, r, (const double *)z, yb, n, d.cam0);
        hipLaunchKernelGGL(border_rhs_apply_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, SigInv, (const double *)yb, z, nb, d.cam0);
      } else if (d.gen) {
        hipLaunchKernelGGL(gen_border_rhs_kernel, dim3(1), dim3(1024), 0, st, Bc, SigInv, r, z, nb, n, d.cam0);
      } else {
        hipLaunchKernelGGL(border_rhs_kernel, dim3(1), dim3(1024), 0, st, Bc, SigInv, r, z, nb, n, d.cam0);
      }
      hipLaunchKernelGGL(border_update_kernel, dim3(nblk(n)), dim3(TPB), 0, st, Wb, z, nb, n, d.cam0);
      return;
    }
    if (use_bcr || use_wide)
      exact_solve_set(one);
    else if (use_ctri)
      hipLaunchKernelGGL(ctri_solve_kernel, dim3(1), dim3(256), 0, st, d, r, z);
    else if (use_band)
      hipLaunchKernelGGL(band_solve_for(d.bw), dim3(1), dim3(64), 0, st, d, r, z);
    else if (d.gen) {
      hipLaunchKernelGGL(gen_precond_apply_kernel, dim3(nblk(d.S + d.g.NB)), dim3(TPB), 0, st, d, r, z, cur_radius);
      return;
    } else
      hipLaunchKernelGGL(precond_apply_kernel, dim3(nblk(d.S + d.NC)), dim3(TPB), 0, st, d, r, z);
    if (d.gen && d.g.NB > 0) hipLaunchKernelGGL(gen_precond_border_kernel, dim3(nblk(d.g.NB, 64)), dim3(64), 0, st, d, r, z, cur_radius);
  }
  int shot_waves() const { return d.S <= kShotWavesBelow ? kShotWavesSmall : 1; }
  void schur_shot(hipStream_t sq) {
    if (shot_waves() > 1) hipLaunchKernelGGL(schur_shot_kernel<kShotWavesSmall>, dim3(d.S), dim3(64 * kShotWavesSmall), 0, sq, d);
    else hipLaunchKernelGGL(schur_shot_kernel<1>, dim3(d.S), dim3(64), 0, sq, d);
  }
  // workgroups of the finish kernel = entries of the p . Ap shares
  int matvec_parts() const { return d.gen ? nblk(d.nred) : nblk(d.cam0) + d.NC; }
  // y_ready: d.y = sc x is there already (PCG's vector kernels leave it); dot_part: the shares of x . out as well (PCG's p . Ap).  Three launches
  // (round 5: five -- the scaling, and the per-camera sums in front of the finish, were launches of their own)
  void matvec(const double *x, double *out, double radius, bool y_ready = false, double *dot_part = nullptr) {
    if (d.gen) return gen_matvec(x, out, radius, st, y_ready, dot_part);
    if (!y_ready) hipLaunchKernelGGL(scale_vec_kernel, dim3(nblk(d.nred)), dim3(TPB), 0, st, d.sc_red, x, d.y, d.nred);
    hipLaunchKernelGGL(schur_point_coop_kernel<0>, dim3(d.nwg), dim3(kCoopObs), 0, st, d, d.y);
    schur_shot(st);
    hipLaunchKernelGGL(schur_finish_kernel, dim3(matvec_parts()), dim3(TPB), 0, st, d, x, (const double *)d.y, out, radius, 0, cams_inert ? 1 : 0, dot_part);
  }
};

}  // namespace

extern "C" void osfm_ba_options_default(osfm_ba_options *o) {
  if (!o) return;
  o->loss = OSFM_LOSS_SOFTLONE;
  o->loss_threshold = 1.0;
  o->max_iterations = 100;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_radius = 1e4;
  o->verbose = 0;
  o->pcg_tolerance = 1e-10;
  // The FIRST iterate of a solve whose preconditioner is the reduced matrix itself (exact band + exact border / constant cameras) is accepted at
  // a relative residual of 1e-6 (rounds 1-5: it had to reach 1e-10 like any other).  That iterate is a direct solve by cyclic reduction -- as
  // Ceres' SPARSE_SCHUR is a direct solve by Cholesky, which nobody refines -- plus a line search; its residual lands between 1e-10 and 1e-7
  // (conditioning x the rounding of the explicit block inverses), so the 1e-10 stop bought a second mat-vec + walk in two LM iterations of
  // three for nothing the trajectory can see: cost histories with and without the rule agree to 4e-16 over 20 iterations at configs[4], both
  // are 1.3e-13 from the oracle's at configs[2] (profiles/r06_pcg_tolerance.json).  With an INEXACT preconditioner a 1e-6 residual is not a
  // direct solve's and does show (1e-6 in the cost on a 108-unknown border): those solves keep iterating to pcg_tolerance.
  o->pcg_direct_tolerance = 1e-6;
  o->pcg_max_iterations = 1000;
  o->preconditioner = 0;
}

// ---- shot ordering --------------------------------------------------------------------------------
// The banded preconditioner needs shots that share points to be close in index.  Sequences come that
// way; unordered collections do not.  When the caller's order has a co-visibility half-width above
// what the band can hold, the shots are renumbered by reverse Cuthill-McKee on the co-visibility
// graph (chain links inside every track: O(observations)); the solve runs in that numbering and the
// poses are written back in the caller's.  Pure relabelling: results are unchanged.
static int covis_half_bandwidth(const osfm_ba_problem *P, const std::vector<int> &new_of_old) {
  const int NP = P->n_points;
  std::vector<int> mn((size_t)NP, 1 << 30), mx((size_t)NP, -1);
  for (long o = 0; o < P->n_obs; o++) {
    const int s = new_of_old[(size_t)P->obs_shot[o]], p = P->obs_point[o];
    mn[(size_t)p] = std::min(mn[(size_t)p], s);
    mx[(size_t)p] = std::max(mx[(size_t)p], s);
  }
  int bw = 0;
  for (int p = 0; p < NP; p++)
    if (mx[(size_t)p] >= 0) bw = std::max(bw, mx[(size_t)p] - mn[(size_t)p]);
  return bw;
}

static void rcm_shot_order(const osfm_ba_problem *P, std::vector<int> &new_of_old) {
  const int S = P->n_shots, NP = P->n_points;
  const long M = P->n_obs;
  // chain links: consecutive observations of a point (in the caller's order) connect their shots
  std::vector<int> last((size_t)NP, -1);
  std::vector<long> deg((size_t)S + 1, 0);
  for (long o = 0; o < M; o++) {
    const int s = P->obs_shot[o], p = P->obs_point[o];
    if (last[(size_t)p] >= 0 && last[(size_t)p] != s) {
      deg[(size_t)s + 1]++;
      deg[(size_t)last[(size_t)p] + 1]++;
    }
    last[(size_t)p] = s;
  }
  for (int s = 0; s < S; s++) deg[(size_t)s + 1] += deg[(size_t)s];
  std::vector<int> adj((size_t)deg[(size_t)S]);
  {
    std::vector<long> fill(deg.begin(), deg.end() - 1);
    std::fill(last.begin(), last.end(), -1);
    for (long o = 0; o < M; o++) {
      const int s = P->obs_shot[o], p = P->obs_point[o];
      const int l = last[(size_t)p];
      if (l >= 0 && l != s) {
        adj[(size_t)fill[(size_t)s]++] = l;
        adj[(size_t)fill[(size_t)l]++] = s;
      }
      last[(size_t)p] = s;
    }
  }
  auto degree = [&](int v) { return deg[(size_t)v + 1] - deg[(size_t)v]; };
  std::vector<int> order;
  order.reserve((size_t)S);
  std::vector<char> seen((size_t)S, 0);
  std::vector<int> level((size_t)S, 0), nb;
  auto bfs = [&](int root, std::vector<int> &out, bool commit) -> int {  // returns the last vertex of the deepest level
    std::vector<char> vis = commit ? std::vector<char>() : std::vector<char>(seen);
    std::vector<char> &mark = commit ? seen : vis;
    const size_t start = out.size();
    out.push_back(root);
    mark[(size_t)root] = 1;
    for (size_t h = start; h < out.size(); h++) {
      const int v = out[h];
      nb.clear();
      for (long q = deg[(size_t)v]; q < deg[(size_t)v + 1]; q++) {
        const int u = adj[(size_t)q];
        if (!mark[(size_t)u]) {
          mark[(size_t)u] = 1;
          nb.push_back(u);
        }
      }
      std::sort(nb.begin(), nb.end(), [&](int a, int b) { return degree(a) != degree(b) ? degree(a) < degree(b) : a < b; });
      for (int u : nb) out.push_back(u);
    }
    const int far = out.back();
    if (!commit) out.resize(start);
    return far;
  };
  for (int s0 = 0; s0 < S; s0++) {
    if (seen[(size_t)s0]) continue;
    // pseudo-peripheral start: the far end of a BFS from the far end of a BFS
    int root = s0;
    for (int sweep = 0; sweep < 2; sweep++) root = bfs(root, order, false);
    bfs(root, order, true);
  }
  new_of_old.assign((size_t)S, 0);
  for (int i = 0; i < S; i++) new_of_old[(size_t)order[(size_t)(S - 1 - i)]] = i;  // reversed
}

// A sweep over the block: shots grouped into lines across the principal axis of their positions (a new line wherever the sorted first
// coordinate jumps by a quarter of the mean shot spacing, which follows from the extent of the two principal axes and the shot count), ordered
// along the second axis inside a line -- the column-major numbering of a survey flown in lines, whichever way it was flown.
static void sweep_shot_order(const osfm_ba_problem *P, std::vector<int> &new_of_old) {
  const int S = P->n_shots;
  double mean[3] = {0, 0, 0}, C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < S; s++)
    for (int i = 0; i < 3; i++) mean[i] += P->shot_pose[6 * (size_t)s + 3 + i] / S;
  for (int s = 0; s < S; s++)
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) C[3 * i + j] += (P->shot_pose[6 * (size_t)s + 3 + i] - mean[i]) * (P->shot_pose[6 * (size_t)s + 3 + j] - mean[j]);
  auto principal = [](const double *Cm, double *v) -> double {  // power iteration; returns the eigenvalue
    v[0] = 1.0; v[1] = 0.7; v[2] = 0.3;
    double lam = 0.0;
    for (int it = 0; it < 100; it++) {
      double w[3];
      for (int i = 0; i < 3; i++) w[i] = Cm[3 * i] * v[0] + Cm[3 * i + 1] * v[1] + Cm[3 * i + 2] * v[2];
      lam = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      if (!(lam > 0)) return 0.0;
      for (int i = 0; i < 3; i++) v[i] = w[i] / lam;
    }
    return lam;
  };
  double v1[3], v2[3];
  const double l1 = principal(C, v1);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C[3 * i + j] -= l1 * v1[i] * v1[j];
  principal(C, v2);
  std::vector<double> p1((size_t)S), p2((size_t)S);
  double lo1 = 1e300, hi1 = -1e300, lo2 = 1e300, hi2 = -1e300;
  for (int s = 0; s < S; s++) {
    const double *t = P->shot_pose + 6 * (size_t)s + 3;
    p1[(size_t)s] = t[0] * v1[0] + t[1] * v1[1] + t[2] * v1[2];
    p2[(size_t)s] = t[0] * v2[0] + t[1] * v2[1] + t[2] * v2[2];
    lo1 = std::min(lo1, p1[(size_t)s]); hi1 = std::max(hi1, p1[(size_t)s]);
    lo2 = std::min(lo2, p2[(size_t)s]); hi2 = std::max(hi2, p2[(size_t)s]);
  }
  const double h = std::sqrt((hi1 - lo1) * (hi2 - lo2) / S);
  // bins = runs of the sorted first coordinate without a gap of a quarter of the spacing (flight lines are such runs; where there are no gaps
  // the order is simply the sweep along the first axis with everything in one bin sorted by the second: the caller keeps the better)
  std::vector<int> by1((size_t)S);
  for (int s = 0; s < S; s++) by1[(size_t)s] = s;
  std::sort(by1.begin(), by1.end(), [&](int a, int b) { return p1[(size_t)a] != p1[(size_t)b] ? p1[(size_t)a] < p1[(size_t)b] : a < b; });
  std::vector<std::pair<std::pair<long, double>, int>> key((size_t)S);
  long bin = 0;
  for (int i = 0; i < S; i++) {
    const int s = by1[(size_t)i];
    if (i > 0 && p1[(size_t)s] - p1[(size_t)by1[(size_t)i - 1]] > 0.25 * h) bin++;
    key[(size_t)s] = {{bin, p2[(size_t)s]}, s};
  }
  if (bin == 0)  // no lines: plain sweep
    for (int s = 0; s < S; s++) key[(size_t)s] = {{0L, p1[(size_t)s]}, s};
  std::sort(key.begin(), key.end());
  new_of_old.assign((size_t)S, 0);
  for (int i = 0; i < S; i++) new_of_old[(size_t)key[(size_t)i].second] = i;
}

// the narrower of the two renumberings; returns its co-visibility half-width
static int best_shot_order(const osfm_ba_problem *P, std::vector<int> &new_of_old) {
  rcm_shot_order(P, new_of_old);
  int bw = covis_half_bandwidth(P, new_of_old);
  if (P->shot_pose) {  // breadth-first levels from a corner are diagonals of a block survey; a sweep along its long side is as narrow as
    std::vector<int> sweep;  // the block is wide
    sweep_shot_order(P, sweep);
    const int bw2 = covis_half_bandwidth(P, sweep);
    if (bw2 < bw) {
      new_of_old.swap(sweep);
      bw = bw2;
    }
  }
  return bw;
}

// The generic mode's inputs beyond what an osfm_ba_problem carries (host arrays, instances already in the order the band is built in;
// ba_generic_host.inc builds it from an osfm_bundle_problem).  With it the osfm_ba_problem means: n_shots = rig instances, shot_pose =
// their poses, shot_fixed, obs_shot = the INSTANCE of a row, obs_xy / obs_sigma of a row (a depth-prior row: x = the depth, sigma its
// sd), points / point_fixed; its cam_params / cam_prior / cam_sigma / shot_camera / gps / up fields are unused.
struct GenInput {
  int NV, NRC;                                 // views (the caller's shots), rig cameras
  const int *view_inst, *view_rc, *view_cam;   // NV; the views of an instance consecutive, instances ascending
  const int *obs_view;                         // per row
  const unsigned char *obs_kind;               // per row or null
  const int *cam_model;                        // NC
  double *cam;                                 // NC x 16, in / out
  const double *cam_prior, *cam_sigma;         // NC x 16
  const unsigned char *cam_fixed;              // NC
  double *bias;                                // NC x 7, in / out
  const unsigned char *bias_fixed;             // NC
  double *rc_pose;                             // NRC x 6, in / out
  const double *rc_prior, *rc_sigma;           // NRC x 6 or null
  const unsigned char *rc_fixed;               // NRC
  const double *gps, *gps_sigma;               // instances x 3 or null
  const int *inst_bias_cam;                    // instances or null
  const double *up, *up_sigma;                 // NV x 3, NV or null
  const double *pan, *pan_sigma, *tilt, *tilt_sigma, *roll, *roll_sigma;  // NV each or null
  const double *pt_prior, *pt_prior_sigma;     // points x 3 or null
  const unsigned char *pt_prior_alt;           // points or null (= all 1)
  double *reproj3;                             // rows x 3 or null: out, residuals with sigma 1 in the caller's row order
  long rows0;                                  // reprojection rows (the depth-prior rows do not count in the RMSE)
};
static int ba_solve_impl(osfm_ctx *ctx, osfm_ba_problem *P, const osfm_ba_options *O, osfm_ba_report *Rp, const GenInput *G = nullptr);

// Host-only helper (no GPU needed): the renumbering osfm_ba_solve would apply.  new_of_old[n_shots] receives the
// reverse Cuthill-McKee order; returns the co-visibility half-width in the caller's order / after renumbering.
extern "C" int osfm_ba_shot_order(const osfm_ba_problem *P, int32_t *new_of_old, int32_t *half_width_before, int32_t *half_width_after) {
  OSFM_REQUIRE(P && new_of_old && P->n_shots > 0 && P->n_points > 0 && P->n_obs > 0 && P->obs_shot && P->obs_point, OSFM_E_INVALID,
               "osfm_ba_shot_order: bad argument");
  for (long o = 0; o < P->n_obs; o++)
    OSFM_REQUIRE(P->obs_shot[o] >= 0 && P->obs_shot[o] < P->n_shots && P->obs_point[o] >= 0 && P->obs_point[o] < P->n_points,
                 OSFM_E_INVALID, "observation %ld references shot %d / point %d", o, P->obs_shot[o], P->obs_point[o]);
  std::vector<int> ident((size_t)P->n_shots), order;
  for (int s = 0; s < P->n_shots; s++) ident[(size_t)s] = s;
  const int bw_after = best_shot_order(P, order);
  if (half_width_before) *half_width_before = covis_half_bandwidth(P, ident);
  if (half_width_after) *half_width_after = bw_after;
  for (int s = 0; s < P->n_shots; s++) new_of_old[s] = order[(size_t)s];
  return OSFM_OK;
}

extern "C" int osfm_ba_solve(osfm_ctx *ctx, osfm_ba_problem *P, const osfm_ba_options *O, osfm_ba_report *Rp) {
  OSFM_REQUIRE(ctx && P && O && Rp, OSFM_E_INVALID, "osfm_ba_solve: null argument");
  if (!(P->n_shots > 2 && P->n_points > 0 && P->n_obs > 0 && P->obs_shot && P->obs_point && P->shot_pose && P->shot_camera) ||
      O->preconditioner == 1)
    return ba_solve_impl(ctx, P, O, Rp);
  const int S = P->n_shots;
  for (long o = 0; o < P->n_obs; o++)
    if (P->obs_shot[o] < 0 || P->obs_shot[o] >= S || P->obs_point[o] < 0 || P->obs_point[o] >= P->n_points)
      return ba_solve_impl(ctx, P, O, Rp);  // reports the invalid index
  std::vector<int> ident((size_t)S), new_of_old;
  for (int s = 0; s < S; s++) ident[(size_t)s] = s;
  const int bw0 = covis_half_bandwidth(P, ident);
  if (bw0 <= 10) return ba_solve_impl(ctx, P, O, Rp);  // already a narrow band: the exact cyclic-reduction path
  const int bw1 = best_shot_order(P, new_of_old);
  if (bw1 >= bw0) return ba_solve_impl(ctx, P, O, Rp);
  // relabelled copy of every per-shot input
  std::vector<int> old_of_new((size_t)S);
  for (int s = 0; s < S; s++) old_of_new[(size_t)new_of_old[(size_t)s]] = s;
  std::vector<double> pose((size_t)6 * S), gps, gps_sg, up, up_sg;
  std::vector<int32_t> cam((size_t)S), oshot((size_t)P->n_obs);
  std::vector<uint8_t> fixed;
  for (int i = 0; i < S; i++) {
    const int q = old_of_new[(size_t)i];
    for (int k = 0; k < 6; k++) pose[(size_t)6 * i + k] = P->shot_pose[6 * (size_t)q + k];
    cam[(size_t)i] = P->shot_camera[q];
  }
  auto permute = [&](const double *src, int width, std::vector<double> &dst) {
    dst.resize((size_t)width * S);
    for (int i = 0; i < S; i++)
      for (int k = 0; k < width; k++) dst[(size_t)width * i + k] = src[(size_t)width * old_of_new[(size_t)i] + k];
  };
  osfm_ba_problem Q = *P;
  Q.shot_pose = pose.data();
  Q.shot_camera = cam.data();
  if (P->shot_fixed) {
    fixed.resize((size_t)S);
    for (int i = 0; i < S; i++) fixed[(size_t)i] = P->shot_fixed[old_of_new[(size_t)i]];
    Q.shot_fixed = fixed.data();
  }
  if (P->shot_gps && P->shot_gps_sigma) {
    permute(P->shot_gps, 3, gps);
    permute(P->shot_gps_sigma, 1, gps_sg);
    Q.shot_gps = gps.data();
    Q.shot_gps_sigma = gps_sg.data();
  }
  if (P->shot_up && P->shot_up_sigma) {
    permute(P->shot_up, 3, up);
    permute(P->shot_up_sigma, 1, up_sg);
    Q.shot_up = up.data();
    Q.shot_up_sigma = up_sg.data();
  }
  for (long o = 0; o < P->n_obs; o++) oshot[(size_t)o] = new_of_old[(size_t)P->obs_shot[o]];
  Q.obs_shot = oshot.data();
  const int rc = ba_solve_impl(ctx, &Q, O, Rp);
  for (int i = 0; i < S; i++)
    for (int k = 0; k < 6; k++) P->shot_pose[6 * (size_t)old_of_new[(size_t)i] + k] = pose[(size_t)6 * i + k];
  Rp->shots_reordered = 1;
  Rp->shot_bandwidth_input = bw0;
  return rc;
}

static int ba_solve_impl(osfm_ctx *ctx, osfm_ba_problem *P, const osfm_ba_options *O, osfm_ba_report *Rp, const GenInput *G) {
  OSFM_REQUIRE(ctx && P && O && Rp, OSFM_E_INVALID, "osfm_ba_solve: null argument");
  const bool gen = G != nullptr;
  if (gen) {
    OSFM_REQUIRE(P->n_cameras > 0 && P->n_shots > 0 && P->n_points >= 0 && P->n_obs >= 0, OSFM_E_INVALID, "empty bundle problem");
    OSFM_REQUIRE(P->shot_pose && (P->n_points == 0 || P->points) && (P->n_obs == 0 || (P->obs_shot && P->obs_point && P->obs_xy && P->obs_sigma && G->obs_view)),
                 OSFM_E_INVALID, "osfm_bundle_solve: a required array is null");
  } else {
    OSFM_REQUIRE(P->n_cameras > 0 && P->n_shots > 0 && P->n_points > 0 && P->n_obs > 0, OSFM_E_INVALID, "empty BA problem");
    OSFM_REQUIRE(P->cam_params && P->cam_prior && P->cam_sigma && P->cam_fixed && P->shot_pose && P->shot_camera && P->points &&
                     P->obs_shot && P->obs_point && P->obs_xy && P->obs_sigma,
                 OSFM_E_INVALID, "osfm_ba_solve: a required array is null");
  }
  OSFM_REQUIRE(O->loss >= 0 && O->loss <= 3, OSFM_E_INVALID, "unknown loss %d (bundle_adjuster.cc:427 throws)", O->loss);
  const auto t_start = std::chrono::steady_clock::now();
  memset(Rp, 0, sizeof(*Rp));
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  const int S = P->n_shots, NP = P->n_points, NC = P->n_cameras;
  const long M = P->n_obs;
  const long G_rows0 = gen ? G->rows0 : M;
  for (long o = 0; o < M; o++) {
    OSFM_REQUIRE(P->obs_shot[o] >= 0 && P->obs_shot[o] < S && P->obs_point[o] >= 0 && P->obs_point[o] < NP, OSFM_E_INVALID,
                 "observation %ld references shot %d / point %d", o, P->obs_shot[o], P->obs_point[o]);
    OSFM_REQUIRE(P->obs_sigma[o] > 0, OSFM_E_INVALID, "observation %ld has std_deviation <= 0", o);
  }
  if (P->cam_model && !gen)
    for (int c = 0; c < NC; c++)
    {
      OSFM_REQUIRE(P->cam_model[c] >= OSFM_CAMERA_PERSPECTIVE && P->cam_model[c] <= OSFM_CAMERA_SIMPLE_RADIAL, OSFM_E_UNSUPPORTED,
                   "camera %d: projection type %d is not on the GPU path (SPHERICAL has a 3-D residual)", c, P->cam_model[c]);
      OSFM_REQUIRE(P->cam_model[c] < 2 || (P->cam_fixed[c] && P->cam_ext), OSFM_E_UNSUPPORTED,
                   "camera %d: projection type %d is supported as a CONSTANT camera only (cam_fixed = 1 and cam_ext given); its "
                   "intrinsics are not optimised on the GPU path", c, P->cam_model[c]);
    }
  for (int s = 0; s < S && !gen; s++)
    OSFM_REQUIRE(P->shot_camera[s] >= 0 && P->shot_camera[s] < NC, OSFM_E_INVALID, "shot %d references camera %d", s, P->shot_camera[s]);

  OSFM_REQUIRE(M < (1L << 31), OSFM_E_UNSUPPORTED, "more than 2^31 observations");

  // ---- device image ----
  Arena A;
  A.ctx = ctx;
  hipError_t e = hipSuccess;
  Solver sv;
  sv.ctx = ctx;
  sv.st = ctx->stream;
  A.st = sv.st;
  struct SideStream {
    Solver &sv;
    ~SideStream() {
      for (int q = 0; q < 2; q++) {
        if (sv.ev_rn[q]) (void)hipEventDestroy(sv.ev_rn[q]);
        if (sv.ev_p[q]) (void)hipEventDestroy(sv.ev_p[q]);
      }
      if (sv.st3) (void)hipStreamDestroy(sv.st3);
    }
  } side_stream{sv};
  if (getenv("OSFM_BA_ONE_STREAM") == nullptr) {  // measurement knob: everything on one stream
    // the side stream and its two events live in the context (creating and destroying a stream per solve is a millisecond of a local
    // bundle adjustment's call); (a low-priority side stream was measured: no difference)
    // (round 6, measured and dropped: the side stream confined to 192 / 128 / 64 CUs by hipExtStreamCreateWithCUMask so that the cyclic
    //  reduction's 117 KB workgroups find free LDS elsewhere -- 2.93 - 2.98 ms per LM iteration at configs[4] for every mask, as without one:
    //  profiles/r06_ba_variants4_side_cu_mask.json)
    if (!ctx->stream_b) OSFM_HIP(hipStreamCreateWithFlags(&ctx->stream_b, hipStreamNonBlocking));
    for (int q = 0; q < 2; q++)
      if (!ctx->ev_side[q]) OSFM_HIP(hipEventCreateWithFlags(&ctx->ev_side[q], hipEventDisableTiming));
    sv.st2 = ctx->stream_b;
    sv.ev_fork = ctx->ev_side[0];
    sv.ev_join = ctx->ev_side[1];
  }
  sv.loss = O->loss;
  sv.loss_a = O->loss_threshold;
  Dev &d = sv.d;
  memset(&d, 0, sizeof(d));
  d.S = S; d.P = NP; d.NC = NC; d.M = M;
  d.cam0 = 6 * S;
  d.gen = gen ? 1 : 0;
  GenDev &g = d.g;
  std::vector<int> inst_view0;
  std::vector<unsigned char> gen_col_slot_host;
  if (gen) {  // border columns: free rig cameras, free cameras, free biases; per view the slots its rows touch
    const int NV = G->NV, NRC = G->NRC;
    g.NV = NV;
    g.NRC = NRC;
    std::vector<int> cam_col((size_t)NC, -1), rc_col((size_t)NRC, -1), bias_col((size_t)NC, -1);
    std::vector<unsigned char> rc_useful((size_t)NRC, 1);
    int nb = 0;
    bool spherical = false;
    for (int q = 0; q < NRC; q++) {
      const bool fixed = G->rc_fixed[q] != 0;
      bool zero = true;
      for (int k = 0; k < 6; k++) zero = zero && G->rc_pose[6 * q + k] == 0.0;
      rc_useful[(size_t)q] = !(fixed && zero);  // IsRigCameraUseful, bundle_adjuster.cc:17-20
      if (!fixed) {
        rc_col[(size_t)q] = nb;
        nb += 6;
      }
    }
    for (int c = 0; c < NC; c++) {
      const int nk = model_num_params(G->cam_model[c]);
      spherical = spherical || G->cam_model[c] == OSFM_CAMERA_SPHERICAL;
      if (!G->cam_fixed[c] && nk > 0) {
        cam_col[(size_t)c] = nb;
        nb += nk;
      }
    }
    for (int c = 0; c < NC; c++)
      if (G->bias_fixed && !G->bias_fixed[c]) {
        bias_col[(size_t)c] = nb;
        nb += 7;
      }
    g.NB = nb;
    g.NRr = spherical ? 3 : 2;
    sv.gen_uniform_model = G->cam_model[0];
    for (int c = 1; c < NC; c++)
      if (G->cam_model[c] != G->cam_model[0]) sv.gen_uniform_model = -1;
    if (getenv("OSFM_BA_GENERIC_EVAL") != nullptr) sv.gen_uniform_model = -1;  // (self-check knob: the unspecialised evaluation kernel)
    {  // the COMPACT rows: one 2-D projection type, reprojection rows only, no free rig camera (its six columns are border slots too); OSFM_BA_GEN_FULL_ROWS keeps the slots
      bool rc_free = false, depth_rows = false;
      for (int q = 0; q < NRC; q++) rc_free = rc_free || (rc_col[(size_t)q] >= 0 && rc_useful[(size_t)q]);
      if (G->obs_kind)
        for (long o = 0; o < M && !depth_rows; o++) depth_rows = G->obs_kind[o] != 0;
      const int um = sv.gen_uniform_model;
      sv.gen_compact = !spherical && !rc_free && !depth_rows && getenv("OSFM_BA_GEN_FULL_ROWS") == nullptr &&
                       (um == OSFM_CAMERA_BROWN || um == OSFM_CAMERA_FISHEYE_OPENCV || um == OSFM_CAMERA_PERSPECTIVE);
    }
    int KW = 0;
    auto view_slots = [&](int v, int *cols) {  // returns the number of slots; cols[i] = border column of slot i
      const int c = G->view_cam[v], q = G->view_rc[v];
      int n = 0;
      if (cam_col[(size_t)c] >= 0)
        for (int k = 0; k < model_num_params(G->cam_model[c]); k++) cols[n++] = cam_col[(size_t)c] + k;
      if (rc_col[(size_t)q] >= 0 && rc_useful[(size_t)q])
        for (int k = 0; k < 6; k++) cols[n++] = rc_col[(size_t)q] + k;
      return n;
    };
    int tmp[kGenMaxKW];
    for (int v = 0; v < NV; v++) KW = std::max(KW, view_slots(v, tmp));
    g.KW = KW;
    g.oJp = g.NRr;
    g.oJc = 4 * g.NRr;
    g.oJb = 10 * g.NRr;
    g.ncomp = sv.gen_compact ? g.NRr * 10 + 4 : g.NRr * (10 + KW);  // (COMPACT: Xc | wt in place of the 2 KW border slots)
    std::vector<int> view_col((size_t)std::max(1, NV * KW), -1);
    std::vector<unsigned char> col_slot((size_t)std::max(1, nb) * NV, 255);
    for (int v = 0; v < NV; v++) {
      const int n = view_slots(v, tmp);
      for (int i = 0; i < n; i++) {
        view_col[(size_t)v * KW + i] = tmp[i];
        col_slot[(size_t)tmp[i] * NV + v] = (unsigned char)i;
      }
    }
    inst_view0.assign((size_t)S + 1, 0);
    for (int v = 0; v < NV; v++) {
      OSFM_REQUIRE(G->view_inst[v] >= 0 && G->view_inst[v] < S && (v == 0 || G->view_inst[v] >= G->view_inst[v - 1]), OSFM_E_INVALID,
                   "generic bundle: the views are not grouped by instance");
      inst_view0[(size_t)G->view_inst[v] + 1]++;
    }
    for (int i = 0; i < S; i++) inst_view0[(size_t)i + 1] += inst_view0[(size_t)i];
    g.view_inst = A.upload(G->view_inst, (size_t)NV, e);
    g.view_rc = A.upload(G->view_rc, (size_t)NV, e);
    g.view_cam = A.upload(G->view_cam, (size_t)NV, e);
    g.view_col = A.upload(view_col.data(), view_col.size(), e);
    g.col_slot = A.upload(col_slot.data(), col_slot.size(), e);
    gen_col_slot_host = col_slot;
    g.inst_view0 = A.upload(inst_view0.data(), inst_view0.size(), e);
    g.cam_col = A.upload(cam_col.data(), (size_t)NC, e);
    g.rc_col = A.upload(rc_col.data(), (size_t)NRC, e);
    g.bias_col = A.upload(bias_col.data(), (size_t)NC, e);
    g.rc_useful = A.upload(rc_useful.data(), (size_t)NRC, e);
    g.cam = A.upload(G->cam, (size_t)16 * NC, e);
    g.cam_n = A.alloc<double>((size_t)16 * NC, e);
    g.cam_prior = A.upload(G->cam_prior, (size_t)16 * NC, e);
    g.cam_sigma = A.upload(G->cam_sigma, (size_t)16 * NC, e);
    g.bias = A.upload(G->bias, (size_t)7 * NC, e);
    g.bias_n = A.alloc<double>((size_t)7 * NC, e);
    g.rc = A.upload(G->rc_pose, (size_t)6 * NRC, e);
    g.rc_n = A.alloc<double>((size_t)6 * NRC, e);
    g.rcR = A.alloc<double>((size_t)36 * NRC, e);
    if (G->rc_prior && G->rc_sigma) {
      g.rc_prior = A.upload(G->rc_prior, (size_t)6 * NRC, e);
      g.rc_sigma = A.upload(G->rc_sigma, (size_t)6 * NRC, e);
    }
    if (G->gps && G->gps_sigma && G->inst_bias_cam) {
      g.gps = A.upload(G->gps, (size_t)3 * S, e);
      g.gps_sigma = A.upload(G->gps_sigma, (size_t)3 * S, e);
      g.inst_bias_cam = A.upload(G->inst_bias_cam, (size_t)S, e);
      for (int i = 0; i < S; i++)
        if (G->gps_sigma[3 * i] > 0 && bias_col[(size_t)G->inst_bias_cam[i]] >= 0) sv.have_bpri = true;
    }
    auto per_view = [&](const double *val, const double *sd, int width, const double *&dv, const double *&ds) {
      if (!val || !sd) return;
      dv = A.upload(val, (size_t)width * NV, e);
      ds = A.upload(sd, (size_t)NV, e);
      for (int v = 0; v < NV; v++)
        if (sd[v] > 0 && rc_col[(size_t)G->view_rc[v]] >= 0) sv.have_bpri = true;
    };
    if (G->up && G->up_sigma)
      for (int v = 0; v < NV; v++) {
        const double *u = G->up + 3 * (size_t)v;
        OSFM_REQUIRE(!(G->up_sigma[v] > 0) || std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) >= 1e-10, OSFM_E_INVALID,
                     "UpVectorError: acceleration vector has near-zero magnitude");
      }
    per_view(G->up, G->up_sigma, 3, g.up, g.up_sigma);
    per_view(G->pan, G->pan_sigma, 1, g.pan, g.pan_sigma);
    per_view(G->tilt, G->tilt_sigma, 1, g.tilt, g.tilt_sigma);
    per_view(G->roll, G->roll_sigma, 1, g.roll, g.roll_sigma);
    if (G->pt_prior && G->pt_prior_sigma && NP > 0) {
      g.pt_prior = A.upload(G->pt_prior, (size_t)3 * NP, e);
      g.pt_prior_sigma = A.upload(G->pt_prior_sigma, (size_t)3 * NP, e);
      std::vector<unsigned char> alt((size_t)NP, 1);
      if (G->pt_prior_alt) alt.assign(G->pt_prior_alt, G->pt_prior_alt + NP);
      g.pt_prior_alt = A.upload(alt.data(), (size_t)NP, e);
    }
    g.PI = A.alloc<double>((size_t)36 * S, e);
    g.Bpri = A.alloc<double>((size_t)std::max(1, nb) * 6 * S, e);
    g.Cpri = A.alloc<double>((size_t)std::max(1, nb * nb), e);
    g.gpri = A.alloc<double>((size_t)6 * S + nb, e);
    g.bdot = A.alloc<double>((size_t)std::max(1, nb) * kBpriSlices, e);
    g.vpart = A.alloc<double>((size_t)std::max(1, NV * 2 * KW), e);
    g.yv = A.alloc<double>((size_t)std::max(1, NV * KW), e);
  }
  const int nbord = gen ? g.NB : 3 * NC;  // border unknowns behind the 6 S of the band
  d.nred = 6 * S + nbord;
  d.cams = gen ? nullptr : A.upload(P->cam_params, (size_t)3 * NC, e);
  d.poses = A.upload(P->shot_pose, (size_t)6 * S, e);
  d.pts = A.upload(P->points, (size_t)3 * NP, e);
  d.cams_n = gen ? nullptr : A.alloc<double>((size_t)3 * NC, e);
  d.poses_n = A.alloc<double>((size_t)6 * S, e);
  d.pts_n = A.alloc<double>((size_t)3 * NP, e);
  d.cam_prior = gen ? nullptr : A.upload(P->cam_prior, (size_t)3 * NC, e);
  d.cam_sigma = gen ? nullptr : A.upload(P->cam_sigma, (size_t)3 * NC, e);
  d.cam_fixed = gen ? nullptr : A.upload(P->cam_fixed, (size_t)NC, e);
  d.shot_camera = gen ? nullptr : A.upload(P->shot_camera, (size_t)S, e);
  d.cam_model = gen ? A.upload(G->cam_model, (size_t)NC, e) : (P->cam_model ? A.upload(P->cam_model, (size_t)NC, e) : nullptr);
  d.cam_ext = (!gen && P->cam_ext) ? A.upload(P->cam_ext, (size_t)16 * NC, e) : nullptr;
  d.shot_fixed = P->shot_fixed ? A.upload(P->shot_fixed, (size_t)S, e) : nullptr;
  d.point_fixed = P->point_fixed ? A.upload(P->point_fixed, (size_t)NP, e) : nullptr;
  d.gps = (!gen && P->shot_gps && P->shot_gps_sigma) ? A.upload(P->shot_gps, (size_t)3 * S, e) : nullptr;
  d.gps_sigma = (!gen && P->shot_gps && P->shot_gps_sigma) ? A.upload(P->shot_gps_sigma, (size_t)S, e) : nullptr;
  if (!gen && P->shot_up && P->shot_up_sigma) {
    std::vector<double> un((size_t)3 * S);
    for (int s = 0; s < S; s++) {
      const double *u = P->shot_up + 3 * (size_t)s;
      const double nrm = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
      OSFM_REQUIRE(!(P->shot_up_sigma[s] > 0) || nrm >= 1e-10, OSFM_E_INVALID, "UpVectorError: acceleration vector has near-zero magnitude");
      for (int i = 0; i < 3; i++) un[(size_t)3 * s + i] = nrm > 0 ? u[i] / nrm : 0.0;
    }
    d.up = A.upload(un.data(), (size_t)3 * S, e);
    d.up_sigma = A.upload(P->shot_up_sigma, (size_t)S, e);
    d.up_r = A.alloc<double>((size_t)3 * S, e);
    d.up_J = A.alloc<double>((size_t)9 * S, e);
    d.prior_rot = A.alloc<double>((size_t)6 * S, e);
  }
  // ---- point-major observation order and shot-major index lists, built on the device: perm = stable argsort of the observations by
  //      point (the order a host counting sort gives), shot_obs = stable argsort of the point-major positions by shot; offsets by
  //      lower bounds.  (The host version of this -- five passes with random scatters over 20 MB arrays -- was 35-40 ms of the 60 ms
  //      setup at configs[4].) ----
  std::vector<long> pt_off((size_t)NP + 1, 0);
  int *d_perm = A.alloc<int>((size_t)M, e);
  int bw_true = 0;
  int track_repeats_shot = 0;
  int *bp_keys = nullptr, *bp_pts = nullptr;  // the points in the order of the first shot of their track (band_mfma_kernel)
  if (M == 0) {  // (generic mode only) a problem of priors: empty segments everywhere
    long *d_pt_off = A.alloc<long>((size_t)NP + 1, e), *d_shot_off = A.alloc<long>((size_t)S + 1, e), *d_view_off = A.alloc<long>((size_t)g.NV + 1, e);
    OSFM_REQUIRE(e == hipSuccess, OSFM_E_NOMEM, "BA device allocation/upload failed: %s", hipGetErrorString(e));
    OSFM_HIP(hipMemsetAsync(d_pt_off, 0, ((size_t)NP + 1) * sizeof(long), sv.st));
    OSFM_HIP(hipMemsetAsync(d_shot_off, 0, ((size_t)S + 1) * sizeof(long), sv.st));
    OSFM_HIP(hipMemsetAsync(d_view_off, 0, ((size_t)g.NV + 1) * sizeof(long), sv.st));
    d.pt_off = d_pt_off;
    d.shot_off = d_shot_off;
    g.view_off = d_view_off;
    d.o_shot = d.o_point = d.shot_obs = A.alloc<int>(1, e);
    g.o_view = g.sm_view = d.o_shot;
  } else {
    int *raw_shot = A.upload(P->obs_shot, (size_t)M, e), *raw_point = A.upload(P->obs_point, (size_t)M, e);
    int *raw_view = gen ? A.upload(G->obs_view, (size_t)M, e) : nullptr, *o_view = gen ? A.alloc<int>((size_t)M, e) : nullptr;
    long *d_view_off = gen ? A.alloc<long>((size_t)g.NV + 1, e) : nullptr;
    int *iota = A.alloc<int>((size_t)M, e), *o_point = A.alloc<int>((size_t)M, e), *o_shot = A.alloc<int>((size_t)M, e);
    int *sm_keys = A.alloc<int>((size_t)M, e), *shot_obs = A.alloc<int>((size_t)M, e);
    long *d_pt_off = A.alloc<long>((size_t)NP + 1, e), *d_shot_off = A.alloc<long>((size_t)S + 1, e);
    int *d_bw = A.alloc<int>(4, e);
    auto bits_for = [](long n) { unsigned b = 1; while (b < 32 && (1L << b) < n) b++; return b; };
    const unsigned pbits = bits_for(NP), sbits = bits_for(gen ? g.NV : S);  // (the shot-major order is keyed by VIEW in the generic mode)
    int *first_shot = A.alloc<int>((size_t)NP, e);
    bp_keys = A.alloc<int>((size_t)NP, e);
    bp_pts = A.alloc<int>((size_t)NP, e);
    const unsigned fbits = bits_for((long)S + 1);
    size_t need1 = 0, need2 = 0, need3 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, need1, raw_point, o_point, iota, d_perm, (size_t)M, 0u, pbits, sv.st);
    (void)rocprim::radix_sort_pairs(nullptr, need2, o_shot, sm_keys, iota, shot_obs, (size_t)M, 0u, sbits, sv.st);
    (void)rocprim::radix_sort_pairs(nullptr, need3, first_shot, bp_keys, iota, bp_pts, (size_t)NP, 0u, fbits, sv.st);
    const size_t tb = std::max(std::max(need1, need2), need3);
    unsigned char *tmp = A.alloc<unsigned char>(tb + 256, e);
    OSFM_REQUIRE(e == hipSuccess, OSFM_E_NOMEM, "BA device allocation/upload failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(iota_int_kernel, dim3(nblk(M)), dim3(TPB), 0, sv.st, iota, M);
    size_t tb1 = tb, tb2 = tb;
    OSFM_HIP(rocprim::radix_sort_pairs(tmp, tb1, raw_point, o_point, iota, d_perm, (size_t)M, 0u, pbits, sv.st));
    hipLaunchKernelGGL(gather_int_kernel, dim3(nblk(M)), dim3(TPB), 0, sv.st, d_perm, raw_shot, M, o_shot);
    hipLaunchKernelGGL(lower_bound_kernel, dim3(nblk(NP + 1L)), dim3(TPB), 0, sv.st, o_point, M, NP, d_pt_off);
    if (gen) {  // views of an instance are consecutive: sorted by view is sorted by instance, and an instance's segment is its views'
      hipLaunchKernelGGL(gather_int_kernel, dim3(nblk(M)), dim3(TPB), 0, sv.st, d_perm, raw_view, M, o_view);
      OSFM_HIP(rocprim::radix_sort_pairs(tmp, tb2, o_view, sm_keys, iota, shot_obs, (size_t)M, 0u, sbits, sv.st));
      hipLaunchKernelGGL(lower_bound_kernel, dim3(nblk(g.NV + 1L)), dim3(TPB), 0, sv.st, sm_keys, M, g.NV, d_view_off);
      hipLaunchKernelGGL(gen_shot_off_kernel, dim3(nblk(S + 1L)), dim3(TPB), 0, sv.st, (const long *)d_view_off, g.inst_view0, S, d_shot_off);
      g.o_view = o_view;
      g.sm_view = sm_keys;
      g.view_off = d_view_off;
    } else {
      OSFM_HIP(rocprim::radix_sort_pairs(tmp, tb2, o_shot, sm_keys, iota, shot_obs, (size_t)M, 0u, sbits, sv.st));
      hipLaunchKernelGGL(lower_bound_kernel, dim3(nblk(S + 1L)), dim3(TPB), 0, sv.st, sm_keys, M, S, d_shot_off);
    }
    OSFM_HIP(hipMemsetAsync(d_bw, 0, 2 * sizeof(int), sv.st));
    hipLaunchKernelGGL(track_width_kernel, dim3(nblk(NP)), dim3(TPB), 0, sv.st, d_pt_off, o_shot, NP, d_bw);
    hipLaunchKernelGGL(track_first_kernel, dim3(nblk(NP)), dim3(TPB), 0, sv.st, d_pt_off, o_shot, NP, S, first_shot, d_bw + 1);
    if ((long)NP <= M) {  // iota holds 0 .. M - 1; a problem with more points than observations keeps the per-shot assembly
      size_t tb3 = tb;
      OSFM_HIP(rocprim::radix_sort_pairs(tmp, tb3, first_shot, bp_keys, iota, bp_pts, (size_t)NP, 0u, fbits, sv.st));
    } else
      bp_pts = nullptr;
    OSFM_HIP(hipGetLastError());
    OSFM_HIP(hipMemcpyAsync(pt_off.data(), d_pt_off, ((size_t)NP + 1) * sizeof(long), hipMemcpyDeviceToHost, sv.st));
    OSFM_HIP(hipMemcpyAsync(&bw_true, d_bw, sizeof(int), hipMemcpyDeviceToHost, sv.st));
    OSFM_HIP(hipMemcpyAsync(&track_repeats_shot, d_bw + 1, sizeof(int), hipMemcpyDeviceToHost, sv.st));
    OSFM_HIP(hipStreamSynchronize(sv.st));
    d.o_shot = o_shot;
    d.o_point = o_point;
    d.pt_off = d_pt_off;
    d.shot_off = d_shot_off;
    d.shot_obs = shot_obs;
  }
  {
    double *raw_xy = A.upload(P->obs_xy, (size_t)2 * M, e), *raw_sg = A.upload(P->obs_sigma, (size_t)M, e);
    double *ox = A.alloc<double>((size_t)M, e), *oy = A.alloc<double>((size_t)M, e), *osg = A.alloc<double>((size_t)M, e);
    if (e == hipSuccess && M > 0) hipLaunchKernelGGL(gather_pm_kernel, dim3(nblk(M)), dim3(TPB), 0, sv.st, d_perm, raw_xy, raw_sg, M, ox, oy, osg);
    d.o_x = ox;
    d.o_y = oy;
    d.o_sigma = osg;
    if (gen && G->obs_kind && M > 0) {
      unsigned char *raw_kind = A.upload(G->obs_kind, (size_t)M, e), *ok = A.alloc<unsigned char>((size_t)M, e), *sk = A.alloc<unsigned char>((size_t)M, e);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(gather_byte_kernel, dim3(nblk(M)), dim3(TPB), 0, sv.st, (const int *)d_perm, (const unsigned char *)raw_kind, M, ok);
        hipLaunchKernelGGL(gather_byte_kernel, dim3(nblk(M)), dim3(TPB), 0, sv.st, d.shot_obs, (const unsigned char *)ok, M, sk);
      }
      g.o_kind = ok;
      g.sm_kind = sk;
    }
  }
  {
    std::vector<int> wg_pt;
    wg_pt.push_back(0);
    int pcur = 0;
    while (pcur < NP) {
      int pe = pcur + 1;
      while (pe < NP && pe - pcur < kCoopObs && pt_off[(size_t)pe + 1] - pt_off[(size_t)pcur] <= kCoopObs) pe++;
      wg_pt.push_back(pe);
      pcur = pe;
    }
    if (wg_pt.size() < 2) wg_pt.push_back(0);  // (no points: one empty workgroup)
    d.nwg = (int)wg_pt.size() - 1;
    d.wg_pt = A.upload(wg_pt.data(), wg_pt.size(), e);
  }
  d.shotR = A.alloc<double>((size_t)36 * S, e);
  d.Jpm = A.alloc<double>((size_t)(gen ? g.ncomp : kRowComps) * M, e);
  d.sm_wt = A.alloc<double>((size_t)std::max<long>(1, M), e);
  {
    int *ss = A.alloc<int>((size_t)M, e), *sp = A.alloc<int>((size_t)M, e);
    double *sx = A.alloc<double>((size_t)M, e), *sy = A.alloc<double>((size_t)M, e), *ssg = A.alloc<double>((size_t)M, e);
    if (e == hipSuccess && M > 0)
      hipLaunchKernelGGL(gather_sm_kernel, dim3(nblk(M)), dim3(TPB), 0, sv.st, d.shot_obs, d.o_shot, d.o_point, d.o_x, d.o_y, d.o_sigma, M, ss,
                         sp, sx, sy, ssg);
    d.sm_shot = ss;
    d.sm_point = sp;
    d.sm_x = sx;
    d.sm_y = sy;
    d.sm_sigma = ssg;
  }
  d.w = A.alloc<double>((size_t)(gen ? g.NRr : 2) * M, e);
  d.g_pt = A.alloc<double>((size_t)3 * NP, e);
  d.Hpp = A.alloc<double>((size_t)6 * NP, e);
  d.Hhat = A.alloc<double>((size_t)6 * NP, e);
  d.sc_pt = A.alloc<double>((size_t)3 * NP, e);
  d.D_pt = A.alloc<double>((size_t)3 * NP, e);
  d.d_pt = A.alloc<double>((size_t)3 * NP, e);
  const size_t nr = (size_t)d.nred;
  d.g_red = A.alloc<double>(nr, e);
  d.diag_red = A.alloc<double>(nr, e);
  d.prior_diag = A.alloc<double>(nr, e);
  d.sc_red = A.alloc<double>(nr, e);
  d.D_red = A.alloc<double>(nr, e);
  d.Hcc = A.alloc<double>((size_t)21 * S + 6 * NC, e);
  d.Binv = A.alloc<double>((size_t)36 * S + 9 * NC + nbord, e);
  d.part = A.alloc<double>((size_t)9 * S, e);
  d.camred = A.alloc<double>((size_t)9 * NC, e);
  d.zc = A.alloc<double>(nr, e);
  d.y = A.alloc<double>(nr, e);
  d.x = A.alloc<double>(nr, e);
  d.r = A.alloc<double>(nr, e);
  d.z = A.alloc<double>(nr, e);
  d.p = A.alloc<double>(nr, e);
  d.Ap = A.alloc<double>(nr, e);
  d.b = A.alloc<double>(nr, e);
  d.scal = A.alloc<double>(32, e);
  const long nbmax = std::max<long>(std::max<long>(nblk(M), nblk(3L * NP)), d.nwg);
  d.partial = A.alloc<double>((size_t)2 * nbmax + 16, e);
  d.partial2 = A.alloc<double>((size_t)2 * nblk(3L * NP) + 16, e);
  d.dotp = A.alloc<double>((size_t)nblk(d.nred) + NC + 16, e);
  d.rrp = A.alloc<double>((size_t)nblk(d.nred) + 16, e);
  // block half-bandwidth of the shot-shot coupling (shots in caller order): bw_true, from track_width_kernel above
  // half-width up to 10: exact band, cyclic reduction in LDS; up to kWMaxBw: exact band, cyclic reduction over dense clusters (dbcr_*); beyond: truncated to kMaxBw
  // (round 6: the dense-cluster solver takes over where the LDS clusters end, at half-width 11 -- until round 5 at 16, and half-widths 11 .. 15 fell to
  //  the sequential band Cholesky WITHOUT the exact border: 180 - 244 CG iterations per LM iteration on a 27-shot scene with nine free cameras)
  constexpr int kLdsBw = 10;  // widest band whose clusters (6 cs unknowns, cs >= bw) the cyclic reduction holds in LDS
  const bool wide = O->preconditioner == 0 && S >= 2 && bw_true > kLdsBw && bw_true <= kWMaxBw && getenv("OSFM_BA_NO_WIDE") == nullptr;
  d.bw = O->preconditioner == 1 ? 0 : (wide ? bw_true : std::min(bw_true, kMaxBw));
  if (S < 2) d.bw = 0;
  // band columns per launch of the per-shot assembly: all of them when one copy fits a workgroup's LDS, else equal slices of at most kBandSlice
  const int band_nslice = (d.bw + 1 + kBandSlice - 1) / kBandSlice, band_slice = (d.bw + 1 + band_nslice - 1) / band_nslice;
  int band_copies = kBandCopies;  // private copies of the band assembly's LDS accumulators
  while (band_copies > 1 && (size_t)band_slice * 36 * band_copies * sizeof(double) > 150 * 1024) band_copies /= 2;
  d.band = A.alloc<double>((size_t)S * (d.bw + 1) * 36, e);
  // exact narrow band, one observation per (track, shot): assembled on the matrix cores (band_mfma_kernel), else per shot with LDS atomics
  const bool win_band = d.bw >= 1 && d.bw <= kMaxBw && d.bw == bw_true && !track_repeats_shot && bp_pts != nullptr && getenv("OSFM_BA_BAND_PER_SHOT") == nullptr &&
                        !(gen && g.NRr == 3);  // (it forms E from the two-row layout of the Jacobian copy)
  size_t win_lds = 0;
  int win_grid = 0;
  // the E blocks as an array: the per-shot assembly's operand only (the matrix-core assembly forms them from the Jacobian copy it reads)
  d.Epm = d.bw > 0 && (!win_band || getenv("OSFM_BA_CHECK_BAND") != nullptr) ? A.alloc<double>((size_t)18 * M, e) : nullptr;
  if (win_band) {
    const int TL = d.bw + 1;
    d.bpNT = (6 * TL + 15) / 16;
    d.bpMode = getenv("OSFM_BA_BM_MODE") ? atoi(getenv("OSFM_BA_BM_MODE")) : 0;
    d.bpPC = kBmSlots / TL;
    d.bpR = (int)std::min<long>(8, std::max<long>(1, ((long)NP + 128L * S - 1) / (128L * S)));
    win_grid = S * d.bpR;
    win_lds = (size_t)2 * 36 * kBmStride * sizeof(double);
    int *off = A.alloc<int>((size_t)S + 1, e), *o0 = A.alloc<int>((size_t)NP, e), *ln = A.alloc<int>((size_t)NP, e);
    unsigned char *tpos = A.alloc<unsigned char>((size_t)16 * NP, e);
    d.bp_part = A.alloc<double>((size_t)win_grid * (d.bpNT * (d.bpNT + 1) / 2) * 256, e);
    d.bp_pts = bp_pts;
    d.bp_off = off;
    d.bp_o0 = o0;
    d.bp_last = ln;
    d.bp_pos = tpos;
    if (e == hipSuccess) {
      hipLaunchKernelGGL(slab_bound_kernel, dim3(nblk(S + 1L)), dim3(TPB), 0, sv.st, bp_keys, NP, 1, S, off);
      hipLaunchKernelGGL(sorted_tracks_kernel, dim3(nblk(NP)), dim3(TPB), 0, sv.st, bp_pts, bp_keys, d.pt_off, d.o_shot, NP, o0, ln, tpos);
    }
  }
  // per-shot assembly with accumulators over the partners a shot really has (band_assemble_compact_kernel): tables once, at setup
  d.bslot = nullptr;
  int bslot_n = 0, bslot_copies = 1;
  if (d.bw > kMaxBw && !win_band && getenv("OSFM_BA_BAND_FULL_ROWS") == nullptr) {
    unsigned char *tab = A.alloc<unsigned char>((size_t)S * (d.bw + 1), e);
    int *d_mx = A.alloc<int>(1, e);
    OSFM_REQUIRE(e == hipSuccess, OSFM_E_NOMEM, "BA device allocation/upload failed: %s", hipGetErrorString(e));
    OSFM_HIP(hipMemsetAsync(d_mx, 0, sizeof(int), sv.st));
    hipLaunchKernelGGL(band_slots_kernel, dim3(S), dim3(TPB), (size_t)(d.bw + 1) * sizeof(int), sv.st, d, tab, d_mx);
    OSFM_HIP(hipMemcpyAsync(&bslot_n, d_mx, sizeof(int), hipMemcpyDeviceToHost, sv.st));
    OSFM_HIP(hipStreamSynchronize(sv.st));
    if (bslot_n >= 1 && bslot_n <= 64) {  // (more distinct partners than that: the full rows, in slices)
      d.bslot = tab;
      bslot_copies = kBandCopies;
      while (bslot_copies > 1 && (size_t)bslot_n * 36 * bslot_copies * sizeof(double) + d.bw + 1 > 72 * 1024) bslot_copies /= 2;  // two workgroups per CU
    }
  }
  d.dinv = A.alloc<double>((size_t)S * 36, e);
  d.cs = 0; d.ncl = 0; d.ncd = 0;
  if (d.bw >= 1 && d.bw <= kLdsBw && d.bw == bw_true && O->preconditioner == 0) {  // exact band, dense clusters fit LDS
    d.cs = d.bw < 2 ? 2 : d.bw;
    if (const char *ecs = getenv("OSFM_BA_CS")) d.cs = std::min(10, std::max(d.cs, atoi(ecs)));  // measurement knob: larger clusters (>= bw)
    d.ncd = 6 * d.cs;
    d.ncl = (S + d.cs - 1) / d.cs;
    const size_t nb = (size_t)d.ncl * d.ncd * d.ncd;
    d.cD = A.alloc<double>(nb, e);
    d.cW = A.alloc<double>(nb + (size_t)d.ncd * d.ncd, e);
    d.cWt = A.alloc<double>(nb + (size_t)d.ncd * d.ncd, e);
    d.cLi = A.alloc<double>(nb, e);
    d.cLit = A.alloc<double>(nb, e);
    d.bD = A.alloc<double>(nb, e);
    d.bE = A.alloc<double>(nb, e);
    d.bG = A.alloc<double>(nb, e);
    d.bH = A.alloc<double>(nb, e);
    d.bD2 = A.alloc<double>(nb, e);
    d.bGt = A.alloc<double>(nb, e);
    d.bHt = A.alloc<double>(nb, e);
    d.bx = A.alloc<double>((size_t)kWalkRhs * d.ncl * d.ncd, e);  // up to kWalkRhs right-hand sides at a time (the camera border's columns + the solve's)
    // few shots: the whole band in one workgroup's LDS (OSFM_BA_NO_SBAND keeps the cyclic reduction: the cross-check of the tests)
    sv.use_sband = sband_factor_lds(S, d.bw) <= kSbLdsMax && sband_solve_lds(S, d.bw) <= kSbLdsMax && getenv("OSFM_BA_NO_SBAND") == nullptr;
    if (sv.use_sband) sv.sbL = A.alloc<double>((size_t)S * (d.bw + 1) * 36, e);
  }
  d.wNB = 0; d.wWb = 0;
  d.qN = 0; d.qm = 0; d.qcs = 0;
  // wide band: cyclic reduction over dense clusters (dbcr_*); OSFM_BA_WIDE_LDLT keeps round 3's block LDL^T chain (measurement knob)
  bool dense_cr = wide && getenv("OSFM_BA_WIDE_LDLT") == nullptr;
  if (dense_cr) {
    // the dense-cluster blocks take ~9 (6 bw)^2 doubles per cluster plus the panel buffers, 2-3 x the LDL^T window's tiles: when the
    // device cannot hold them beside everything already allocated, the block LDL^T chain below (slower, a third of the memory) solves
    // the same band instead of the call failing with OSFM_E_NOMEM.  OSFM_BA_DENSE_CR_BUDGET (bytes) stands in for the free memory in tests.
    const size_t qm = (size_t)6 * d.bw, nq = (size_t)(S + d.bw - 1) / d.bw, np = (qm + kWB - 1) / kWB, qT = ((qm + np - 1) / np + 5) / 6 * 6;
    const size_t need = (nq * qm * qm * 9 + nq * qm * 8 + (nq / 2 + 1) * (3 * qT * qT + 3 * qT * qm)) * sizeof(double);
    size_t free_b = 0, total_b = 0;
    if (const char *bud = getenv("OSFM_BA_DENSE_CR_BUDGET")) free_b = (size_t)atoll(bud);
    else if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = ~(size_t)0;
    else free_b += ctx->pool_bytes;  // the context's cached blocks are not "free" to the runtime, but an allocation that fails takes them back
    if (need > free_b - free_b / 8) dense_cr = false;
  }
  if (dense_cr) {
    d.qcs = d.bw;
    d.qm = 6 * d.qcs;
    d.qN = (S + d.qcs - 1) / d.qcs;
    const size_t m2 = (size_t)d.qm * d.qm, nq = (size_t)d.qN;
    d.qD = A.alloc<double>(nq * m2, e);
    d.qE[0] = A.alloc<double>(nq * m2, e);
    d.qE[1] = A.alloc<double>(nq * m2, e);
    d.qX = A.alloc<double>(nq * 2 * m2, e);
    d.qXt = A.alloc<double>(nq * 2 * m2, e);
    d.qx = A.alloc<double>(nq * d.qm * 4, e);
    d.qy = A.alloc<double>(nq * d.qm * 4, e);
    {  // panels of the blocked inversion: as few as 96 columns allow, balanced, multiples of 6
      const int np = (d.qm + kWB - 1) / kWB;
      d.qT = ((d.qm + np - 1) / np + 5) / 6 * 6;
      const size_t nb = nq / 2 + 1;  // clusters of the largest level
      d.qP = A.alloc<double>(nb * d.qT * d.qT, e);
      d.qR = A.alloc<double>(nb * d.qT * d.qm, e);
      d.qRn = A.alloc<double>(nb * d.qT * d.qm, e);
      d.qC = A.alloc<double>(nb * d.qT * d.qm, e);
      d.qP2 = A.alloc<double>(nb * d.qT * d.qT, e);
      d.qBn = A.alloc<double>(nb * d.qT * d.qT, e);
    }
    // (measured in round 4 and left off: 131.3 against 117.5 ms for ten LM iterations on the 50 x 100 grid, 68.2 against 66.2 ms on ragged
    // tracks -- two cross-stream hand-overs per panel cost more than the 60 us of pivot inversion they hide, as with round 3's two-stream
    // LDL^T; OSFM_BA_LOOKAHEAD=1 turns it on, tests/test_gpu_ba.py keeps it correct)
    if (getenv("OSFM_BA_LOOKAHEAD") != nullptr && getenv("OSFM_BA_ONE_STREAM") == nullptr) {
      OSFM_HIP(hipStreamCreateWithFlags(&sv.st3, hipStreamNonBlocking));
      for (int q = 0; q < 2; q++) {
        OSFM_HIP(hipEventCreateWithFlags(&sv.ev_rn[q], hipEventDisableTiming));
        OSFM_HIP(hipEventCreateWithFlags(&sv.ev_p[q], hipEventDisableTiming));
      }
    }
  }
  if (wide && !dense_cr) {
    d.wNB = (S + kWcs - 1) / kWcs;
    d.wWb = std::min((d.bw + kWcs - 1) / kWcs, d.wNB - 1);
    const size_t nt = (size_t)d.wNB * (d.wWb + 1) * kWB * kWB;
    d.wA = A.alloc<double>(nt, e);
    d.wL = A.alloc<double>(nt, e);
    d.wLt = A.alloc<double>(nt, e);
    d.wDinv = A.alloc<double>((size_t)d.wNB * kWB * kWB, e);
    d.wx = A.alloc<double>((size_t)d.wNB * kWB * 4, e);  // up to 4 right-hand sides side by side
  }
  if ((d.ncl > 0 || wide) && nbord >= 1 && nbord <= (gen ? kGenMaxNB : 6)) {  // exact camera border: see border_rhs_kernel
    sv.Bc = A.alloc<double>((size_t)nbord * 6 * S, e);
    sv.Wb = A.alloc<double>((size_t)nbord * 6 * S, e);
    sv.SigInv = A.alloc<double>((size_t)nbord * nbord, e);
    sv.yb = A.alloc<double>((size_t)nbord + 8, e);
    sv.dots = A.alloc<double>((size_t)nbord * nbord, e);
    sv.dCm = A.alloc<double>((size_t)nbord * nbord, e);
    if (!gen) {
      sv.wB = A.alloc<double>((size_t)2 * 3 * NC * M, e);
      sv.partB = A.alloc<double>((size_t)S * 9 * NC, e);
    } else {
      std::vector<int> cols, col_pos((size_t)nbord, -1);
      for (int j = 0; j < nbord; j++) {
        bool held = false;
        for (int v = 0; v < G->NV && !held; v++) held = gen_col_slot_host[(size_t)j * G->NV + v] != 255;
        if (held) {
          col_pos[(size_t)j] = (int)cols.size();
          cols.push_back(j);
        }
      }
      sv.g_ncols = (int)cols.size();
      if (cols.empty()) cols.push_back(0);
      sv.g_cols = A.upload(cols.data(), cols.size(), e);
      sv.g_col_pos = A.upload(col_pos.data(), col_pos.size(), e);
      sv.g_wB = A.alloc<double>((size_t)std::max<long>(1, M) * std::max(1, sv.g_ncols) * g.NRr, e);
      sv.g_vpartB = A.alloc<double>((size_t)std::max(1, G->NV) * std::max(1, sv.g_ncols) * std::max(1, g.KW), e);
    }
  }
  bool border_ok = getenv("OSFM_BA_NO_BORDER") == nullptr;  // exact camera border: every camera free (generic mode: the border holds free blocks only)
  bool all_cams_fixed = !gen;
  for (int c = 0; c < NC && !gen; c++) {
    if (P->cam_fixed[c]) border_ok = false;
    else all_cams_fixed = false;
  }
  sv.cams_inert = all_cams_fixed;
  if (all_cams_fixed && e == hipSuccess) OSFM_HIP(hipMemsetAsync(d.camred, 0, (size_t)9 * NC * sizeof(double), sv.st));
  int *d_status = A.alloc<int>(4, e);
  double *d_reproj = (gen ? G->reproj3 != nullptr : P->reproj_err != nullptr) ? A.alloc<double>((size_t)(gen ? 3 : 2) * M, e) : nullptr;
  OSFM_REQUIRE(e == hipSuccess, OSFM_E_NOMEM, "BA device allocation/upload failed: %s", hipGetErrorString(e));
  OSFM_HIP(hipMemsetAsync(d.scal, 0, 32 * sizeof(double), sv.st));

  hipStream_t st = sv.st;
  const int nred = d.nred;
  const int nbr = nblk(nred);
  {
    const int rcp = sv.pinned(nbr);
    if (rcp != OSFM_OK) return rcp;
  }
  double *hs = sv.hscal;
  OSFM_HIP(hipStreamSynchronize(sv.st));
  const auto t_run = std::chrono::steady_clock::now();  // from here: what ceres::Solve would cover
  double cost = 0, sumsq = 0;
  int rc = sv.eval(d.cams, d.poses, d.pts, true, &cost, &sumsq);
  if (rc != OSFM_OK) return rc;
  if (getenv("OSFM_BA_DEBUG_NAN") != nullptr) {  // debugging aid: which of the first evaluation's arrays hold a NaN
    OSFM_HIP(hipStreamSynchronize(st));
    auto scan = [&](const char *name, const double *p, size_t n) {
      std::vector<double> h(n);
      (void)hipMemcpy(h.data(), p, n * sizeof(double), hipMemcpyDeviceToHost);
      size_t bad = 0;
      for (double x : h) bad += !(x == x);
      if (bad) fprintf(stderr, "[osfm_ba] %s: %zu of %zu not finite\n", name, bad, n);
    };
    scan("Jpm", d.Jpm, (size_t)(gen ? g.ncomp : kRowComps) * M);
    scan("sm_wt", d.sm_wt, (size_t)M);
    if (gen) {
      scan("PI", g.PI, (size_t)36 * S);
      scan("gpri", g.gpri, (size_t)nred);
    }
  }
  Rp->initial_cost = cost;
  Rp->seconds_setup = std::chrono::duration<double>(t_run - t_start).count();
  Rp->rmse_normalized_initial = std::sqrt(sumsq / (double)std::max<long>(1, gen ? G_rows0 : M));
  Rp->cost_history[0] = cost;
  const bool trace = getenv("OSFM_BA_TRACE") != nullptr;  // debugging aid: a line per phase with the stream drained, to see where a solve stalls
  auto mark = [&](const char *what) -> int {
    if (!trace) return OSFM_OK;
    fprintf(stderr, "[osfm_ba trace] %s ...", what);
    fflush(stderr);
    OSFM_HIP(hipStreamSynchronize(st));
    if (sv.st2) OSFM_HIP(hipStreamSynchronize(sv.st2));
    OSFM_HIP(hipGetLastError());
    fprintf(stderr, " done\n");
    return OSFM_OK;
  };
  if (mark("first evaluation") != OSFM_OK) return OSFM_E_HIP;
  double radius = O->initial_radius > 0 ? O->initial_radius : 1e4;
  double decrease_factor = 2.0;
  bool need_prepare = true, have_scale = false;
  bool fast_ok = getenv("OSFM_BA_NO_FAST") == nullptr;  // the straight-line iteration is tried (until it fails once in this solve)
  const int fast_fail_at = getenv("OSFM_BA_FAST_FAIL_AT") ? atoi(getenv("OSFM_BA_FAST_FAIL_AT")) : -1;
  int n_invalid = 0, iter = 0;
  double gmax = 0;
  Rp->termination = 0;
  double lin_seconds = 0;

  // after a Jacobian evaluation: gradients, (first time) the Jacobi scaling, the LM diagonal and max |gradient| into scal[10]
  auto prepare_enqueue = [&]() -> int {
    sv.gradients();
    if (!have_scale) {
      hipLaunchKernelGGL(scale_init_kernel, dim3(nblk(std::max<long>(nred, 3L * NP))), dim3(TPB), 0, st, d);
      have_scale = true;
    }
    if (gen) OSFM_HIP(hipMemsetAsync(d.scal + 10, 0, sizeof(double), st));  // (the [k1 k2 focal] mode: cleared by prior_cost_kernel, which every evaluation runs)
    if (std::max<long>(nred, 3L * NP) <= 65536) {  // small problems: the LM diagonal and max |gradient| in one launch (at configs[4] size the fused kernel was 34 us against 10 + 13)
      hipLaunchKernelGGL(lm_diag_absmax_kernel, dim3((unsigned)nblk(std::max<long>(nred, 3L * NP))), dim3(256), 0, st, d, d.scal + 10);
    } else {
      hipLaunchKernelGGL(lm_diag_kernel, dim3(nblk(std::max<long>(nred, 3L * NP))), dim3(TPB), 0, st, d);
      hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, st, d.g_red, (long)nred, d.g_pt, 3L * NP, d.scal + 10);
    }
    return OSFM_OK;
  };
  for (;;) {
    if (need_prepare) {
      const int rcp = prepare_enqueue();
      if (rcp != OSFM_OK) return rcp;
      {
        const int rcf = sv.fetch(d.scal + 10, 1, 0);
        if (rcf != OSFM_OK) return rcf;
      }
      gmax = hs[0];
      need_prepare = false;
    }
    if (iter >= O->max_iterations) { Rp->termination = 0; break; }
    if (gmax <= O->gradient_tolerance) { Rp->termination = 2; break; }
    if (radius < 1e-32) { Rp->termination = 4; break; }
    iter++;
    if (mark("gradients / scaling") != OSFM_OK) return OSFM_E_HIP;
    if (iter < 256) Rp->cost_history[iter] = cost;  // every exit below (tolerances, invalid step) leaves the slot of this iteration defined
    const auto t_lin = std::chrono::steady_clock::now();
    // ---- linear solve: PCG on the implicit Schur complement ----
    hipLaunchKernelGGL(point_hhat_kernel, dim3(nblk(NP)), dim3(TPB), 0, st, d, radius);
    sv.use_band = false;
    sv.cur_radius = radius;
    // ---- fork: camera-border columns (if this problem uses them) and the right-hand side only need the Jacobian and Hhat: they run on
    //      the side stream next to the band assembly (a gather that leaves HBM bandwidth unused), so that the cyclic-reduction levels
    //      -- workgroups that need a whole CU's LDS -- find the CUs free afterwards ----
    const bool want_border = d.bw > 0 && (d.ncl > 0 || wide) && O->preconditioner == 0 && sv.Bc && border_ok;
    // few shots, constant cameras: the one-workgroup factorisation carries the solve, so the right-hand side goes in front of it, on the main stream -- the
    // side stream has nothing to do in such an iteration (side2 = null)
    const bool sband_fuse = sv.use_sband && !gen && all_cams_fixed && d.bw > 0 && d.ncl > 0 && O->preconditioner == 0 && !want_border &&
                            getenv("OSFM_BA_NO_SBAND_FUSE") == nullptr;
    hipStream_t side2 = sband_fuse ? nullptr : sv.st2;
    hipStream_t sx = side2 ? side2 : st;
    // where the side stream starts: the per-shot assembly (LDS atomics) leaves HBM idle, so the border's passes run beside it; the
    // matrix-core assembly fills the CUs (four workgroups of 39 KB LDS each), and the side stream starts after it, beside the cyclic
    // reduction's levels -- one 117 KB workgroup per CU, which leaves the CU room for a border workgroup
    // (round 6: with the matrix-core assembly the side stream starts behind the FIRST LEVEL of the cyclic reduction -- its 278 workgroups of 117 KB
    //  LDS need every CU twice over, the later levels leave half of them and more to the border's kernels: 3.13 -> 3.09 ms per LM iteration at
    //  configs[4], profiles/r06_ba_variants.json; everything on one stream: 3.31)
    int fork_at = win_band ? (d.ncl > 1 && O->preconditioner == 0 ? 2 : 1) : 0;  // 0: before the assembly, 1: after it, 2: after the first level of the cyclic reduction
    sv.sband_solved = false;
    if (sband_fuse) {  // (three launches of ~17 us in all: on the main stream, in front of the assembly -- a fork and a join cost as much in event latencies)
      hipLaunchKernelGGL(schur_point_coop_kernel<1>, dim3(d.nwg), dim3(kCoopObs), 0, st, d, d.y);
      sv.schur_shot(st);
      hipLaunchKernelGGL(schur_finish_kernel, dim3(sv.matvec_parts()), dim3(TPB), 0, st, d, (const double *)d.x, (const double *)d.y, d.b, radius, 1, sv.cams_inert ? 1 : 0,
                         (double *)nullptr);
      fork_at = 0;
    }
    if (const char *fk = getenv("OSFM_BA_FORK")) fork_at = std::min(2, std::max(0, fk[0] - '0'));
    const bool fork_late = fork_at >= 1;
    if (side2 && !fork_late) {
      OSFM_HIP(hipEventRecord(sv.ev_fork, st));
      OSFM_HIP(hipStreamWaitEvent(side2, sv.ev_fork, 0));
    }
    if (d.bw > 0) {
      static OsfmPerDeviceOnce once;
      const int rca = once.run(ctx->device, []() -> int {
        OSFM_HIP(hipFuncSetAttribute((const void *)band_assemble_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        OSFM_HIP(hipFuncSetAttribute((const void *)wide_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return OSFM_OK;
      });
      if (rca != OSFM_OK) return rca;
      if (win_band) {
        static OsfmPerDeviceOnce once_w;
        const int rcw = once_w.run(ctx->device, []() -> int {
          OSFM_HIP(hipFuncSetAttribute((const void *)band_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          OSFM_HIP(hipFuncSetAttribute((const void *)band_mfma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          OSFM_HIP(hipFuncSetAttribute((const void *)band_mfma_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          OSFM_HIP(hipFuncSetAttribute((const void *)band_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          OSFM_HIP(hipFuncSetAttribute((const void *)band_mfma_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          OSFM_HIP(hipFuncSetAttribute((const void *)band_mfma_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          return OSFM_OK;
        });
        if (rcw != OSFM_OK) return rcw;
        switch (d.bpNT) {
          case 1: hipLaunchKernelGGL(band_mfma_kernel<1>, dim3(win_grid), dim3(384), win_lds, st, d); break;
          case 2: hipLaunchKernelGGL(band_mfma_kernel<2>, dim3(win_grid), dim3(384), win_lds, st, d); break;
          case 3: hipLaunchKernelGGL(band_mfma_kernel<3>, dim3(win_grid), dim3(384), win_lds, st, d); break;
          case 4: hipLaunchKernelGGL(band_mfma_kernel<4>, dim3(win_grid), dim3(384), win_lds, st, d); break;
          case 5: hipLaunchKernelGGL(band_mfma_kernel<5>, dim3(win_grid), dim3(384), win_lds, st, d); break;
          default: hipLaunchKernelGGL(band_mfma_kernel<6>, dim3(win_grid), dim3(384), win_lds, st, d); break;
        }
        hipLaunchKernelGGL(band_finish_kernel, dim3(nblk((long)S * (d.bw + 1) * 36)), dim3(TPB), 0, st, d, radius);
        if (getenv("OSFM_BA_CHECK_BAND") != nullptr) {  // self-check knob of the tests: the per-shot kernel must agree to rounding
          const size_t nbd = (size_t)S * (d.bw + 1) * 36;
          std::vector<double> b_win(nbd), b_shot(nbd);
          OSFM_HIP(hipMemcpyAsync(b_win.data(), d.band, nbd * sizeof(double), hipMemcpyDeviceToHost, st));
          hipLaunchKernelGGL(band_assemble_kernel, dim3(S), dim3(TPB), (size_t)(d.bw + 1) * 36 * band_copies * sizeof(double), st, d, radius, band_copies, 0,
                             d.bw + 1);
          OSFM_HIP(hipMemcpyAsync(b_shot.data(), d.band, nbd * sizeof(double), hipMemcpyDeviceToHost, st));
          OSFM_HIP(hipStreamSynchronize(st));
          double amax = 0, dmax = 0;
          for (size_t q = 0; q < nbd; q++) {
            amax = std::max(amax, std::fabs(b_shot[q]));
            dmax = std::max(dmax, std::fabs(b_shot[q] - b_win[q]));
          }
          OSFM_REQUIRE(dmax <= 1e-10 * amax, OSFM_E_NUMERIC, "band_mfma_kernel differs from band_assemble_kernel: max |diff| %.3e against max |entry| %.3e",
                       dmax, amax);
          OSFM_HIP(hipMemcpyAsync(d.band, b_win.data(), nbd * sizeof(double), hipMemcpyHostToDevice, st));
          OSFM_HIP(hipStreamSynchronize(st));
        }
      } else if (d.bslot) {
        static OsfmPerDeviceOnce once_c;
        const int rcc = once_c.run(ctx->device, []() -> int {
          OSFM_HIP(hipFuncSetAttribute((const void *)band_assemble_compact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          return OSFM_OK;
        });
        if (rcc != OSFM_OK) return rcc;
        hipLaunchKernelGGL(band_assemble_compact_kernel, dim3(S), dim3(TPB), (size_t)bslot_n * 36 * bslot_copies * sizeof(double) + ((d.bw + 1 + 15) / 16) * 16, st, d,
                           radius, bslot_copies, bslot_n);
      } else {
        for (int lo = 0; lo <= d.bw; lo += band_slice)
          hipLaunchKernelGGL(band_assemble_kernel, dim3(S), dim3(TPB), (size_t)std::min(band_slice, d.bw + 1 - lo) * 36 * band_copies * sizeof(double), st, d,
                             radius, band_copies, lo, std::min(band_slice, d.bw + 1 - lo));
      }
      sv.use_ctri = false;
      sv.use_bcr = false;
    }
    if (mark("band assembly") != OSFM_OK) return OSFM_E_HIP;
    const bool fork_in_bcr = fork_at == 2 && side2 && d.bw > 0 && d.ncl > 1 && O->preconditioner == 0;
    // the side stream's work: the camera border's columns and the right-hand side, from the point of the main stream where it is called
    auto side_work = [&](bool fork_here) -> int {
      if (side2 && fork_here) {
        OSFM_HIP(hipEventRecord(sv.ev_fork, st));
        OSFM_HIP(hipStreamWaitEvent(side2, sv.ev_fork, 0));
      }
      if (want_border && gen && getenv("OSFM_BA_BORDER_BY_MATVECS") == nullptr) {
        sv.gen_border_columns(radius, sx);
      } else if (want_border && gen) {  // (self-check knob) column j = the mat-vec of the unit vector e_(cam0 + j): its instance rows are B's, its border rows C's
        for (int j = 0; j < nbord; j++) {
          hipLaunchKernelGGL(unit_vec_kernel, dim3(nbr), dim3(TPB), 0, sx, d.p, nred, d.cam0 + j);
          sv.gen_matvec(d.p, d.Ap, radius, sx);
          hipLaunchKernelGGL(gen_border_store_kernel, dim3(nbr), dim3(TPB), 0, sx, (const double *)d.Ap, sv.Bc, sv.dCm, j, nbord, 6 * S);
        }
      } else if (want_border) {  // all nb columns of B (and of the camera block C) in one pass over the observations
        if (3 * NC == 3) {
          hipLaunchKernelGGL(border_point_kernel<3>, dim3(d.nwg), dim3(kCoopObs), 0, sx, d, sv.wB);
          if (sv.shot_waves() > 1) hipLaunchKernelGGL((border_shot_kernel<3, kShotWavesSmall>), dim3(S), dim3(64 * kShotWavesSmall), 0, sx, d, sv.wB, sv.Bc, sv.partB);
          else hipLaunchKernelGGL((border_shot_kernel<3, 1>), dim3(S), dim3(64), 0, sx, d, sv.wB, sv.Bc, sv.partB);
          hipLaunchKernelGGL(border_cam_kernel<3>, dim3(NC), dim3(TPB), 0, sx, d, sv.partB, sv.dCm, radius);
        } else {
          hipLaunchKernelGGL(border_point_kernel<6>, dim3(d.nwg), dim3(kCoopObs), 0, sx, d, sv.wB);
          if (sv.shot_waves() > 1) hipLaunchKernelGGL((border_shot_kernel<6, kShotWavesSmall>), dim3(S), dim3(64 * kShotWavesSmall), 0, sx, d, sv.wB, sv.Bc, sv.partB);
          else hipLaunchKernelGGL((border_shot_kernel<6, 1>), dim3(S), dim3(64), 0, sx, d, sv.wB, sv.Bc, sv.partB);
          hipLaunchKernelGGL(border_cam_kernel<6>, dim3(NC), dim3(TPB), 0, sx, d, sv.partB, sv.dCm, radius);
        }
      }
      // rhs
      if (sband_fuse) {
        // (done already, on the main stream)
      } else if (gen) {
        sv.gen_rows_apply(1, sx);
        hipLaunchKernelGGL(gen_schur_finish_kernel, dim3(nbr), dim3(TPB), 0, sx, d, (const double *)d.x, (const double *)d.y, d.b, radius, 1, M > 0 ? 1 : 0, 0, (double *)nullptr);
      } else {
        hipLaunchKernelGGL(schur_point_coop_kernel<1>, dim3(d.nwg), dim3(kCoopObs), 0, sx, d, d.y);
        sv.schur_shot(sx);
        hipLaunchKernelGGL(schur_finish_kernel, dim3(sv.matvec_parts()), dim3(TPB), 0, sx, d, (const double *)d.x, (const double *)d.y, d.b, radius, 1, sv.cams_inert ? 1 : 0,
                           (double *)nullptr);
      }
      if (side2) OSFM_HIP(hipEventRecord(sv.ev_join, side2));
      return OSFM_OK;
    };
    if (!fork_in_bcr) {
      const int rcs = side_work(fork_late);
      if (rcs != OSFM_OK) return rcs;
    }
    if (mark("side work (border columns, right-hand side)") != OSFM_OK) return OSFM_E_HIP;
    bool joined = side2 == nullptr;
    auto join = [&]() -> int {  // the main stream continues after the side stream's work
      if (!joined) OSFM_HIP(hipStreamWaitEvent(st, sv.ev_join, 0));
      joined = true;
      return OSFM_OK;
    };
    // The cyclic reduction (and the camera border on top of it) is issued without asking whether it succeeded: its status words come
    // back with the first scalars of PCG -- one host round trip for the factorisation, the border and the start of the solve.  When a
    // status says no (a pivot block that is not positive definite, a singular border) the fallbacks are issued and PCG starts again.
    const bool try_bcr = d.bw > 0 && d.ncl > 0 && O->preconditioner == 0;
    const bool try_border = (try_bcr || wide) && sv.Bc && border_ok;
    sv.use_bcr = false;
    sv.use_wide = false;
    sv.use_border = false;
    bool z_solved = false;  // M^-1 b went through the border's walk
    auto border_sigma = [&](int nb, int n6) -> int {  // Sigma^-1 = (C - B^T W)^-1 on the device; its status joins the factorisation's
      hipLaunchKernelGGL(border_dots_kernel, dim3(nb * nb), dim3(TPB), 0, st, sv.Bc, sv.Wb, nb, n6, sv.dots);
      if (nb <= 6) {
        hipLaunchKernelGGL(border_sigma_kernel, dim3(1), dim3(64), 0, st, sv.dCm, sv.dots, sv.SigInv, nb, d_status + 1);
      } else {
        static OsfmPerDeviceOnce once_s;
        const int rcs = once_s.run(ctx->device, []() -> int {
          OSFM_HIP(hipFuncSetAttribute((const void *)gen_border_sigma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));  // (the kernel also has two static words)
          return OSFM_OK;
        });
        if (rcs != OSFM_OK) return rcs;
        hipLaunchKernelGGL(gen_border_sigma_kernel, dim3(1), dim3(256), (size_t)nb * 2 * nb * sizeof(double), st, (const double *)sv.dCm, (const double *)sv.dots,
                           sv.SigInv, nb, d_status + 1);
      }
      return OSFM_OK;
    };
    if (wide) {
      if (dense_cr) {  // cyclic reduction over dense clusters: log2(S / bw) levels of batched dense operations
        const int rcq = sv.dbcr_factor(d_status);
        if (rcq != OSFM_OK) return rcq;
      } else {  // direct block LDL^T of the exact band: one launch per block column
        hipLaunchKernelGGL(wide_tiles_kernel, dim3(d.wNB, d.wWb + 1), dim3(256), 0, st, d, d_status);
        const int ntile = d.wWb * (d.wWb + 1) / 2;
        for (int J = 0; J < d.wNB; J++) {
          const int left = std::min(d.wWb, d.wNB - 1 - J);  // rows of the window that exist below block column J
          hipLaunchKernelGGL(wide_factor_kernel, dim3(1 + (left == d.wWb ? ntile : left * (left + 1) / 2)), dim3(256), (size_t)2 * kWB * kWLd * sizeof(double), st, d, J, d_status);
        }
      }
      sv.use_wide = true;
      if (try_border) {
        const int nb = nbord, n6 = 6 * S;
        const int rcj = join();
        if (rcj != OSFM_OK) return rcj;
        sv.wide_solve_set(RhsSet{sv.Bc, n6, sv.Wb, n6, nb + 1, d.b, d.z, nb, gen ? -1 : nb});  // the border's columns and the solve's own right-hand side
        z_solved = true;
        const int rcg = border_sigma(nb, n6);
        if (rcg != OSFM_OK) return rcg;
        sv.use_border = true;
      }
    }
    if (try_bcr) {
      const int N = d.ncl;
      if (sv.use_sband) {  // one workgroup factorises the band; the side stream's work runs beside it
        static OsfmPerDeviceOnce once;
        const int rca = once.run(ctx->device, []() -> int {
          OSFM_HIP(hipFuncSetAttribute((const void *)sband_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));  // (+ its two static tables)
          OSFM_HIP(hipFuncSetAttribute((const void *)sband_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          return OSFM_OK;
        });
        if (rca != OSFM_OK) return rca;
        if (fork_in_bcr) {
          const int rcs = side_work(true);
          if (rcs != OSFM_OK) return rcs;
        }
        if (sband_fuse) {  // the right-hand side must be there: the side stream was forked before the assembly
          const int rcj = join();
          if (rcj != OSFM_OK) return rcj;
          hipLaunchKernelGGL(sband_factor_kernel, dim3(1), dim3(kSbThreads), sband_factor_lds(S, d.bw), st, d, sv.sbL, d_status, RhsSet{d.b, 0, d.z, 0, 1, nullptr, nullptr, -1, 0},
                             SbFuse{1, radius, d.x, d.r, d.p, d.y, d.scal + 0, d.scal + 4, d.sc_red});
          sv.sband_solved = true;
        } else {
          hipLaunchKernelGGL(sband_factor_kernel, dim3(1), dim3(kSbThreads), sband_factor_lds(S, d.bw), st, d, sv.sbL, d_status, RhsSet{nullptr, 0, nullptr, 0, 0, nullptr, nullptr, -1, -1},
                             SbFuse{0, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});
        }
      } else {
      hipLaunchKernelGGL(bcr_build_kernel, dim3(N), dim3(256), 0, st, d, d_status);
      const BcrLaunch lv = bcr_level_for(d.cs);
      {
        static OsfmPerDeviceOnce once;
        const int rca = once.run(ctx->device, []() -> int {
          for (int q = 2; q <= 10; q++)
            OSFM_HIP(hipFuncSetAttribute((const void *)bcr_level_for(q).fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          return OSFM_OK;
        });
        if (rca != OSFM_OK) return rca;
      }
      for (int stq = 1; stq < N; stq *= 2) {
        hipLaunchKernelGGL(lv.fn, dim3((N + 2 * stq - 1) / (2 * stq)), dim3(lv.threads), lv.lds_bytes, st, d, stq, 0, d_status);
        if (stq == 1 && fork_in_bcr) {
          const int rcs = side_work(true);
          if (rcs != OSFM_OK) return rcs;
        }
      }
      hipLaunchKernelGGL(lv.fn, dim3(1), dim3(lv.threads), lv.lds_bytes, st, d, 1, 1, d_status);
      }
      sv.use_bcr = true;
      if (try_border) {  // exact camera border (few cameras, all free): B = S e_j restricted to the shot rows, W = A^-1 B
        const int nb = nbord, n6 = 6 * S;
        // the columns of B and C were formed on the side stream; W = A^-1 B for all of them in one walk of the levels
        const int rcj = join();
        if (rcj != OSFM_OK) return rcj;
        for (int q0 = 0; q0 < nb + 1; q0 += kWalkRhs) {  // (the work vectors of a walk hold kWalkRhs right-hand sides)
          const int cnt = std::min(kWalkRhs, nb + 1 - q0), qx = (nb >= q0 && nb < q0 + cnt) ? nb - q0 : -1;
          sv.bcr_solve_set(RhsSet{sv.Bc + (long)q0 * n6, n6, sv.Wb + (long)q0 * n6, n6, cnt, d.b, d.z, qx, gen ? -1 : qx});  // the border's columns and the solve's own right-hand side
        }
        z_solved = true;
        const int rcg = border_sigma(nb, n6);
        if (rcg != OSFM_OK) return rcg;
        sv.use_border = true;
      }
    }
    auto fallback_band = [&]() -> int {  // sequential banded block Cholesky (truncated band, or the cyclic reduction said no)
      const int R = d.bw + 1;
      {
        static OsfmPerDeviceOnce once;
        const int rca = once.run(ctx->device, []() -> int {
          OSFM_HIP(hipFuncSetAttribute((const void *)band_cholesky_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
          return OSFM_OK;
        });
        if (rca != OSFM_OK) return rca;
      }
      hipLaunchKernelGGL(band_cholesky_kernel, dim3(1), dim3(64), (size_t)((kMaxBw + 1) * R * 36 + (kMaxBw + 1) * 36 + 72) * sizeof(double), st, d, d_status);
      sv.hstat[3] = 1;
      OSFM_HIP(hipMemcpyAsync(sv.hstat + 3, d_status, sizeof(int), hipMemcpyDeviceToHost, st));
      OSFM_HIP(hipStreamSynchronize(st));
      sv.use_band = (sv.hstat[3] == 0);  // a truncated band may lose positive definiteness: fall back to block Jacobi
      if (sv.use_band && d.ncl > 0) {
        const size_t n2 = (size_t)d.ncd * d.ncd;
        OSFM_HIP(hipMemsetAsync(d.cD, 0, (size_t)d.ncl * n2 * sizeof(double), st));
        OSFM_HIP(hipMemsetAsync(d.cW, 0, (size_t)(d.ncl + 1) * n2 * sizeof(double), st));
        OSFM_HIP(hipMemsetAsync(d.cWt, 0, (size_t)(d.ncl + 1) * n2 * sizeof(double), st));
        hipLaunchKernelGGL(ctri_pad_kernel, dim3(1), dim3(64), 0, st, d);
        hipLaunchKernelGGL(ctri_scatterL_kernel, dim3(S), dim3(TPB), 0, st, d);
        {
          static OsfmPerDeviceOnce once;
          const int rca = once.run(ctx->device, []() -> int {
            OSFM_HIP(hipFuncSetAttribute((const void *)ctri_inverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            return OSFM_OK;
          });
          if (rca != OSFM_OK) return rca;
        }
        hipLaunchKernelGGL(ctri_inverse_kernel, dim3(d.ncl), dim3(64), 2 * n2 * sizeof(double), st, d);
        sv.use_ctri = true;
      }
      return OSFM_OK;
    };
    if (d.bw > 0 && !try_bcr && !wide) {
      const int rcf = fallback_band();
      if (rcf != OSFM_OK) return rcf;
    }
    {
      const int rcj = join();  // the right-hand side (and its use of part / camred) is complete
      if (rcj != OSFM_OK) return rcj;
    }
    if (mark("factorisation + border solve") != OSFM_OK) return OSFM_E_HIP;
    int *hst = sv.hstat;
    hst[0] = hst[1] = hst[2] = 0;
    auto start_pcg_enqueue = [&]() {
      // block-Jacobi blocks (6x6 per shot, 3x3 per camera): the fallback preconditioner, and the camera rows of the band
      // preconditioners -- not needed when the cyclic reduction came out with the exact camera border
      if (gen) {
        if (!(sv.use_bcr || sv.use_wide || sv.use_ctri || sv.use_band)) {
          if (g.NRr == 3) hipLaunchKernelGGL(gen_precond_shot_kernel<3>, dim3(S), dim3(64), 0, st, d, radius);
          else hipLaunchKernelGGL(gen_precond_shot_kernel<2>, dim3(S), dim3(64), 0, st, d, radius);
        }
      } else if (sv.use_bcr && sv.use_sband && !sv.use_border && all_cams_fixed && !z_solved) {
        // ... and with the one-workgroup band solve the camera blocks, the solve and the start of PCG are ONE launch (three until round 6) --
        // or none: the factorisation's launch has done it all
        if (sv.sband_solved) {
          sv.sband_solved = false;
          return;
        }
        hipLaunchKernelGGL(sband_solve_kernel, dim3(1), dim3(64 * kSbSolveWaves), sband_solve_lds(S, d.bw), st, d, (const double *)sv.sbL,
                           RhsSet{d.b, 0, d.z, 0, 1, nullptr, nullptr, -1, 0}, SbFuse{1, radius, d.x, d.r, d.p, d.y, d.scal + 0, d.scal + 4, d.sc_red});
        return;
      } else if ((sv.use_bcr || sv.use_wide) && all_cams_fixed) {
        // local / pose-only bundle adjustment: the band is the whole preconditioner and the camera rows are inert (zero scale, zero right-hand
        // side) -- their 3 x 3 blocks are the LM diagonal alone; the per-shot Schur blocks (63 us per LM iteration on a 48-shot problem) are not needed
        hipLaunchKernelGGL(precond_cam_kernel, dim3(nblk(NC, 64)), dim3(64), 0, st, d, radius);  // (camred: zeros since setup)
      } else if (!((sv.use_bcr || sv.use_wide) && sv.use_border)) {
        hipLaunchKernelGGL(precond_shot_kernel, dim3(S), dim3(64), 0, st, d, radius);
        hipLaunchKernelGGL(cam_reduce_kernel, dim3(NC), dim3(kCamRedT), 0, st, d, 6, (const double *)nullptr);
        hipLaunchKernelGGL(precond_cam_kernel, dim3(nblk(NC, 64)), dim3(64), 0, st, d, radius);
      }
      sv.precond(d.b, d.z, z_solved);
      z_solved = false;
      hipLaunchKernelGGL(pcg_init_kernel, dim3(1), dim3(1024), 0, st, d.b, d.z, d.x, d.r, d.p, nred, d.scal + 0, d.scal + 4, (const double *)d.sc_red, d.y);  // x = 0, r = b, p = z, y = sc p
    };
    auto start_pcg = [&]() -> int {
      start_pcg_enqueue();
      return sv.fetch(d.scal, 5, 0, d_status, (try_bcr || wide) ? 3 : 0, 0);
    };
    // one CG iteration's first half: the mat-vec and x += alpha p, r -= alpha Ap (r . r shares into rrp)
    auto pcg_half = [&](int rz_cur, bool leave_y = false) {  // leave_y: y = sc x behind the step (the straight-line iteration goes on to the back-substitution)
      sv.matvec(d.p, d.Ap, radius, true, d.dotp);
      hipLaunchKernelGGL(pcg_step1_kernel, dim3(nbr), dim3(TPB), 0, st, d.x, d.r, (const double *)d.p, (const double *)d.Ap, nred, (const double *)(d.scal + rz_cur),
                         (const double *)d.dotp, sv.matvec_parts(), d.scal + 1, d.rrp, (const double *)d.sc_red, leave_y ? d.y : (double *)nullptr);
    };
    auto swap_blocks = [&]() {
      std::swap(d.cams, d.cams_n);
      std::swap(d.poses, d.poses_n);
      std::swap(d.pts, d.pts_n);
      if (gen) {
        std::swap(g.cam, g.cam_n);
        std::swap(g.bias, g.bias_n);
        std::swap(g.rc, g.rc_n);
      }
    };
    auto relinearise_old_point = [&]() -> int {
      swap_blocks();
      sv.eval_enqueue(d.cams, d.poses, d.pts, true);
      return prepare_enqueue();  // (nothing to read: cost, sum of squares and max |gradient| of this point are on the host already)
    };
    // Back-substitution, model change, candidate -- and the candidate is LINEARISED before the host has seen the model change (round 6;
    // rounds 2-5 evaluated its cost alone here and, once the host had accepted the step, came back for the Jacobian): the blocks change
    // places, the evaluation with Jacobian rows, the gradients, the LM diagonal and max |gradient| are queued behind the back-substitution,
    // and ONE round trip brings the model change, the step's norms, the candidate's cost and its gradient norm.  An accepted step -- nearly
    // every step of a converging problem -- has then cost one evaluation instead of two (0.09 ms at configs[4]) and one round trip instead
    // of two; a rejected or invalid step puts the blocks back and linearises the old point again (the same kernels on the same inputs: the
    // same bits as before).
    auto candidate_enqueue = [&](bool y_ready = false) -> int {
      if (!y_ready) hipLaunchKernelGGL(scale_vec_kernel, dim3(nbr), dim3(TPB), 0, st, d.sc_red, d.x, d.y, nred);
      if (gen) {
        if (M > 0 && g.KW > 0) hipLaunchKernelGGL(gen_view_gather_kernel, dim3(nblk((long)g.NV * g.KW)), dim3(TPB), 0, st, d, (const double *)d.y);
        if (g.NRr == 3) hipLaunchKernelGGL((gen_schur_point_kernel<3, 2>), dim3(d.nwg), dim3(kCoopObs), 0, st, d, (const double *)d.y);
        else sv.gen_schur_point2(2, st);
      } else
        hipLaunchKernelGGL(schur_point_coop_kernel<2>, dim3(d.nwg), dim3(kCoopObs), 0, st, d, d.y);
      if (gen) hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(1024), 0, st, d.partial, (long)d.nwg, 1, d.scal + 16);  // the model change's observation part
      if (gen) {
        hipLaunchKernelGGL(gen_candidate_kernel, dim3(1), dim3(1024), 0, st, d, (const double *)d.y, d.scal + 16);
        hipLaunchKernelGGL(gen_prior_kernel, dim3(nblk(sv.gen_nprior(), kGenPriorTPB)), dim3(kGenPriorTPB), sv.gen_prior_lds(2), st, d, (const double *)g.cam, (const double *)g.bias,
                           (const double *)g.rc, (const double *)d.poses, 2, (const double *)d.y, d.scal + 16);
        if (g.pt_prior_sigma && NP > 0) hipLaunchKernelGGL(gen_point_prior_kernel, dim3(nblk(NP)), dim3(TPB), 0, st, d, (const double *)d.pts, 2, d.scal + 16);
      }
      if (gen) {
        hipLaunchKernelGGL(candidate_points_kernel, dim3(nblk(3L * NP)), dim3(TPB), 0, st, d, d.partial);
        hipLaunchKernelGGL(finish_reduce_kernel, dim3(1), dim3(1024), 0, st, d.partial, (long)nblk(3L * NP), 2, d.scal + 20);
      } else {  // the points' candidate first (its shares in partial2), then the cameras' and shots' with both sums and the candidate's rotation blocks
        hipLaunchKernelGGL(candidate_points_kernel, dim3(nblk(3L * NP)), dim3(TPB), 0, st, d, d.partial2);
        hipLaunchKernelGGL(candidate_kernel, dim3(1), dim3(1024), 0, st, d, (const double *)d.y, (const double *)d.partial, (long)d.nwg, d.scal + 16,
                           (const double *)d.partial2, (long)nblk(3L * NP), d.scal + 20, S <= 1024 ? 1 : 0);
      }
      swap_blocks();
      sv.eval_enqueue(d.cams, d.poses, d.pts, true, !gen && S <= 1024);
      return prepare_enqueue();
    };
    bool bad = false;
    int k = 0;
    double *rr_part = sv.hrr;
    const double *dec = hs;  // scal[8 .. 21] of the iteration's last round trip
    // ---- the straight-line iteration (round 6) ----
    // With the exact band and the exact border (or constant cameras) the preconditioner is the reduced matrix: CG is one mat-vec, and nothing
    // the host learns on the way -- the factorisation's status words, |b|, the residual after the first iterate -- changes what is launched
    // next, except in the rare failure.  So everything is queued back to back: the start of PCG, its first iteration, the back-substitution,
    // the candidate and its linearisation; ONE round trip brings the status words, the PCG scalars, the shares of r . r, the model change
    // and the candidate's cost.  When a status word or the residual says no, the blocks go back, the old point is linearised again and the
    // iteration is redone on the careful path below (which also keeps the fallbacks); after one such failure a solve stays on the careful
    // path.  Three round trips per LM iteration become one (OSFM_BA_NO_FAST keeps the careful path: the cross-check of the tests).
    bool fast_done = false;
    const bool exact_expected = (try_bcr || wide) && (try_border || (gen ? g.NB == 0 : all_cams_fixed)) && O->pcg_direct_tolerance > O->pcg_tolerance;
    if (exact_expected && fast_ok && !trace) {
      start_pcg_enqueue();
      pcg_half(0, true);
      rc = candidate_enqueue(true);
      if (rc != OSFM_OK) return rc;
      {
        const int rcf = sv.fetch(d.scal, 22, 0, d_status, 3, 0, d.rrp, nbr);
        if (rcf != OSFM_OK) return rcf;
      }
      const double bb1 = hs[4];
      double rr = 0.0;
      for (int q = 0; q < nbr; q++) rr += rr_part[(size_t)q];
      const bool status_ok = !((try_bcr && hst[0] != 0) || (sv.use_wide && hst[2] != 0) || (sv.use_border && hst[1] != 0));
      const double tol_first = std::max(O->pcg_tolerance, O->pcg_direct_tolerance);
      const bool forced_failure = fast_fail_at == iter;  // test knob OSFM_BA_FAST_FAIL_AT: the way back to the careful path, exercised on purpose
      if (!forced_failure && status_ok && bb1 == bb1 && !std::isinf(bb1) && bb1 > 0 && rr == rr && rr <= tol_first * tol_first * bb1) {
        k = 1;
        Rp->pcg_iterations_total += 1;
        dec = hs + 8;
        fast_done = true;
      } else {  // back to the old point; the factorisation, the border and the right-hand side stand (the status words are read again below)
        fast_ok = false;
        rc = relinearise_old_point();
        if (rc != OSFM_OK) return rc;
      }
    }
    if (!fast_done) {
    {
      const int rcs = start_pcg();
      if (rcs != OSFM_OK) return rcs;
    }
    if ((try_bcr && hst[0] != 0) || (sv.use_wide && hst[2] != 0) || (sv.use_border && hst[1] != 0)) {
      if (try_bcr && hst[0] != 0) {
        sv.use_bcr = false;
        sv.use_border = false;
        const int rcf = fallback_band();
        if (rcf != OSFM_OK) return rcf;
      } else if (sv.use_wide && hst[2] != 0) {  // a pivot block of the wide band is not positive definite: block Jacobi
        sv.use_wide = false;
        sv.use_border = false;
      } else {
        sv.use_border = false;
      }
      const int rcs = start_pcg();
      if (rcs != OSFM_OK) return rcs;
    }
    const double bb = hs[4];
    bad = !(bb == bb) || std::isinf(bb);
    if (!bad && bb > 0) {
      const double tol2 = O->pcg_tolerance * O->pcg_tolerance * bb;
      // the preconditioner is the reduced matrix itself: the first iterate is a direct solve (see osfm_ba_options_default)
      const bool exact_precond = (sv.use_bcr || sv.use_wide) && (sv.use_border || (gen ? g.NB == 0 : all_cams_fixed));
      const double tol2_first = exact_precond ? std::max(tol2, O->pcg_direct_tolerance * O->pcg_direct_tolerance * bb) : tol2;
      const int kmax = O->pcg_max_iterations > 0 ? O->pcg_max_iterations : 1000;
      // r.z lives in scal[0] and scal[2] alternately (rz_cur: the current one); p . Ap is added up from the mat-vec's shares by the step kernel
      int rz_cur = 0, rz_nxt = 2;
      for (k = 1; k <= kmax; k++) {
        pcg_half(rz_cur);
        // the convergence test comes before the preconditioner is applied to the new residual: the last iteration of a solve does not
        // pay for a walk of the cyclic reduction whose result nobody reads
        // (an exact band -- with the camera border on top, or with constant cameras as in local bundle adjustment -- makes the
        // preconditioner the matrix itself: CG is done after one or two iterations, so the first two are polled)
        if ((k & 3) == 0 || k == kmax || ((sv.use_bcr || sv.use_wide) && k <= 2)) {
          {
            const int rcf = sv.fetch(nullptr, 0, 0, nullptr, 0, 0, d.rrp, nbr);
            if (rcf != OSFM_OK) return rcf;
          }
          double rr = 0.0;
          for (int q = 0; q < nbr; q++) rr += rr_part[(size_t)q];
          if (!(rr == rr)) { bad = true; break; }
          if (rr <= (k == 1 ? tol2_first : tol2)) break;
        }
        sv.precond(d.r, d.z);
        hipLaunchKernelGGL(dot2_kernel, dim3(1), dim3(1024), 0, st, d.r, d.z, (const double *)nullptr, (const double *)nullptr, nred,
                           d.scal + rz_nxt, d.scal + 3);
        hipLaunchKernelGGL(pcg_step2_kernel, dim3(nbr), dim3(TPB), 0, st, d.p, (const double *)d.z, nred, (const double *)(d.scal + rz_cur), (const double *)(d.scal + rz_nxt),
                           (const double *)d.sc_red, d.y);
        std::swap(rz_cur, rz_nxt);
      }
      Rp->pcg_iterations_total += k;
    }
    if (mark("pcg") != OSFM_OK) return OSFM_E_HIP;
    if (trace) fprintf(stderr, "[osfm_ba trace] iteration %d: %d pcg iterations, status %d %d %d, band %d bcr %d (one workgroup: %d) wide %d dense %d border %d\n", iter, k, hst[0], hst[1],
                       hst[2], (int)sv.use_band, (int)sv.use_bcr, (int)(sv.use_bcr && sv.use_sband), (int)sv.use_wide, (int)(sv.use_wide && dense_cr), (int)sv.use_border);
    rc = candidate_enqueue();
    if (rc != OSFM_OK) return rc;
    {
      const int rcf = sv.fetch(d.scal + 8, 14, 0);  // scal[8..21]
      if (rcf != OSFM_OK) return rcf;
    }
    dec = hs;
    }  // (careful path)
    lin_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_lin).count();
    const double model_change = dec[8];
    const double step_sq = dec[9] + dec[12], x_sq = dec[10] + dec[13];
    if (bad || !(model_change > 0)) {  // HandleInvalidStep + StepIsInvalid
      radius *= 0.5;
      rc = relinearise_old_point();
      if (rc != OSFM_OK) return rc;
      if (++n_invalid >= 5) { Rp->termination = -1; break; }
      continue;
    }
    n_invalid = 0;
    const double cost_n = dec[0];
    const double step_norm = std::sqrt(step_sq), x_norm = std::sqrt(x_sq);
    if (step_norm <= O->parameter_tolerance * (x_norm + O->parameter_tolerance)) {
      Rp->termination = 3;
      swap_blocks();  // the step is not taken (the rows and gradients on the device are the candidate's: nothing reads them after the loop)
      break;
    }
    const double cost_change = cost - cost_n;
    if (std::fabs(cost_change) <= O->function_tolerance * cost) {
      Rp->termination = 1;
      swap_blocks();
      break;
    }
    const double rho = cost_change / model_change;
    if (O->verbose & 1)
      fprintf(stderr, "[osfm_ba] it %d cost %.9e -> %.9e rho %.3f radius %.3e pcg %d\n", iter, cost, cost_n, rho, radius, k);
    if (rho > 1e-3) {  // StepAccepted: the blocks are in place and the new point is linearised
      const double t = 2.0 * rho - 1.0;
      radius = radius / std::fmax(1.0 / 3.0, 1.0 - t * t * t);
      radius = std::fmin(1e16, radius);
      decrease_factor = 2.0;
      Rp->successful_steps++;
      cost = dec[0];
      sumsq = dec[1];
      gmax = dec[2];
    } else {  // StepRejected
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      rc = relinearise_old_point();
      if (rc != OSFM_OK) return rc;
    }
    if (iter < 256) Rp->cost_history[iter] = cost;
  }
  Rp->iterations = iter;
  Rp->final_cost = cost;
  Rp->seconds_linear_solver = lin_seconds;
  OSFM_HIP(hipStreamSynchronize(st));
  const auto t_tear = std::chrono::steady_clock::now();
  Rp->seconds_run = std::chrono::duration<double>(t_tear - t_run).count();
  Rp->preconditioner_bandwidth = (sv.use_band || sv.use_ctri || sv.use_bcr || sv.use_wide) ? d.bw : 0;
  Rp->shot_bandwidth = bw_true;
  Rp->ms_matvec_total = 0.0;
  Rp->matvec_calls = 0;
  // mat-vec timing sample (HIP events on the solver stream), for the roofline of the dominant kernel: ten extra mat-vecs, only when the
  // caller asks (OSFM_BA_TIME_MATVEC in options->verbose: the bench does) -- they were 3.9 ms of every configs[4] call's tear-down and as
  // much as a whole LM iteration of a local bundle adjustment
  if (O->verbose & OSFM_BA_TIME_MATVEC) {
    const int reps = 10;
    hipLaunchKernelGGL(point_hhat_kernel, dim3(nblk(NP)), dim3(TPB), 0, st, d, radius);
    OSFM_HIP(hipEventRecord(ctx->ev[6], st));
    for (int i = 0; i < reps; i++) sv.matvec(d.p, d.Ap, radius);
    OSFM_HIP(hipEventRecord(ctx->ev[7], st));
    OSFM_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    OSFM_HIP(hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7]));
    Rp->ms_matvec_total = ms;
    Rp->matvec_calls = reps;
  }
  // outputs: parameters, reprojection errors (sigma = 1), NaN/Inf check (ba_helpers.cc:780-814)
  sv.rot(d.poses);
  if (gen) {
    hipLaunchKernelGGL(gen_rc_rot_kernel, dim3(nblk(g.NRC, 64)), dim3(64), 0, st, d, (const double *)g.rc);
    if (d_reproj && M > 0) hipLaunchKernelGGL(gen_reproj_kernel, dim3(nblk(M)), dim3(TPB), 0, st, d, d_reproj);
    OSFM_HIP(hipMemcpyAsync(G->cam, g.cam, (size_t)16 * NC * sizeof(double), hipMemcpyDeviceToHost, st));
    OSFM_HIP(hipMemcpyAsync(G->bias, g.bias, (size_t)7 * NC * sizeof(double), hipMemcpyDeviceToHost, st));
    OSFM_HIP(hipMemcpyAsync(G->rc_pose, g.rc, (size_t)6 * g.NRC * sizeof(double), hipMemcpyDeviceToHost, st));
  } else {
    if (d_reproj) hipLaunchKernelGGL(reproj_kernel, dim3(nblk(M)), dim3(TPB), 0, st, d, d_reproj);
    OSFM_HIP(hipMemcpyAsync(P->cam_params, d.cams, (size_t)3 * NC * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  OSFM_HIP(hipMemcpyAsync(P->shot_pose, d.poses, (size_t)6 * S * sizeof(double), hipMemcpyDeviceToHost, st));
  if (NP > 0) OSFM_HIP(hipMemcpyAsync(P->points, d.pts, (size_t)3 * NP * sizeof(double), hipMemcpyDeviceToHost, st));
  if (d_reproj && gen && M > 0) {
    double *d_out = A.alloc<double>((size_t)3 * M, e);
    OSFM_REQUIRE(e == hipSuccess, OSFM_E_NOMEM, "BA device allocation failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(unpermute3_kernel, dim3(nblk(M)), dim3(TPB), 0, st, d_perm, d_reproj, M, d_out);
    OSFM_HIP(hipMemcpyAsync(G->reproj3, d_out, (size_t)3 * M * sizeof(double), hipMemcpyDeviceToHost, st));
  } else if (d_reproj && !gen) {  // back to the caller's observation order on the device, one contiguous copy
    double *d_out = A.alloc<double>((size_t)2 * M, e);
    OSFM_REQUIRE(e == hipSuccess, OSFM_E_NOMEM, "BA device allocation failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(unpermute2_kernel, dim3(nblk(M)), dim3(TPB), 0, st, d_perm, d_reproj, M, d_out);
    OSFM_HIP(hipMemcpyAsync(P->reproj_err, d_out, (size_t)2 * M * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  OSFM_HIP(hipStreamSynchronize(st));
  Rp->rmse_normalized_final = std::sqrt(sumsq / (double)std::max<long>(1, gen ? G_rows0 : M));
  Rp->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  Rp->seconds_teardown = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_tear).count();
  for (int i = 0; i < (gen ? 16 : 3) * NC; i++)
    OSFM_REQUIRE(std::isfinite(gen ? G->cam[i] : P->cam_params[i]), OSFM_E_NUMERIC, "camera has either NaN or INF values");
  for (long i = 0; i < 6L * S; i++) OSFM_REQUIRE(std::isfinite(P->shot_pose[i]), OSFM_E_NUMERIC, "shot pose has either NaN or INF values");
  for (long i = 0; i < 3L * NP; i++) OSFM_REQUIRE(std::isfinite(P->points[i]), OSFM_E_NUMERIC, "point has either NaN or INF values");
  return OSFM_OK;
}

#include "ba_generic_host.inc"
