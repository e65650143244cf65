// match.hip -- fused descriptor distance + top-2 + Lowe ratio + mutual check for gfx950 (MI355X).
//
// Replaces, per image pair, the cv2 calls under opensfm/matching.py:723-777
// (match_brute_force / match_brute_force_symmetric): an n1 x n2 x 128 distance computation, the
// two nearest neighbours of every feature in BOTH directions, the ratio test and the set
// intersection -- in ONE kernel, one workgroup per pair.  The n1 x n2 distance matrix never
// leaves the register file.
//
// Arithmetic (exact, integer): descriptors are integer-valued in [0,255] (features.py:526-534);
// stored as int8 a' = a - 128.  d^2(a,b) = |a'|^2 + |b'|^2 - 2 a'.b'  with a'.b' from
// v_mfma_i32_32x32x32_i8.  All quantities are exact int32, so the result is bit-identical to the
// fp32 computation cv2 performs (every partial sum < 2^24).
//
// Reduction: "best-only + lazy exact second" (see query_pass below).  The second-nearest neighbour is only needed for the
// ratio test, and only its VALUE; per query and class of 32 targets only the best accumulator value is kept, the norm of the
// target riding in the accumulator seed.  Classes are merged into (best, second-largest class best =: s_c); the true second s
// satisfies s >= s_c and the ratio test is monotone in d2: fails with s_c => fails (final, the vast majority); passes with
// s_c => the winner's own class is re-examined exactly with v_dot4 dot products.
// (Earlier generations -- exact top-2 in both directions 0.16, both directions per pass 0.24, one direction per pass with packed
// keys 0.39 of the int8 peak, profiles/r01_*, r02_bench_mid.json -- were removed; the exact VALU kernel below remains as the
// on-GPU cross-check.)
#include <cstdlib>

#include "osfm_internal.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#ifdef OSFM_DBG_PHASES
// instrumented builds only (tools/match_phases.py): 100 MHz ticks of workgroup thread 0, summed over the workgroups
__device__ unsigned long long g_phase[16];
#define OSFM_TICK(var) const unsigned long long var = wall_clock64();
#define OSFM_PHASE(i, t0, t1) if (tid == 0) atomicAdd(&g_phase[i], (t1) - (t0));
extern "C" int osfm_dbg_phases(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(g_phase)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#else
#define OSFM_TICK(var)
#define OSFM_PHASE(i, t0, t1)
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kRT = 2;                         // 32-row tiles per wave
constexpr int kNone = 0xFFFF;
constexpr int kCollisionD2 = 1 << 22;  // above this sqrtf() is no longer injective on integers

// Lowe ratio exactly as the reference evaluates it: float32 distances (cv2), compared in Python
// doubles: m.distance < ratio * n.distance (matching.py:752).  d^2 are exact ints < 2^24.
// Squared mode (ratio < 0 encodes it: -ratio is then float32(lowes_ratio^2)) is match_flann's test on SQUARED
// distances (matching.py:695-696): a numpy float32 array times a Python float stays float32, so
//     d0 < float32(ratio^2) * d1     with one float32 rounding of the product.
__device__ __forceinline__ bool ratio_ok(int d1sq, int d2sq, double ratio) {
  if (ratio < 0.0) return (float)d1sq < (float)(-ratio) * (float)d2sq;
  const float f1 = sqrtf((float)d1sq), f2 = sqrtf((float)d2sq);
  return (double)f1 < ratio * (double)f2;
}

__device__ __forceinline__ long xcd_remap(long b, long n) {
  // blocks are dealt round-robin to the 8 XCDs; give every XCD a contiguous range of pairs so
  // that the pairs in flight on one L2 share their first image (bijective for any n).
  const long q = n >> 3, r = n & 7;
  const long xcd = b & 7, within = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
}

struct MatchArgs {
  const int8_t *tiles;
  const int32_t *norms;
  const int32_t *hneg;  // -ceil(norm / 2): accumulator seeds
  const float *descf;   // float store: row (tile * 32 + r) x 128
  const float *qerr;    // float store: per image, max residual norm of the 8-bit quantisation (quantised units)
  const int64_t *tile_off;
  const int32_t *counts;
  const int32_t *pairs;
  long n_pairs;
  double ratio;      // Lowe ratio; NEGATIVE = squared mode: -ratio is float32(lowes_ratio^2) (see ratio_ok)
  int symmetric;
  int query_second;  // one-way matching with the SECOND image as the query set (match_flann(index1, f2), matching.py:683-697)
  int cap;
  int ncap;  // LDS capacity (features), multiple of 128
  int32_t *out_counts;
  uint32_t *out_matches;
  int32_t *out_flags;
  long pad_tile;            // index of the store's first slack tile
  // read by the exact / Hamming kernels only; kept behind the fused kernel's arguments so that its kernarg layout (and with it the
  // register allocation tests/test_kernel_budgets.py pins) does not move
  const uint32_t *bin;  // binary store: 16 dwords per row (tile * 32 + r)
  const float *seg;     // segmentation column (129th descriptor dimension, label x 35) per row, or null
};

// ---------------------------------------------------------------------------------------------
// Shared tail: ratio test on the column side, mutual check, ordered compaction.
// colBI[c] = row index of the best row for column c (or kNone), rowres[r] = best column for row r
// after the ratio test (or kNone).  Emits (c, r) sorted by c.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void emit_matches(const MatchArgs &a, long p, int nC, const int *colBI,
                                             const unsigned short *rowres, int *misc, int tid, bool swap_halves) {
  const int lane = tid & 63, w = tid >> 6;
  int base = 0;
  for (int j0 = 0; j0 < nC; j0 += kThreads) {
    const int j = j0 + tid;
    bool m = false;
    int r = kNone;
    if (j < nC) {
      r = colBI[j];
      m = (r != kNone) && (!a.symmetric || rowres[r] == j);
    }
    const unsigned long long bal = __ballot(m);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) misc[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < kWaves; ++w2) {
      const int cnt = misc[w2];
      woff += (w2 < w) ? cnt : 0;
      total += cnt;
    }
    if (m) {
      const int k = base + woff + prefix;
      // low half = feature of image 1, high half = feature of image 2
      if (k < a.cap) a.out_matches[p * a.cap + k] = swap_halves ? ((uint32_t)r | ((uint32_t)j << 16)) : ((uint32_t)j | ((uint32_t)r << 16));
    }
    base += total;
    __syncthreads();
  }
  if (tid == 0) a.out_counts[p] = base;
}

// a'.b' of two stored descriptors (tile layout): all 16 loads issued before the first use, one memory latency per candidate
__device__ __forceinline__ int dot_rows8(const int8_t *tilesA, int rowA, const int8_t *tilesB, int rowB) {
  const int8_t *pa = tilesA + (long)(rowA >> 5) * OSFM_TILE_BYTES + (rowA & 31) * 16;
  const int8_t *pb = tilesB + (long)(rowB >> 5) * OSFM_TILE_BYTES + (rowB & 31) * 16;
  v4i av[8], bv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    av[q] = *(const v4i *)(pa + q * 512);
    bv[q] = *(const v4i *)(pb + q * 512);
  }
  int s0 = 0, s1 = 0;
#pragma unroll
  for (int q = 0; q < 8; q += 2)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s0 = __builtin_amdgcn_sdot4(av[q][e], bv[q][e], s0, false);
      s1 = __builtin_amdgcn_sdot4(av[q + 1][e], bv[q + 1][e], s1, false);
    }
  return s0 + s1;
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
  return v;
}

// max over the 16 lanes of a DPP row, in every lane of the row (no LDS traffic): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ int row16_max(int x) {
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));
  x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));
  return x;
}

// ---------------------------------------------------------------------------------------------
// FQ mode of the fused kernel: float descriptors (root-SIFT ...) through the int8 matrix pipe with rigorous bounds.
// The store holds x^ = round(x) of x = (v - lo) * 255 / (hi - lo) - 128 next to the float rows, and per image E = max ||x - x^||.
// With D^ = ||x^_q - x^_t|| (an exact integer square root away) and D = scale * (the float32 distance cv2 computes),
//     | D - D^ |  <=  eps + kappa * D,      eps = E_Q + E_T   (triangle inequality; kappa covers the float32 roundings of the
//                                                               reference's accumulation: < 11 ulp on the squared sum)
// so the sweep's (best, class-level second) bounds decide most queries outright:
//     surely fails:   lower bound of the best   >= ratio * upper bound of the second     (the vast majority)
//     surely passes:  upper bound of the best   <  ratio * lower bound of the second, which also makes the integer winner the
//                     float winner (every other target is farther than the winner's upper bound)
// and whatever is left is evaluated in float32, operation for operation as the oracle / cv2 does (l2sqr_rows_f32), against every
// target whose quantised distance does not exclude it from the top two.  Results are bit-identical to the exact float kernel.
// ---------------------------------------------------------------------------------------------
constexpr double kFqSlack = 8e-6;
__device__ __forceinline__ double fq_ratio(double ratio) { return ratio < 0.0 ? sqrt(-ratio) : ratio; }  // squared mode compares d^2
__device__ __forceinline__ bool fq_surely_fails(int d0lo_sq, int d1hi_sq, double eps, double r) {
  const double D0lo = fmax(sqrt((double)max(d0lo_sq, 0)) - eps, 0.0) * (1.0 - kFqSlack);
  const double D1hi = (sqrt((double)max(d1hi_sq, 0)) + eps) * (1.0 + kFqSlack);
  return D0lo >= r * D1hi;
}
__device__ __forceinline__ bool fq_surely_passes(int d0hi_sq, int d1lo_sq, double eps, double r) {
  const double D0hi = (sqrt((double)max(d0hi_sq, 0)) + eps) * (1.0 + kFqSlack);
  const double D1lo = fmax(sqrt((double)max(d1lo_sq, 0)) - eps, 0.0) * (1.0 - kFqSlack);
  return D0hi < r * D1lo && D0hi < D1lo;
}
__device__ __forceinline__ float wave_minf(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
  return v;
}

// ---------------------------------------------------------------------------------------------
// Fused MFMA kernel: "one direction at a time".
//   pass A: the queries are the features of image A, the targets those of image B; resA[a] = best b if it passes the ratio test;
//   pass B: (symmetric matching only) the mutual check needs "best a for b" only for the b that some a chose: those candidates
//           (typically a small subset of image B) become the gathered queries of a second, much smaller pass against image A;
//   a pair (a, b) is emitted iff resA[a] == b and resB[b] == a.
// For symmetric matching A is the pair's SECOND image, so that the image whose row blocks every step re-reads is the first one,
// which the ~64 pairs in flight on an XCD share in L2 (xcd_remap).
// ---------------------------------------------------------------------------------------------
constexpr int kCT4 = 8;                              // a chunk stages 256 queries
constexpr int kChunkCols4 = kCT4 * 32;
constexpr int kChunkBytes4 = kCT4 * OSFM_TILE_BYTES;  // 32 KiB, double buffered

// ---------------------------------------------------------------------------------------------
// Version 5 of the pass: "targets in registers, queries in LDS, the norm in the accumulator".
//
// v4's pass is bounded by its integer epilogue (6.6 VALU per MFMA, profiles/r01_final_match_pmc.txt: VALU port 48 %, matrix pipe
// 45 %, and on gfx950 one integer VALU instruction of a 64-wide wave costs 5-6 cycles of a SIMD, profiles/r01_ubench_valu_issue.txt --
// more than the 32 cycles of the MFMA they follow).  Every one of those instructions builds or compares a key
// (2 a.b - |b|^2) * 2^k + tag.  v5 removes the key arithmetic:
//   * the QUERIES are the streamed operand (columns, one per lane), the TARGETS the register operand (rows, one per accumulator
//     register).  The quantity to maximise over the targets i of a query j is v = 2 a_i.b_j - |a_i|^2; the per-target constant now
//     belongs to a REGISTER, so it rides in the accumulator seed: the first MFMA of a K chain takes C = -ceil(|a_i|^2 / 2) (a
//     16-register tuple per row tile that stays put while the queries stream by), and the chain ends with
//     u = a_i.b_j - ceil(|a_i|^2 / 2), v = 2u + (|a_i|^2 & 1).  No shift, no add.
//   * per query and class (the 32 targets a lane sees in one step) only max u is kept: a v_max3 tree over the 32 accumulators,
//     then second = med3(best, m, second), class = m > best ? step : class, best = max(best, m): 20 VALU per 8 MFMAs.
//   * u orders v up to the parity bit, which is all the lazy scheme needs: the winner has u = max u; the other classes bound the
//     second neighbour between 2 s and 2 s + 1.  A query is final when the ratio test fails even with the most favourable of those
//     bounds (the vast majority); otherwise the winner's class (32 targets) is re-examined exactly with v_dot4, and if the answer
//     still depends on the unknown parity bit, or two classes tie, the query is re-done exactly against all targets.
//   * the chunk of 256 queries stays in LDS for the whole sweep over the targets, so there is no barrier, no DMA wait and no
//     merge inside the sweep: each wave streams its own target rows global -> VGPR one step ahead.
// Results are bit-identical to v4 / the exact kernel / the oracle (tests/test_gpu_matching.py).
// ---------------------------------------------------------------------------------------------
struct QueryPassShared {
  unsigned char *bbuf;  // [2][32 KiB] query chunks (tile layout); the drained one doubles as the merge scratch
  int *hbuf;            // [2][256] accumulator seeds of the row block in flight / the next one
};

// queries: slot q in [0, nslots) is feature qsel[q] of image Q (qsel == nullptr: identity); targets: all nT features of image T.
// out[query feature] = its nearest target if the ratio test passes, else kNone.  Returns the collision flag.
template <bool GATHER, bool FQ>
__device__ __forceinline__ int query_pass(const QueryPassShared &sh, const int8_t *tilesQ, const int32_t *normQ, int nQ, int nslots,
                                          const unsigned short *qsel, const int8_t *tilesT, const int32_t *normT, const int32_t *hnegT,
                                          int nT, const int8_t *tiles_pad, const int32_t *hneg_pad, unsigned short *out, double ratio,
                                          int tid, const float *descQ, const float *descT, double eps) {
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tT = (nT + 31) >> 5;
  const int tS = (nslots + 31) >> 5;
  const int nchunks = (tS + kCT4 - 1) / kCT4;
  const int nrb = (tT + kWaves * kRT - 1) / (kWaves * kRT);
  int flag = 0;

  // one chunk = 8 query tiles, global -> LDS by DMA; thread (w, lane) moves bytes [w*1024 + lane*16, +16) of every tile, i.e. the
  // K slice w of feature (lane & 31), half (lane >> 5) -- which is also how a gathered query is addressed
  auto dma_chunk = [&](int c) {
    unsigned char *dst = sh.bbuf + (c & 1) * kChunkBytes4;
#pragma unroll
    for (int q = 0; q < kCT4; ++q) {
      const int gt = c * kCT4 + q;
      const int8_t *src;
      if (GATHER) {
        const int slot = gt * 32 + (lane & 31);
        const int f = slot < nslots ? (int)qsel[slot] : 0;
        src = tilesQ + (long)(f >> 5) * OSFM_TILE_BYTES + w * 1024 + (lane >> 5) * 512 + (f & 31) * 16;
      } else {
        src = tilesQ + (long)(gt < tS ? gt : 0) * OSFM_TILE_BYTES + tid * 16;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(dst + q * OSFM_TILE_BYTES + w * 1024), 16, 0, 0);
    }
  };

  // the wave's two target row tiles of row block rb.  Branch-free: a row tile beyond the image reads the store's slack tile (zero
  // descriptors, padding norms), whose seed -2^22 can never be a maximum.
  //   A operands: global -> VGPR, one step ahead, into the register set the running step does not use;
  //   accumulator seeds: global -> LDS by DMA (each wave stages the 64 seeds of its own two row tiles, so only its own vmcnt
  //   orders them), read into the seed tuples as soon as the running step has issued its last seeded MFMA.
  auto load_targets = [&](int rb, v4i (&af)[kRT][4], int slot) {
    {
      const int t = rb * (kWaves * kRT) + w * kRT + (lane >> 5);
      const int32_t *hp = ((t < tT) ? hnegT + (long)t * 32 : hneg_pad) + (lane & 31);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)hp,
                                       (__attribute__((address_space(3))) void *)(sh.hbuf + slot * 256 + w * 64), 4, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt) {
      const int t = rb * (kWaves * kRT) + w * kRT + rt;  // wave-uniform
      const int8_t *tp = (t < tT) ? tilesT + (long)t * OSFM_TILE_BYTES : tiles_pad;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[rt][ks] = *(const v4i *)(tp + ks * 1024 + lane * 16);
    }
  };
  v16i hn[kRT];
  // accumulator register r of half h belongs to row (r & 3) + 8 (r >> 2) + 4 h of the tile
  auto read_seeds = [&](int slot) {
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const v4i hv = *(const v4i *)(sh.hbuf + slot * 256 + w * 64 + rt * 32 + 8 * g + 4 * (lane >> 5));
        hn[rt][4 * g + 0] = hv[0];
        hn[rt][4 * g + 1] = hv[1];
        hn[rt][4 * g + 2] = hv[2];
        hn[rt][4 * g + 3] = hv[3];
      }
  };

  int cb[kCT4], cs[kCT4], ci[kCT4];
  const unsigned char *bb = sh.bbuf;
  v4i bf[4];  // B operands of the tile in flight; slice ks is refilled for the next tile as soon as both chains have consumed it

  // max over the 16 accumulators of one chain: two interleaved v_max3 chains
#define OSFM_TREE_A(X, A, B)             \
  A = max(max(X[0], X[1]), X[2]);        \
  B = max(max(X[8], X[9]), X[10]);
#define OSFM_TREE_B(X, A, B)             \
  A = max(max(A, X[3]), X[4]);           \
  B = max(max(B, X[11]), X[12]);
#define OSFM_TREE_C(X, A, B)             \
  A = max(max(A, X[5]), X[6]);           \
  B = max(max(B, X[13]), X[14]);
#define OSFM_TREE_D(X, A, B, M)          \
  M = max(max(A, B), X[7]);              \
  M = max(M, X[15]);
#define OSFM_PIN __builtin_amdgcn_sched_barrier(0);

  // One step = this wave's 64 targets (af, hn) against the 8 query tiles of the chunk: 64 MFMAs.  Per tile two K chains (one per row
  // tile) run interleaved on two of THREE accumulator tuples; the tuples of the previous tile are reduced in the issue gaps of the
  // current one (an in-order wavefront hides about five plain VALU instructions behind a 32-cycle MFMA, MI355X_MICROARCH.md), and the
  // first chain's tuple of tile t-1 becomes the second chain's tuple of tile t.  The class update of tile t-1 rides in the later gaps.
  auto step = [&](const v4i (&af)[kRT][4], int rb, int next_slot, bool chunk_dma_in_flight) {
    v16i T[3];
    int mp = 0;
#pragma unroll
    for (int t = 0; t < kCT4; ++t) {
      v16i &C1 = T[(2 * t) % 3];
      v16i &C2 = T[(2 * t + 1) % 3];      // = chain 1 of tile t-1: reduced in the first gap
      v16i &P2 = T[(2 * t + 2) % 3];      // = chain 2 of tile t-1
      const unsigned char *nb = bb + ((t + 1) & (kCT4 - 1)) * OSFM_TILE_BYTES + lane * 16;  // tile 7 prefetches tile 0 (next step)
      int a0 = 0, b0 = 0, a1 = 0, b1 = 0, m2 = 0;
      C1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0][0], bf[0], hn[0], 0, 0, 0);
      OSFM_PIN
      if (t > 0) {
        OSFM_TREE_A(C2, a0, b0)
        OSFM_TREE_B(C2, a0, b0)
        OSFM_TREE_C(C2, a0, b0)
        OSFM_TREE_D(C2, a0, b0, mp)
      }
      OSFM_PIN
      C2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[1][0], bf[0], hn[1], 0, 0, 0);
      OSFM_PIN
      if (t == kCT4 - 1) {
        // the step's last seeded MFMAs are out: fetch the next step's seeds.  Their DMA is older than this step's 8 A-operand loads
        // (and than the 8 DMAs of the next chunk, when those were issued in this step)
        if (chunk_dma_in_flight)
          __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * kRT * 4 + kCT4));  // vmcnt(16)
        else
          __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * kRT * 4));  // vmcnt(8)
        read_seeds(next_slot);
      }
      bf[0] = *(const v4i *)(nb);
      if (t > 0) { OSFM_TREE_A(P2, a1, b1) }
      OSFM_PIN
      C1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0][1], bf[1], C1, 0, 0, 0);
      OSFM_PIN
      if (t > 0) { OSFM_TREE_B(P2, a1, b1) }
      OSFM_PIN
      C2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[1][1], bf[1], C2, 0, 0, 0);
      OSFM_PIN
      bf[1] = *(const v4i *)(nb + 1024);
      if (t > 0) { OSFM_TREE_C(P2, a1, b1) }
      OSFM_PIN
      C1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0][2], bf[2], C1, 0, 0, 0);
      OSFM_PIN
      if (t > 0) { OSFM_TREE_D(P2, a1, b1, m2) }
      OSFM_PIN
      C2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[1][2], bf[2], C2, 0, 0, 0);
      OSFM_PIN
      bf[2] = *(const v4i *)(nb + 2048);
      if (t > 0) {
        m2 = max(m2, mp);
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cs[t - 1]) : "v"(cb[t - 1]), "v"(m2), "v"(cs[t - 1]));  // cs <= cb: the second-largest of {cb, m, cs}
      }
      OSFM_PIN
      C1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0][3], bf[3], C1, 0, 0, 0);
      OSFM_PIN
      if (t > 0) ci[t - 1] = m2 > cb[t - 1] ? rb : ci[t - 1];
      OSFM_PIN
      C2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[1][3], bf[3], C2, 0, 0, 0);
      OSFM_PIN
      bf[3] = *(const v4i *)(nb + 3072);
      if (t > 0) cb[t - 1] = max(cb[t - 1], m2);
      OSFM_PIN
    }
    {  // drain: the last tile's tuples
      v16i &Q1 = T[(2 * (kCT4 - 1)) % 3];
      v16i &Q2 = T[(2 * (kCT4 - 1) + 1) % 3];
      int a0, b0, a1, b1, m1, m2;
      OSFM_TREE_A(Q1, a0, b0)
      OSFM_TREE_B(Q1, a0, b0)
      OSFM_TREE_C(Q1, a0, b0)
      OSFM_TREE_D(Q1, a0, b0, m1)
      OSFM_TREE_A(Q2, a1, b1)
      OSFM_TREE_B(Q2, a1, b1)
      OSFM_TREE_C(Q2, a1, b1)
      OSFM_TREE_D(Q2, a1, b1, m2)
      m2 = max(m2, m1);
      asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cs[kCT4 - 1]) : "v"(cb[kCT4 - 1]), "v"(m2), "v"(cs[kCT4 - 1]));
      ci[kCT4 - 1] = m2 > cb[kCT4 - 1] ? rb : ci[kCT4 - 1];
      cb[kCT4 - 1] = max(cb[kCT4 - 1], m2);
    }
  };

  dma_chunk(0);
  // two register sets for the A operands: the step that computes on one set fetches the next step's targets into the other
  v4i afA[kRT][4], afB[kRT][4];
  load_targets(0, afA, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): chunk 0, the first targets and their seeds
  read_seeds(0);
  int sidx = 0;  // global step counter: the seeds of step s are staged in slot s & 1

  for (int c = 0; c < nchunks; ++c) {
    // the query this thread decides at the end of the chunk (slot c*256 + tid): its norm, fetched now
    const int myslot = c * kChunkCols4 + tid;
    const int myf = myslot < nslots ? (GATHER ? (int)qsel[myslot] : myslot) : -1;
    const int myna = (myf >= 0 && myf < nQ) ? normQ[myf] : OSFM_PAD_NORM;
    // chunk c was requested during the first step of chunk c - 1 (chunk 0: above); everything this thread has in flight is older
    // than what the coming step will issue
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    bb = sh.bbuf + (c & 1) * kChunkBytes4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bf[ks] = *(const v4i *)(bb + ks * 1024 + lane * 16);
#pragma unroll
    for (int t = 0; t < kCT4; ++t) {
      cb[t] = INT_MIN;
      cs[t] = INT_MIN;
      ci[t] = 0;
    }

    // steps come in pairs: the first computes on set A and fetches into set B, the second the other way round; an odd last step
    // computes on A and then moves B (the next chunk's first targets) to A, so that every chunk starts on set A
    auto one_step = [&](const v4i (&cur)[kRT][4], v4i (&nxt)[kRT][4], int rb) {
      const int rbn = (rb + 1 < nrb) ? rb + 1 : 0;  // the next chunk starts over at row block 0 (the very last prefetch is unused)
      const bool live = rb * (kWaves * kRT) + w * kRT < tT;
      const bool dma = (rb == 0) && (c + 1 < nchunks);
      const int nslot = (sidx + 1) & 1;
      load_targets(rbn, nxt, nslot);
      if (dma) dma_chunk(c + 1);  // its buffer was released by the merge of chunk c - 1
      if (live) {
        step(cur, rb, nslot, dma);
      } else {  // a wave without targets in this row block still has to pick up the next step's seeds
        __builtin_amdgcn_s_waitcnt(0x0F70);
        read_seeds(nslot);
      }
      ++sidx;
    };
    OSFM_TICK(tk0)
    int rb = 0;
    for (; rb + 1 < nrb; rb += 2) {
      one_step(afA, afB, rb);
      one_step(afB, afA, rb + 1);
    }
    if (rb < nrb) {
      one_step(afA, afB, rb);
#pragma unroll
      for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) afA[rt][ks] = afB[rt][ks];
    }

    // ---- end of the chunk: merge the 8 partial classes (4 waves x 2 halves) of every query, decide, re-examine ----
    OSFM_TICK(tk1)
    __syncthreads();  // everyone is done reading the chunk: its buffer becomes the scratch
    OSFM_TICK(tk2)
    int *scr = (int *)(sh.bbuf + (c & 1) * kChunkBytes4);  // [8 parts][3][256]
    {
      const int part = w * 2 + (lane >> 5);
#pragma unroll
      for (int t = 0; t < kCT4; ++t) {
        scr[(part * 3 + 0) * kChunkCols4 + t * 32 + (lane & 31)] = cb[t];
        scr[(part * 3 + 1) * kChunkCols4 + t * 32 + (lane & 31)] = cs[t];
        scr[(part * 3 + 2) * kChunkCols4 + t * 32 + (lane & 31)] = ci[t];
      }
    }
    __syncthreads();
    int b = INT_MIN, s2 = INT_MIN, id = 0;
#pragma unroll
    for (int part = 0; part < 2 * kWaves; ++part) {
      const int pb = scr[(part * 3 + 0) * kChunkCols4 + tid], ps = scr[(part * 3 + 1) * kChunkCols4 + tid];
      const int pi = scr[(part * 3 + 2) * kChunkCols4 + tid];
      s2 = max(max(s2, ps), min(b, pb));
      id = pb > b ? (pi * 8 + part) : id;
      b = max(b, pb);
    }
    __syncthreads();  // scratch read: the buffer may be refilled by the next chunk's prefetch (issued at the top of the loop)
    bool want = false;
    if (myf >= 0 && myf < nQ) {
      // v_best in {2b, 2b+1}; the best of the other classes in {2 s2, 2 s2 + 1}, the true second is at least that
      const int d1lo = max(myna - (2 * b + 1), 0), d2hi = myna - 2 * s2;
      if (FQ) {
        want = !fq_surely_fails(d1lo, d2hi, eps, fq_ratio(ratio));
      } else {
        if (d2hi >= kCollisionD2 && ratio >= 0.0) flag = 1;  // squared mode never takes a square root
        want = ratio_ok(d1lo, d2hi, ratio);
      }
#ifdef OSFM_DBG_NORECHECK
      want = false;
#endif
      if (!want) out[myf] = kNone;
    }
    // ---- re-examination, four queries at a time: one query per 16-lane row, two of the winner class's 32 targets per lane.  The
    //      candidates travel as packed keys v * 32 + (31 - ordinal) (ordinal = position in ascending target order), so one max
    //      gives the winner with cv2's lowest-index rule and a second max the runner-up; both are DPP reductions inside the row.
    unsigned long long pending = __ballot(want);
    unsigned long long redo = 0;  // queries that have to be re-done against all targets
    OSFM_TICK(tk3)
#ifdef OSFM_DBG_PHASES
    if (tid == 0) atomicAdd(&g_phase[(GATHER ? 5 : 0) + 4], (unsigned long long)__popcll(pending));
#endif
    while (pending) {
      int srcs[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        srcs[g] = pending ? (int)__builtin_ctzll(pending) : -1;
        pending &= pending - 1;  // 0 stays 0
      }
      const int g = lane >> 4, sl = lane & 15;
      const int src = g == 0 ? srcs[0] : g == 1 ? srcs[1] : g == 2 ? srcs[2] : srcs[3];
      const int ssrc = src < 0 ? 0 : src;
      const int qf = __shfl(myf, ssrc), qna = __shfl(myna, ssrc), qb = __shfl(b, ssrc), qs = __shfl(s2, ssrc), qid = __shfl(id, ssrc);
      const bool tie = (qs == qb);  // two classes tie on u: the winner may sit in either
      const int qrb = qid >> 3, qw = (qid >> 1) & 3, qh = qid & 1;
      const int row0 = (qrb * (kWaves * kRT) + qw * kRT) * 32 + (sl & 3) + 8 * (sl >> 2) + 4 * qh, row1 = row0 + 32;
      int k0 = INT_MIN, k1 = INT_MIN;
      if (src >= 0 && !tie) {
        // a row beyond the image reads row 0 instead (its key is dropped below): the ADDRESS is selected, never the loaded value, so
        // that all 26 loads are in flight before the first wait (a value select made the compiler wait after every query slice:
        // ten memory round trips per step, profiles/r03_match_phases.txt)
        const bool ok0 = row0 < nT, ok1 = row1 < nT;
        const int r0 = ok0 ? row0 : 0, r1 = ok1 ? row1 : 0;
        const int8_t *pa = tilesQ + (long)(qf >> 5) * OSFM_TILE_BYTES + (qf & 31) * 16;
        const int8_t *p0 = tilesT + (long)(r0 >> 5) * OSFM_TILE_BYTES + (r0 & 31) * 16;
        const int8_t *p1 = tilesT + (long)(r1 >> 5) * OSFM_TILE_BYTES + (r1 & 31) * 16;
        v4i av[8], bv0[8], bv1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          av[q] = *(const v4i *)(pa + q * 512);
          bv0[q] = *(const v4i *)(p0 + q * 512);
          bv1[q] = *(const v4i *)(p1 + q * 512);
        }
        const int n0 = normT[r0], n1 = normT[r1];
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 = __builtin_amdgcn_sdot4(av[q][e], bv0[q][e], s0, false);
            s1 = __builtin_amdgcn_sdot4(av[q][e], bv1[q][e], s1, false);
          }
        if (ok0) k0 = (2 * s0 - n0) * 32 + (31 - sl);
        if (ok1) k1 = (2 * s1 - n1) * 32 + (15 - sl);
      }
      const int kmax = row16_max(max(k0, k1));
      const int ksec = row16_max(k0 == kmax ? k1 : (k1 == kmax ? k0 : max(k0, k1)));
      if (src >= 0) {
        bool full = tie;
        int win = kNone;
        if (!tie) {
          const int m1 = kmax >> 5, ord = 31 - (kmax & 31);
          const int jwin = (qrb * (kWaves * kRT) + qw * kRT + (ord >> 4)) * 32 + (ord & 3) + 8 * ((ord & 15) >> 2) + 4 * qh;
          const int m2 = ksec >> 5;  // INT_MIN >> 5 when the class has a single target: far below any bound
          const int sa = max(m2, 2 * qs), sb = max(m2, 2 * qs + 1);
          if (FQ) {
            // m1 is the exact quantised best; the quantised second lies in [sa, sb]
            const double r = fq_ratio(ratio);
            if (fq_surely_passes(qna - m1, qna - sb, eps, r))
              win = jwin;
            else if (!fq_surely_fails(qna - m1, qna - sa, eps, r))
              full = true;  // the bounds do not decide: float evaluation
          } else {
            const bool ra = ratio_ok(qna - m1, qna - sa, ratio), rbb = ratio_ok(qna - m1, qna - sb, ratio);
            if (ra == rbb)
              win = ra ? jwin : kNone;
            else
              full = true;  // the decision hangs on the parity bit of a norm in another class
          }
        }
        if (sl == 0 && !full) out[qf] = (unsigned short)win;
        redo |= (full && sl == 0) ? (1ull << src) : 0ull;
      }
    }
    OSFM_TICK(tk4)
    OSFM_PHASE((GATHER ? 5 : 0) + 0, tk0, tk1)  // sweep
    OSFM_PHASE((GATHER ? 5 : 0) + 1, tk1, tk2)  // wait for the other waves
    OSFM_PHASE((GATHER ? 5 : 0) + 2, tk2, tk3)  // merge + decide
    OSFM_PHASE((GATHER ? 5 : 0) + 3, tk3, tk4)  // class re-examination
    // the rare queries whose decision needs every target: exact best (lowest index among equals) and exact second, by the whole wave
    {
      unsigned lo = (unsigned)redo, hi = (unsigned)(redo >> 32);
#pragma unroll
      for (int m = 32; m >= 16; m >>= 1) {  // lanes 0, 16, 32, 48 hold the bits of their rows
        lo |= (unsigned)__shfl_xor((int)lo, m);
        hi |= (unsigned)__shfl_xor((int)hi, m);
      }
      redo = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
    }
    if (FQ) flag += __popcll(redo);  // diagnostic: queries that needed the float evaluation (per wave, summed by the caller)
    while (FQ && redo) {
      // float evaluation of one query by the whole wave: every target whose quantised distance does not exclude it from the float
      // top two (D^ <= upper bound of the quantised second + 2 eps) is evaluated exactly as the oracle does; cv2's K = 2 insertion
      // (lowest index first among equals) per lane in ascending index order, then across the lanes
      const int src = __builtin_ctzll(redo);
      redo &= redo - 1;
      const int qf = __shfl(myf, src), qna = __shfl(myna, src), qs2 = __shfl(s2, src);
      const double T = (sqrt((double)max(qna - 2 * qs2, 0)) + 2.0 * eps) * (1.0 + 4.0 * kFqSlack) + 1e-6;
      const double T2 = T * T;
      const int vthr = T2 >= 2.0e9 ? INT_MIN : qna - (int)T2 - 1;
      const float *qrow = descQ + (long)qf * OSFM_DESC_DIM;
      float bd0 = INFINITY, bd1 = INFINITY;
      int bi0 = INT_MAX;
      for (int i = lane; i < nT; i += 64) {
        const int v = 2 * dot_rows8(tilesQ, qf, tilesT, i) - normT[i];
        if (v >= vthr) {
          const float sq = l2sqr_rows_f32(qrow, descT + (long)i * OSFM_DESC_DIM);
          const float d = ratio < 0.0 ? sq : sqrtf(sq);
          if (d < bd1) {
            if (bd0 > d) {
              bd1 = bd0;
              bd0 = d;
              bi0 = i;
            } else {
              bd1 = d;
            }
          }
        }
      }
      const float m = wave_minf(bd0);
      const int jwin = -wave_max(bd0 == m ? -bi0 : INT_MIN);
      const float sec = wave_minf(bi0 == jwin ? bd1 : bd0);
      const bool ok = ratio < 0.0 ? (m < (float)(-ratio) * sec) : ((double)m < ratio * (double)sec);
      if (lane == 0) out[qf] = (unsigned short)(ok ? jwin : kNone);
    }
    while (!FQ && redo) {
      const int src = __builtin_ctzll(redo);
      redo &= redo - 1;
      const int qf = __shfl(myf, src), qna = __shfl(myna, src);
      int lv = INT_MIN, lj = INT_MAX, ls = INT_MIN;
      for (int i = lane; i < nT; i += 64) {
        const int v = 2 * dot_rows8(tilesQ, qf, tilesT, i) - normT[i];
        ls = max(ls, min(lv, v));
        if (v > lv) {  // ascending i within a lane: the first maximum is the lowest index
          lv = v;
          lj = i;
        }
      }
      const int m1 = wave_max(lv);
      const int jwin = -wave_max(lv == m1 ? -lj : INT_MIN);
      const int m2 = wave_max(lj == jwin ? ls : lv);
      if (lane == 0) out[qf] = (unsigned short)(ratio_ok(qna - m1, qna - m2, ratio) ? jwin : kNone);
    }
  }  // chunks
  __syncthreads();
  return flag;
}

template <bool FQ>
__global__ void __launch_bounds__(kThreads, 2) match_fused_kernel(MatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  QueryPassShared sh;
  sh.bbuf = smem;                               // [2][32 KiB]
  sh.hbuf = (int *)(smem + 2 * kChunkBytes4);  // [2][256]
  int *misc = (int *)(smem + 2 * kChunkBytes4) + 2 * kChunkCols4;  // [16]
  unsigned short *resA = (unsigned short *)(misc + 16);            // [ncap] per feature of image A
  unsigned short *resB = resA + a.ncap;                            // [ncap] per feature of image B
  unsigned short *cand = resB + a.ncap;                            // [ncap] candidate list (features of B)

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const long p = xcd_remap(blockIdx.x, a.n_pairs);
  const int img1 = a.pairs[2 * p], img2 = a.pairs[2 * p + 1];
  const int n1 = a.counts[img1], n2 = a.counts[img2];
  if (n1 < 2 || n2 < 2) {  // matching.py:363-374 / knnMatch returns < 2 neighbours
    if (tid == 0) {
      a.out_counts[p] = 0;
      a.out_flags[p] = 0;
    }
    return;
  }
  // symmetric: A = second image (rows of the big pass), B = first image (streamed, shared in L2)
  const bool rows_second = a.symmetric || a.query_second;
  const int imgA = rows_second ? img2 : img1, imgB = rows_second ? img1 : img2;
  const int nA = rows_second ? n2 : n1, nB = rows_second ? n1 : n2;
  const int8_t *tilesA = a.tiles + a.tile_off[imgA] * OSFM_TILE_BYTES;
  const int8_t *tilesB = a.tiles + a.tile_off[imgB] * OSFM_TILE_BYTES;
  const int32_t *normA = a.norms + a.tile_off[imgA] * 32;
  const int32_t *normB = a.norms + a.tile_off[imgB] * 32;
  const int32_t *hnegA = a.hneg + a.tile_off[imgA] * 32;
  const int32_t *hnegB = a.hneg + a.tile_off[imgB] * 32;
  const int8_t *tiles_pad = a.tiles + a.pad_tile * OSFM_TILE_BYTES;  // the store's first slack tile: zero descriptors, padding norms
  const int32_t *hneg_pad = a.hneg + a.pad_tile * 32;

  for (int j = tid; j < a.ncap; j += kThreads) {
    resA[j] = kNone;
    resB[j] = kNone;
  }
  if (tid == 0) {
    misc[8] = 0;
    misc[9] = 0;
  }
  __syncthreads();
  // FQ: the float rows of the two images and the quantisation error bound of the pair
  OSFM_TICK(tq0)
  const float *descA = FQ ? a.descf + a.tile_off[imgA] * (long)(32 * OSFM_DESC_DIM) : nullptr;
  const float *descB = FQ ? a.descf + a.tile_off[imgB] * (long)(32 * OSFM_DESC_DIM) : nullptr;
  const double eps = FQ ? (double)a.qerr[imgA] + (double)a.qerr[imgB] : 0.0;
  int flag = query_pass<false, FQ>(sh, tilesA, normA, nA, nA, nullptr, tilesB, normB, hnegB, nB, tiles_pad, hneg_pad, resA, a.ratio, tid,
                                   descA, descB, eps);
  OSFM_TICK(tq1)
  OSFM_PHASE(10, tq0, tq1)  // pass A
  if (a.symmetric) {
    // candidates: the features of B that some row of A chose.  resB doubles as the mark array
    // (0 = chosen) until the candidate list is built, in ascending feature order.
    for (int q = tid; q < nA; q += kThreads) {
      const int b = resA[q];
      if (b != kNone) resB[b] = 0;
    }
    __syncthreads();
    int base = 0;
    for (int j0 = 0; j0 < nB; j0 += kThreads) {
      const int j = j0 + tid;
      const bool m = j < nB && resB[j] == 0;
      const unsigned long long bal = __ballot(m);
      if (lane == 0) misc[w] = __popcll(bal);
      __syncthreads();
      int woff = 0, total = 0;
#pragma unroll
      for (int w2 = 0; w2 < kWaves; ++w2) {
        const int cnt = misc[w2];
        woff += (w2 < w) ? cnt : 0;
        total += cnt;
      }
      if (m) {
        cand[base + woff + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)j;
        resB[j] = kNone;
      }
      base += total;
      __syncthreads();
    }
    const int nK = base;
    OSFM_TICK(tq2)
    OSFM_PHASE(11, tq1, tq2)  // candidate list
    if (nK > 0)
      flag += query_pass<true, FQ>(sh, tilesB, normB, nB, nK, cand, tilesA, normA, hnegA, nA, tiles_pad, hneg_pad, resB, a.ratio, tid, descB, descA,
                                   eps);
  }
  OSFM_TICK(tq3)
  OSFM_PHASE(12, tq1, tq3)  // candidate list + pass B
  // integer store: flag = the pair has to be re-run by the exact kernel; float store: the number of queries evaluated in float
  if (FQ) {
    if (lane == 0 && flag) atomicAdd(&misc[8], flag);
  } else if (flag) {
    misc[8] = 1;
  }
  __syncthreads();
  if (tid == 0) a.out_flags[p] = misc[8];
  // ---- ordered emission: over the features of the pair's first image, or (query_second) of its second image,
  //      the order in which the reference lists the matches of match_flann (matching.py:697) ----
  {
    const bool qs = !a.symmetric && a.query_second;
    const unsigned short *res1 = a.symmetric ? resB : resA;  // indexed by the emission feature -> its partner
    const unsigned short *res2 = a.symmetric ? resA : nullptr;
    const int nE = qs ? n2 : n1;
    int base = 0;
    for (int j0 = 0; j0 < nE; j0 += kThreads) {
      const int j = j0 + tid;
      bool m = false;
      int r = kNone;
      if (j < nE) {
        r = res1[j];
        m = (r != kNone) && (!res2 || res2[r] == j);
      }
      const unsigned long long bal = __ballot(m);
      const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) misc[w] = __popcll(bal);
      __syncthreads();
      int woff = 0, total = 0;
#pragma unroll
      for (int w2 = 0; w2 < kWaves; ++w2) {
        const int cnt = misc[w2];
        woff += (w2 < w) ? cnt : 0;
        total += cnt;
      }
      if (m) {
        const int k = base + woff + prefix;
        // low half = feature of image 1, high half = feature of image 2
        if (k < a.cap) a.out_matches[p * a.cap + k] = qs ? ((uint32_t)r | ((uint32_t)j << 16)) : ((uint32_t)j | ((uint32_t)r << 16));
      }
      base += total;
      __syncthreads();
    }
    if (tid == 0) a.out_counts[p] = base;
  }
  OSFM_TICK(tq4)
  OSFM_PHASE(13, tq3, tq4)  // emission
  OSFM_PHASE(14, tq0, tq4)
#ifdef OSFM_DBG_PHASES
  if (tid == 0) atomicAdd(&g_phase[15], 1ull);
#endif
}

// ---------------------------------------------------------------------------------------------
// Exact kernel: float-key semantics of cv2 (top-2 selected on sqrtf(d^2) with lowest-index
// ties), VALU only.  Used (a) for the rare pairs whose second-nearest d^2 >= 2^22, where distinct
// integers may round to the same float distance, and (b) as an on-GPU cross-check of the fused
// kernel.  One workgroup per pair, one thread per query feature.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_row(const int8_t *tiles, int row, v4i out[8]) {
  const int8_t *t = tiles + (long)(row >> 5) * OSFM_TILE_BYTES + (row & 31) * 16;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    out[2 * ks] = *(const v4i *)(t + ks * 1024);
    out[2 * ks + 1] = *(const v4i *)(t + ks * 1024 + 512);
  }
}

__device__ void exact_direction(const int8_t *tilesQ, const int32_t *normQ, int nQ, const int8_t *tilesT,
                                const int32_t *normT, int nT, double ratio, int tid, int *res_int,
                                unsigned short *res_u16, const float *segQ, const float *segT) {
  for (int q = tid; q < nQ; q += kThreads) {
    v4i qa[8];
    load_row(tilesQ, q, qa);
    const int nq = normQ[q];
    const float sq = segQ ? segQ[q] : 0.f;
    float bd0 = INFINITY, bd1 = INFINITY;
    int bi0 = kNone;
    for (int t = 0; t < nT; ++t) {
      v4i ta[8];
      load_row(tilesT, t, ta);
      int sdot = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) sdot = __builtin_amdgcn_sdot4(qa[k][e], ta[k][e], sdot, false);
      const int d2i = nq + normT[t] - 2 * sdot;
      float d2f = (float)d2i;  // exact: < 2^24
      if (segQ) {  // the 129th dimension as cv2's normL2Sqr_ adds it: the scalar tail after the vector blocks, d += t * t (no contraction)
        const float ts = sq - segT[t];
        const float tt = ts * ts;
        d2f = d2f + tt;
      }
      const float d = ratio < 0.0 ? d2f : sqrtf(d2f);  // squared mode: FLANN reports and compares squared distances
      if (d < bd1) {  // cv2 batchDistance K=2 insertion
        if (bd0 > d) {
          bd1 = bd0;
          bd0 = d;
          bi0 = t;
        } else {
          bd1 = d;
        }
      }
    }
    const bool ok = ratio < 0.0 ? (bd0 < (float)(-ratio) * bd1) : ((double)bd0 < ratio * (double)bd1);
    const int v = ok ? bi0 : kNone;
    if (res_int) res_int[q] = v;
    if (res_u16) res_u16[q] = (unsigned short)v;
  }
}

__global__ void __launch_bounds__(kThreads) match_exact_kernel(MatchArgs a, int only_flagged) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int *colBI = (int *)smem;
  unsigned short *rowres = (unsigned short *)(colBI + a.ncap);
  int *misc = (int *)(rowres + a.ncap);
  const int tid = threadIdx.x;
  const long p = blockIdx.x;
  if (only_flagged && a.out_flags[p] == 0) return;
  const bool qs = !a.symmetric && a.query_second;  // queries = the pair's second image (match_flann)
  const int imgC = a.pairs[2 * p + (qs ? 1 : 0)], imgR = a.pairs[2 * p + (qs ? 0 : 1)];
  const int nC = a.counts[imgC], nR = a.counts[imgR];
  if (nC < 2 || nR < 2) {
    if (tid == 0) a.out_counts[p] = 0;
    return;
  }
  const int8_t *tilesC = a.tiles + a.tile_off[imgC] * OSFM_TILE_BYTES;
  const int8_t *tilesR = a.tiles + a.tile_off[imgR] * OSFM_TILE_BYTES;
  const int32_t *normC = a.norms + a.tile_off[imgC] * 32;
  const int32_t *normR = a.norms + a.tile_off[imgR] * 32;
  const float *segC = a.seg ? a.seg + a.tile_off[imgC] * 32 : nullptr, *segR = a.seg ? a.seg + a.tile_off[imgR] * 32 : nullptr;
  exact_direction(tilesC, normC, nC, tilesR, normR, nR, a.ratio, tid, colBI, nullptr, segC, segR);
  if (a.symmetric) exact_direction(tilesR, normR, nR, tilesC, normC, nC, a.ratio, tid, nullptr, rowres, segR, segC);
  __syncthreads();
  emit_matches(a, p, nC, colBI, rowres, misc, tid, qs);
}

// ---------------------------------------------------------------------------------------------
// Float kernel: descriptors that are not integers in [0, 255] (root-SIFT ...; opensfm/features.py:292-298 with feature_root).
// cv2's float semantics as the oracle restates them (oracle/match_oracle.c): squared distance accumulated in four
// 8-lane float vectors over blocks of 32 dimensions (normL2Sqr_ on an AVX2 build), reduced (d0 + d1) + (d2 + d3) and then
// ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)); sqrtf; K = 2 insertion with the lowest index first among equals; Lowe's
// test in doubles (squared mode: in float on the squared distances).  Every float operation is the oracle's, in its order
// (-ffp-contract=off), so the result is bit-identical -- on the VALU: one thread per query with its descriptor in registers,
// the targets staged through LDS 32 at a time and read as broadcasts.  ~13 k pairs/s at 2000 x 2000: the functional path for
// float descriptors; a half-precision MFMA candidate pass with an exact re-examination is the planned fast one.
// ---------------------------------------------------------------------------------------------
constexpr int kFloatTileFloats = 32 * OSFM_DESC_DIM;

__device__ void float_direction(const float *descQ, int nQ, const float *descT, int nT, double ratio, int tid, float *stage, int *res_int,
                                unsigned short *res_u16) {
  const int tT = (nT + 31) >> 5;
  for (int q0 = 0; q0 < nQ; q0 += kThreads) {
    const int q = q0 + tid;
    const bool live = q < nQ;
    float qa[OSFM_DESC_DIM];
#pragma unroll
    for (int k4 = 0; k4 < OSFM_DESC_DIM; k4 += 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) v = *(const float4 *)(descQ + (size_t)q * OSFM_DESC_DIM + k4);
      qa[k4] = v.x;
      qa[k4 + 1] = v.y;
      qa[k4 + 2] = v.z;
      qa[k4 + 3] = v.w;
    }
    float bd0 = INFINITY, bd1 = INFINITY;
    int bi0 = kNone;
    for (int tt = 0; tt < tT; ++tt) {
      __syncthreads();
      for (int k = tid * 4; k < kFloatTileFloats; k += kThreads * 4) *(float4 *)(stage + k) = *(const float4 *)(descT + (size_t)tt * kFloatTileFloats + k);
      __syncthreads();
      const int nt = min(32, nT - tt * 32);
      for (int t = 0; t < nt; ++t) {
        const float *b = stage + t * OSFM_DESC_DIM;
        float acc[4][8];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int l = 0; l < 8; ++l) acc[v][l] = 0.f;
#pragma unroll
        for (int jb = 0; jb < OSFM_DESC_DIM; jb += 32)
#pragma unroll
          for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int l4 = 0; l4 < 8; l4 += 4) {
              const float4 c = *(const float4 *)(b + jb + 8 * v + l4);
              const float cc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float d = qa[jb + 8 * v + l4 + e] - cc[e];
                const float sq = d * d;
                acc[v][l4 + e] = acc[v][l4 + e] + sq;
              }
            }
        float sv[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) sv[l] = (acc[0][l] + acc[1][l]) + (acc[2][l] + acc[3][l]);
        const float sqd = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
        const float d = ratio < 0.0 ? sqd : sqrtf(sqd);
        if (d < bd1) {  // cv2 batchDistance K = 2 insertion
          if (bd0 > d) {
            bd1 = bd0;
            bd0 = d;
            bi0 = tt * 32 + t;
          } else {
            bd1 = d;
          }
        }
      }
    }
    if (live) {
      const bool ok = ratio < 0.0 ? (bd0 < (float)(-ratio) * bd1) : ((double)bd0 < ratio * (double)bd1);
      const int v = ok ? bi0 : kNone;
      if (res_int) res_int[q] = v;
      if (res_u16) res_u16[q] = (unsigned short)v;
    }
  }
}

__global__ void __launch_bounds__(kThreads) match_float_kernel(MatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *stage = (float *)smem;  // 32 target descriptors
  int *colBI = (int *)(stage + kFloatTileFloats);
  unsigned short *rowres = (unsigned short *)(colBI + a.ncap);
  int *misc = (int *)(rowres + a.ncap);
  const int tid = threadIdx.x;
  const long p = blockIdx.x;
  if (tid == 0 && a.out_flags) a.out_flags[p] = 0;
  const bool qs = !a.symmetric && a.query_second;  // queries = the pair's second image (match_flann)
  const int imgC = a.pairs[2 * p + (qs ? 1 : 0)], imgR = a.pairs[2 * p + (qs ? 0 : 1)];
  const int nC = a.counts[imgC], nR = a.counts[imgR];
  if (nC < 2 || nR < 2) {
    if (tid == 0) a.out_counts[p] = 0;
    return;
  }
  const float *descC = a.descf + a.tile_off[imgC] * kFloatTileFloats;
  const float *descR = a.descf + a.tile_off[imgR] * kFloatTileFloats;
  float_direction(descC, nC, descR, nR, a.ratio, tid, stage, colBI, nullptr);
  if (a.symmetric) float_direction(descR, nR, descC, nC, a.ratio, tid, stage, nullptr, rowres);
  __syncthreads();
  emit_matches(a, p, nC, colBI, rowres, misc, tid, qs);
}

// ---------------------------------------------------------------------------------------------
// Hamming kernel: binary descriptors (uint8 bit strings; matching.py:737-740 switches cv2 to BruteForce-Hamming).  cv2's batchDistance
// with NORM_HAMMING gives int distances, the same K = 2 insertion (strict <, lowest index first among equals) and DMatch.distance =
// float(int); Lowe's test in doubles (matching.py:752).  Integer work, no matrix cores: one thread per query with its 512 bits in 16
// registers, the targets staged through LDS 256 rows at a time and read as broadcasts, v_bcnt_u32_b32 (popcount + add) per dword.
// ---------------------------------------------------------------------------------------------
constexpr int kHamStageRows = 256;

__device__ void hamming_direction(const uint32_t *binQ, int nQ, const uint32_t *binT, int nT, double ratio, int tid, uint32_t *stage, int *res_int,
                                  unsigned short *res_u16) {
  for (int q0 = 0; q0 < nQ; q0 += kThreads) {
    const int q = q0 + tid;
    const bool live = q < nQ;
    uint4 qa[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) qa[k] = live ? *(const uint4 *)(binQ + (size_t)q * 16 + 4 * k) : make_uint4(0u, 0u, 0u, 0u);
    int bd0 = 0x7fffffff, bd1 = 0x7fffffff, bi0 = kNone;
    for (int t0 = 0; t0 < nT; t0 += kHamStageRows) {
      const int nt = min(kHamStageRows, nT - t0);
      __syncthreads();
      for (int k = tid; k < nt * 4; k += kThreads) *(uint4 *)(stage + 4 * k) = *(const uint4 *)(binT + (size_t)t0 * 16 + 4 * k);
      __syncthreads();
      for (int t = 0; t < nt; ++t) {
        const uint4 *b = (const uint4 *)(stage + t * 16);
        int d = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint4 c = b[k];
          d += __popc(qa[k].x ^ c.x) + __popc(qa[k].y ^ c.y) + __popc(qa[k].z ^ c.z) + __popc(qa[k].w ^ c.w);
        }
        if (d < bd1) {  // cv2 batchDistance K = 2 insertion
          if (bd0 > d) {
            bd1 = bd0;
            bd0 = d;
            bi0 = t0 + t;
          } else {
            bd1 = d;
          }
        }
      }
    }
    if (live) {
      const bool ok = (double)(float)bd0 < ratio * (double)(float)bd1;  // DMatch.distance is float32 of the int; nT >= 2: both are set
      const int v = ok ? bi0 : kNone;
      if (res_int) res_int[q] = v;
      if (res_u16) res_u16[q] = (unsigned short)v;
    }
  }
}

__global__ void __launch_bounds__(kThreads) match_hamming_kernel(MatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *stage = (uint32_t *)smem;  // kHamStageRows target rows
  int *colBI = (int *)(stage + kHamStageRows * 16);
  unsigned short *rowres = (unsigned short *)(colBI + a.ncap);
  int *misc = (int *)(rowres + a.ncap);
  const int tid = threadIdx.x;
  const long p = blockIdx.x;
  if (tid == 0 && a.out_flags) a.out_flags[p] = 0;
  const bool qs = !a.symmetric && a.query_second;  // queries = the pair's second image (match_flann on bit strings, round 6)
  const int imgC = a.pairs[2 * p + (qs ? 1 : 0)], imgR = a.pairs[2 * p + (qs ? 0 : 1)];
  const int nC = a.counts[imgC], nR = a.counts[imgR];
  if (nC < 2 || nR < 2) {
    if (tid == 0) a.out_counts[p] = 0;
    return;
  }
  const uint32_t *binC = a.bin + a.tile_off[imgC] * (32 * 16);
  const uint32_t *binR = a.bin + a.tile_off[imgR] * (32 * 16);
  hamming_direction(binC, nC, binR, nR, a.ratio, tid, stage, colBI, nullptr);
  if (a.symmetric) hamming_direction(binR, nR, binC, nC, a.ratio, tid, stage, nullptr, rowres);
  __syncthreads();
  emit_matches(a, p, nC, colBI, rowres, misc, tid, qs);
}

}  // namespace

size_t osfm_match_lds_bytes(int ncap) { return (size_t)2 * kChunkBytes4 + 2 * kChunkCols4 * 4 + 64 + (size_t)ncap * 6; }

static int ensure_kernel_attributes(int device) {
  static OsfmPerDeviceOnce once;
  return once.run(device, []() -> int {
    OSFM_HIP(hipFuncSetAttribute((const void *)match_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OSFM_HIP(hipFuncSetAttribute((const void *)match_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OSFM_HIP(hipFuncSetAttribute((const void *)match_exact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OSFM_HIP(hipFuncSetAttribute((const void *)match_float_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OSFM_HIP(hipFuncSetAttribute((const void *)match_hamming_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return OSFM_OK;
  });
}

int osfm_launch_match(osfm_ctx *ctx, const osfm_store *store, const int32_t *d_pairs, int64_t n_pairs,
                      double ratio, int symmetric, int squared_ratio, int cap, int32_t *d_counts, uint32_t *d_matches,
                      int32_t *d_flags, bool exact_kernel, hipStream_t stream) {
  if (n_pairs == 0) return OSFM_OK;
  MatchArgs a;
  a.tiles = store->d_tiles;
  a.norms = store->d_norms;
  a.hneg = store->d_hneg;
  a.descf = store->d_descf;
  a.qerr = store->d_qerr;
  a.seg = store->d_seg;
  a.bin = store->d_bin;
  a.tile_off = store->d_tile_off;
  a.counts = store->d_counts;
  a.pairs = d_pairs;
  a.n_pairs = n_pairs;
  // squared mode (FLANN semantics, matching.py:683-720): the kernels see -float32(ratio^2); one-way matching then queries
  // with the pair's second image, as match_flann(index1, f2) does
  a.ratio = squared_ratio ? -(double)(float)(ratio * ratio) : ratio;
  a.symmetric = symmetric;
  a.query_second = squared_ratio && !symmetric;
  a.cap = cap;
  a.ncap = ((store->max_count + 127) / 128) * 128;
  if (a.ncap < 128) a.ncap = 128;
  a.out_counts = d_counts;
  a.out_matches = d_matches;
  a.out_flags = d_flags;
  a.pad_tile = store->tile_off[store->n_images];
  OSFM_REQUIRE(a.ncap <= OSFM_MAX_FEATURES, OSFM_E_UNSUPPORTED, "more than %d features in an image", OSFM_MAX_FEATURES);
  OSFM_REQUIRE(n_pairs < (1ll << 31), OSFM_E_INVALID, "too many pairs in one launch");
  {
    const int rc = ensure_kernel_attributes(ctx->device);
    if (rc != OSFM_OK) return rc;
  }
  if (store->is_binary) {
    // bit strings: Hamming distance on the VALU, every pair in one launch; nothing is ever flagged for a second run.
    // matcher_type FLANN on bit strings (round 6; cv2's LSH index searched EXACTLY, as the kd-forest is for floats): knnSearch returns int32
    // Hamming distances, so `dists[:, 0] < lowes_ratio ** 2 * dists[:, 1]` (matching.py:695-696) is a test in DOUBLES with the squared ratio --
    // the kernel's own test with that ratio (positive: nothing here takes a square root); one-way matching queries with the second image
    if (squared_ratio) a.ratio = ratio * ratio;
    if (exact_kernel && d_flags != nullptr) return OSFM_OK;
    const size_t lds = (size_t)kHamStageRows * 64 + (size_t)a.ncap * 6 + 64;
    hipLaunchKernelGGL(match_hamming_kernel, dim3((unsigned)n_pairs), dim3(kThreads), lds, stream, a);
  } else if (store->d_seg) {
    // segmentation column: every pair on the exact kernel with the label term (the int8 matrix path has no 129th dimension)
    if (exact_kernel && d_flags != nullptr) return OSFM_OK;  // "re-run the flagged pairs": none, the first launch was exact
    if (d_flags) OSFM_HIP(hipMemsetAsync(d_flags, 0, (size_t)n_pairs * sizeof(int32_t), stream));
    const size_t lds = (size_t)a.ncap * 6 + 64;
    hipLaunchKernelGGL(match_exact_kernel, dim3((unsigned)n_pairs), dim3(kThreads), lds, stream, a, 0);
  } else if (store->is_float) {
    // float store: the fused kernel on the 8-bit quantisation with rigorous bounds and float evaluation of what they leave open
    // (FQ mode); the exact float kernel for every pair when asked for (cross-check) or when the store could not be quantised.
    // Nothing is ever flagged for a second run: out_flags then counts the queries that went through the float evaluation.
    if (exact_kernel && d_flags != nullptr) return OSFM_OK;  // "re-run the flagged pairs": none
    if (!exact_kernel && store->quantised) {
      hipLaunchKernelGGL(match_fused_kernel<true>, dim3((unsigned)n_pairs), dim3(kThreads), osfm_match_lds_bytes(a.ncap), stream, a);
    } else {
      if (d_flags) OSFM_HIP(hipMemsetAsync(d_flags, 0, (size_t)n_pairs * sizeof(int32_t), stream));
      const size_t lds = (size_t)kFloatTileFloats * sizeof(float) + (size_t)a.ncap * 6 + 64;
      hipLaunchKernelGGL(match_float_kernel, dim3((unsigned)n_pairs), dim3(kThreads), lds, stream, a);
    }
  } else if (!exact_kernel) {
    hipLaunchKernelGGL(match_fused_kernel<false>, dim3((unsigned)n_pairs), dim3(kThreads), osfm_match_lds_bytes(a.ncap), stream, a);
  } else {
    const size_t lds = (size_t)a.ncap * 6 + 64;
    hipLaunchKernelGGL(match_exact_kernel, dim3((unsigned)n_pairs), dim3(kThreads), lds, stream, a, d_flags != nullptr ? 1 : 0);
  }
  OSFM_HIP(hipGetLastError());
  return OSFM_OK;
}
