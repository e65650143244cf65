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
// Epilogue (the kernel is bounded by integer VALU + skeleton work next to the matrix pipe, K is only
// 128): "best-only + lazy exact second".  The second-nearest neighbour is only needed for the ratio
// test, and only its VALUE.  Per class of candidates (the 32 column classes j mod 32 held by the 32
// lanes of a half-wave) only the BEST packed key is kept:  best = v_max3_i32(best, key_a, key_b),
// key = (2S - norm_other) * 2^k + tile tag (one v_lshl_add_u32).  Classes are merged into (global best,
// second-largest class best =: s_c).  The true second s satisfies s >= s_c and the ratio test is
// monotone in d2:  fails with s_c => fails (final, the vast majority);  passes with s_c => the winner's
// own class is re-examined exactly by the whole wavefront with v_dot4 dot products.
// (Earlier generations -- exact top-2 in both directions, both directions per pass -- measured 0.16 and
// 0.24 of the int8 peak, profiles/r01_match_v1_rocprof.txt / r01_match_v2_pmc.txt; they were removed in
// round 2, the exact VALU kernel below remains as the on-GPU cross-check.)
#include <cstdlib>

#include "osfm_internal.h"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

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
  const int32_t *pad_norm;  // one device int holding OSFM_PAD_NORM (source for out-of-range norm DMA)
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
      if (k < a.cap) a.out_matches[p * a.cap + k] = (uint32_t)j | ((uint32_t)r << 16);
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

// ---------------------------------------------------------------------------------------------
// Fused MFMA kernel, version 4: "one direction at a time".
//
// Measured on MI355X (tools/prof_match.py + compile-time variants of v2): the kernel time is
//   skeleton (chunk DMA, barriers, LDS operand reads, column atomics, merges)  +  VALU epilogue  +
//   MFMA time, and the three do NOT overlap: removing the MFMAs or removing the epilogue saves the
//   same ~5 ms out of 15.6 (19 900 pairs of 2000x2000), removing both leaves 5.1 ms.
// The MFMA part is at its floor, so v4 removes VALU and skeleton work instead: the distance matrix
// is only reduced along ONE direction per pass (1.5 VALU ops per element instead of 3, no column
// partials, no LDS atomics):
//   pass A: rows = image A in registers, image B streamed; resA[a] = best b if it passes the ratio
//           test (class bests + lazy exact second, exactly as v2's row direction);
//   pass B: (symmetric matching only) the mutual check needs "best a for b" only for the b that some
//           row chose: those candidates (typically a small subset of image B) become the rows of a
//           second, much smaller pass against the streamed image A;
//   a pair (a, b) is emitted iff resA[a] == b and resB[b] == a.
// For symmetric matching A is the pair's SECOND image, so that the image streamed by the big pass
// is the first one, which the ~64 pairs in flight on an XCD share in L2 (xcd_remap).
// Results are bit-identical to v1/v2/the exact kernel/the oracle.
// ---------------------------------------------------------------------------------------------
constexpr int kCT4 = 8;                              // v4 stages 256 columns per chunk: half the barriers / DMA issues of v2
constexpr int kChunkCols4 = kCT4 * 32;
constexpr int kChunkBytes4 = kCT4 * OSFM_TILE_BYTES;  // 32 KiB, double buffered
struct RowPassShared {
  unsigned char *bbuf;  // [2][16 KiB] chunk double buffer
  int *nbuf;            // [2][128] norms of the staged chunk
  const int32_t *pad_norm;
};

// rows: slot q in [0, nslots) is feature rowsel[q] of image X (rowsel == nullptr: identity).
// out[feature of X] = best feature of Y (ratio test passed) or kNone.  Returns the collision flag.
template <bool GATHER>
__device__ __forceinline__ int row_pass(const RowPassShared &sh, const int8_t *tilesX, const int32_t *normX, int nX, int nslots,
                                        const unsigned short *rowsel, const int8_t *tilesY, const int32_t *normY, int nY,
                                        unsigned short *out, double ratio, int tid) {
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tY = (nY + 31) >> 5;
  const int tS = (nslots + 31) >> 5;  // row tiles (slots)
  const int nchunks = (tY + kCT4 - 1) / kCT4;
  const int nrb = (tS + kWaves * kRT - 1) / (kWaves * kRT);
  const int nsteps = nrb * nchunks;
  unsigned char *bbuf = sh.bbuf;
  int *nbuf = sh.nbuf;
  int flag = 0;
  // images above 4096 features (128 tiles) do not fit a tile index into the 32-bit keys: value-only keys, and the
  // re-examination of the rows that pass also has to find the winner (cold path)
  const bool big = tY > 128;
  const int ksh = big ? 1 : 8;

  // first chunk of Y: global -> registers -> LDS
  {
    uint4 pre[kCT4];
#pragma unroll
    for (int q = 0; q < kCT4; ++q) {
      pre[q] = make_uint4(0, 0, 0, 0);
      if (q < tY) pre[q] = *(const uint4 *)(tilesY + (long)q * OSFM_TILE_BYTES + tid * 16);
    }
#pragma unroll
    for (int q = 0; q < kCT4; ++q) *(uint4 *)(bbuf + q * OSFM_TILE_BYTES + tid * 16) = pre[q];
    for (int j = tid; j < kChunkCols4; j += kThreads) nbuf[j] = (j < tY * 32) ? normY[j] : OSFM_PAD_NORM;
  }
  __syncthreads();

  v4i afrag[kRT][4], anext[kRT][4];
  int nrm = OSFM_PAD_NORM, nrm_next = OSFM_PAD_NORM;
  int xrow = 0, xrow_next = 0;  // feature index of the row slot this lane describes (lane = slot - rt0*32)
  int rbst[kRT][16];

  // A operands (+ norm, + feature index) of the 64 row slots of this wave in row block rb
  auto load_rows = [&](int rb, v4i (&af)[kRT][4], int &nr, int &xr) {
    const int slot0 = (rb * (kWaves * kRT) + w * kRT) * 32;
    {
      const int q = slot0 + lane;
      xr = (q < nslots) ? (GATHER ? (int)rowsel[q] : q) : -1;
      nr = (xr >= 0 && xr < nX) ? normX[xr] : OSFM_PAD_NORM;
    }
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt) {
      const int q = slot0 + rt * 32 + (lane & 31);
      int f = (q < nslots) ? (GATHER ? (int)rowsel[q] : q) : -1;
      if (!GATHER && q >= tS * 32) f = -1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        v4i z = {0, 0, 0, 0};
        af[rt][ks] = z;
        // operand layout: lane = half*32 + row-in-tile holds bytes [ks*32 + half*16, +16) of that row
        if (f >= 0) af[rt][ks] = *(const v4i *)(tilesX + (long)(f >> 5) * OSFM_TILE_BYTES + ks * 1024 + (lane >> 5) * 512 + (f & 31) * 16);
      }
    }
  };

  for (int rb = 0; rb < nrb; ++rb) {
    const int rt0 = rb * (kWaves * kRT) + w * kRT;
    const int nrt = min(kRT, max(0, tS - rt0));
    if (rb == 0) {
      load_rows(0, afrag, nrm, xrow);
    } else {
      nrm = nrm_next;
      xrow = xrow_next;
#pragma unroll
      for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) afrag[rt][ks] = anext[rt][ks];
    }
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) rbst[rt][r] = INT_MIN;
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(afrag[rt][ks]));

    for (int c = 0; c < nchunks; ++c) {
      const int s = rb * nchunks + c;
      const bool has_next = (s + 1 < nsteps);
      const int cn = (c + 1 == nchunks) ? 0 : c + 1;
      if (has_next) {
        unsigned char *nb2 = bbuf + ((s + 1) & 1) * kChunkBytes4;
#pragma unroll
        for (int q = 0; q < kCT4; ++q) {
          const int gt = cn * kCT4 + q;
          const int8_t *src = tilesY + (long)(gt < tY ? gt : 0) * OSFM_TILE_BYTES + tid * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                           (__attribute__((address_space(3))) void *)(nb2 + q * OSFM_TILE_BYTES + w * 1024), 16, 0, 0);
        }
        if (w * 64 < kChunkCols4) {
          const int jn = cn * kChunkCols4 + tid;
          const int32_t *srcn = (jn < tY * 32) ? normY + jn : sh.pad_norm;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)srcn,
                                           (__attribute__((address_space(3))) void *)(nbuf + ((s + 1) & 1) * kChunkCols4 + w * 64), 4, 0, 0);
        }
      }
      // ---- compute: a hand-made software pipeline over the step's 8 work items ----
      // item = (pair of column tiles, row tile): 8 MFMAs (two accumulation chains) + a 48-op integer
      // epilogue (2 v_lshl_add + 1 v_max3 per row register).  Inside one wavefront the chain
      // LDS read -> MFMA -> epilogue is strictly dependent, and with two waves per SIMD the hardware
      // cannot hide it (tools/ubench_overlap.hip: +24 % from pipelining exactly this shape).  So:
      // the B operands of the NEXT pair are read from LDS one pair ahead, and every MFMA of item i+1 is
      // followed by one eighth of the epilogue of item i, pinned with sched_barrier; operands and
      // accumulators ping-pong between two register sets (no copies).  Branch-free: row tiles beyond the
      // image have zero operands and padding norms, column tiles beyond it carry the padding norm, so
      // neither can win.
      const unsigned char *bb = bbuf + (s & 1) * kChunkBytes4;
      if (nrt > 0) {
        const int *nbs = nbuf + (s & 1) * kChunkCols4;
        v4i bf[2][2][4];
        int ck[2][2];
        v16i acc[2][2];
#define OSFM_LOAD_PAIR(BUF, PR)                                                                              \
  {                                                                                                          \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                          \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                       \
        bf[BUF][h][ks] = *(const v4i *)(bb + (2 * (PR) + h) * OSFM_TILE_BYTES + ks * 1024 + lane * 16);     \
      const int nb = nbs[(2 * (PR) + h) * 32 + (lane & 31)];                                                 \
      /* <= 128 tiles: key = value * 128 + (127 - tile): the winner's tile rides in the key.  More tiles: the value alone */ \
      ck[BUF][h] = big ? -nb : -(nb << 7) + (127 - ((c * kCT4 + 2 * (PR) + h) & 127));                     \
    }                                                                                                        \
  }
#ifdef OSFM_DBG_NOMFMA
#define OSFM_MFMA(AB, RT, BUF, I) acc[AB][(I) & 1][(I) >> 1] += afrag[RT][(I) >> 1][0] ^ bf[BUF][(I) & 1][(I) >> 1][1];
#else
#define OSFM_MFMA(AB, RT, BUF, I) \
  acc[AB][(I) & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[RT][(I) >> 1], bf[BUF][(I) & 1][(I) >> 1], acc[AB][(I) & 1], 0, 0, 0);
#endif
#ifdef OSFM_DBG_NOEPI
#define OSFM_EPI(AB, RT, BUF, I) \
  if ((I) == 7) rbst[RT][0] = max(rbst[RT][0], acc[AB][0][0] + acc[AB][1][5] + ck[BUF][0] + ck[BUF][1]);
#else
#define OSFM_EPI(AB, RT, BUF, I)                                                                             \
  {                                                                                                          \
    _Pragma("unroll") for (int rr = 2 * (I); rr < 2 * (I) + 2; ++rr) {                                       \
      const int k0 = (acc[AB][0][rr] << ksh) + ck[BUF][0], k1 = (acc[AB][1][rr] << ksh) + ck[BUF][1];        \
      rbst[RT][rr] = max(max(rbst[RT][rr], k0), k1);                                                         \
    }                                                                                                        \
  }
#endif
        const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        OSFM_LOAD_PAIR(0, 0)
        acc[0][0] = zero16;
        acc[0][1] = zero16;
#pragma unroll
        for (int i = 0; i < 8; ++i) OSFM_MFMA(0, 0, 0, i)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int item = 1; item < kCT4; ++item) {  // kCT4 / 2 pairs x 2 row tiles = kCT4 items
          const int pr = item >> 1, rt = item & 1, buf = pr & 1, ab = item & 1;
          const int ppr = (item - 1) >> 1, prt = (item - 1) & 1, pbuf = ppr & 1, pab = (item - 1) & 1;
          if (rt == 1 && pr + 1 < kCT4 / 2) OSFM_LOAD_PAIR(buf ^ 1, pr + 1)  // one pair ahead of its first use
          acc[ab][0] = zero16;
          acc[ab][1] = zero16;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            OSFM_MFMA(ab, rt, buf, i)
            __builtin_amdgcn_sched_barrier(0);
            OSFM_EPI(pab, prt, pbuf, i)
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) OSFM_EPI((kCT4 - 1) & 1, (kCT4 - 1) & 1, ((kCT4 - 1) >> 1) & 1, i)
#undef OSFM_LOAD_PAIR
#undef OSFM_MFMA
#undef OSFM_EPI
      }
      __syncthreads();
      // ---- end of a row block: merge the 32 column classes of every row (see v2) ----
      if (c == nchunks - 1) {
        if (rb + 1 < nrb) load_rows(rb + 1, anext, nrm_next, xrow_next);
        unsigned char *trb = bbuf + (s & 1) * kChunkBytes4 + w * 1024;
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) {
          if (rt < nrt) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              *(int *)(trb + (r >> 2) * OSFM_TILE_BYTES + ((r & 3) + 4 * (lane >> 5)) * 128 + (lane & 31) * 4) = rbst[rt][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int row32 = lane >> 1, hc = lane & 1;
            int bkey = INT_MIN, bcls = 0, skey = INT_MIN;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int4 v4 = *(const int4 *)(trb + (row32 >> 3) * OSFM_TILE_BYTES + (row32 & 7) * 128 + hc * 64 + q * 16);
              const int vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                skey = max(skey, min(bkey, vv[e]));
                const bool up = vv[e] > bkey;  // strict: the lowest class keeps ties (lowest column)
                bcls = up ? hc * 16 + q * 4 + e : bcls;
                bkey = up ? vv[e] : bkey;
              }
            }
            {
              const int pk = __shfl_xor(bkey, 1), pc = __shfl_xor(bcls, 1), ps = __shfl_xor(skey, 1);
              const int nsk = max(min(bkey, pk), max(skey, ps));
              const bool take = (pk > bkey) || (pk == bkey && pc < bcls);
              bcls = take ? pc : bcls;
              bkey = take ? pk : bkey;
              skey = nsk;
            }
            const int il = rt * 32 + row32;
            const int na = __shfl(nrm, il);
            const int xr = __shfl(xrow, il);
            const int bv = big ? bkey : bkey >> 7, sv = big ? skey : skey >> 7;
            const int bj = (127 - (bkey & 127)) * 32 + bcls;  // winner's column when the key carries its tile (!big)
            // value-only keys: every class whose best equals the overall best may hold the lowest-index winner
            unsigned tmask = 0;
            if (big) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int4 v4 = *(const int4 *)(trb + (row32 >> 3) * OSFM_TILE_BYTES + (row32 & 7) * 128 + hc * 64 + q * 16);
                const int vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) tmask |= (vv[e] == bkey) ? (1u << (hc * 16 + q * 4 + e)) : 0u;
              }
              tmask |= __shfl_xor(tmask, 1);
            }
            bool want = false;
            if (hc == 0 && xr >= 0 && xr < nX) {
              const int d1 = na - bv, d2 = na - sv;
              if (d2 >= kCollisionD2 && ratio >= 0.0) flag = 1;  // squared mode never takes a square root
              want = ratio_ok(d1, d2, ratio);  // passes against the class bound: re-examine
              if (!want) out[xr] = kNone;
            }
            unsigned long long pending = __ballot(want);
            while (pending) {
              const int src = __builtin_ctzll(pending);
              pending &= pending - 1;
              const int qsv = __shfl(sv, src), qna = __shfl(na, src), qxr = __shfl(xr, src);
              if (!big) {
                // exact second inside the winner's class = columns {t*32 + (bj&31)}
                const int qbj = __shfl(bj, src), qbv = __shfl(bv, src);
                int mx = INT_MIN;
                for (int t0 = 0; t0 < tY; t0 += 64) {
                  const int t = t0 + lane;
                  const int j = t * 32 + (qbj & 31);
                  if (t < tY && j != qbj) mx = max(mx, 2 * dot_rows8(tilesX, qxr, tilesY, j) - normY[j]);
                }
                mx = wave_max(mx);
                if (lane == 0) {
                  const int s2 = max(qsv, mx);
                  out[qxr] = ratio_ok(qna - qbv, qna - s2, ratio) ? qbj : kNone;
                }
              } else {
                // the classes that reach the best value are re-examined exactly: lowest index among equals
                // (cv2's rule) and exact second
                unsigned qmask = (unsigned)__shfl((int)tmask, src);
                int lv = INT_MIN, lj = INT_MAX, ls = INT_MIN;
                while (qmask) {
                  const int qcls = __builtin_ctz(qmask);
                  qmask &= qmask - 1;
                  for (int t0 = 0; t0 < tY; t0 += 64) {
                    const int t = t0 + lane;
                    if (t < tY) {
                      const int j = t * 32 + qcls;
                      const int v = 2 * dot_rows8(tilesX, qxr, tilesY, j) - normY[j];
                      ls = max(ls, min(lv, v));
                      if (v > lv || (v == lv && j < lj)) {
                        lv = v;
                        lj = j;
                      }
                    }
                  }
                }
                const int m1 = wave_max(lv);
                const int jwin = -wave_max(lv == m1 ? -lj : INT_MIN);
                const int m2 = wave_max(lj == jwin ? ls : lv);
                if (lane == 0) {
                  const int s2 = max(qsv, m2);
                  out[qxr] = ratio_ok(qna - m1, qna - s2, ratio) ? jwin : kNone;
                }
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
        }
      }
    }  // chunks
  }    // row blocks
  __syncthreads();
  return flag;
}

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
constexpr int kPadHneg = -(1 << 23);  // accumulator seed of a padding target: can never be the maximum

struct QueryPassShared {
  unsigned char *bbuf;  // [2][32 KiB] query chunks (tile layout); the drained one doubles as the merge scratch
};

// queries: slot q in [0, nslots) is feature qsel[q] of image Q (qsel == nullptr: identity); targets: all nT features of image T.
// out[query feature] = its nearest target if the ratio test passes, else kNone.  Returns the collision flag.
template <bool GATHER>
__device__ __forceinline__ int query_pass(const QueryPassShared &sh, const int8_t *tilesQ, const int32_t *normQ, int nQ, int nslots,
                                          const unsigned short *qsel, const int8_t *tilesT, const int32_t *normT, const int32_t *hnegT,
                                          int nT, unsigned short *out, double ratio, int tid) {
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

  // the wave's two target row tiles of row block rb: A operands and accumulator seeds
  auto load_targets = [&](int rb, v4i (&af)[kRT][4], v16i (&hn)[kRT]) {
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt) {
      const int t = rb * (kWaves * kRT) + w * kRT + rt;
      const bool live = t < tT;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        v4i z = {0, 0, 0, 0};
        af[rt][ks] = z;
        if (live) af[rt][ks] = *(const v4i *)(tilesT + (long)t * OSFM_TILE_BYTES + ks * 1024 + lane * 16);
      }
      // accumulator register r of half h belongs to row (r & 3) + 8 (r >> 2) + 4 h of the tile
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v4i hv = {kPadHneg, kPadHneg, kPadHneg, kPadHneg};
        if (live) hv = *(const v4i *)(hnegT + (long)t * 32 + 8 * g + 4 * (lane >> 5));
        hn[rt][4 * g + 0] = hv[0];
        hn[rt][4 * g + 1] = hv[1];
        hn[rt][4 * g + 2] = hv[2];
        hn[rt][4 * g + 3] = hv[3];
      }
    }
  };

  dma_chunk(0);
  v4i afrag[kRT][4], anext[kRT][4];
  v16i hinit[kRT], hnext[kRT];
  load_targets(0, afrag, hinit);

  for (int c = 0; c < nchunks; ++c) {
    // the query this thread decides at the end of the chunk (slot c*256 + tid): its norm, fetched now
    const int myslot = c * kChunkCols4 + tid;
    const int myf = myslot < nslots ? (GATHER ? (int)qsel[myslot] : myslot) : -1;
    const int myna = (myf >= 0 && myf < nQ) ? normQ[myf] : OSFM_PAD_NORM;
    if (c + 1 < nchunks) dma_chunk(c + 1);  // its buffer was released by the merge of chunk c - 1
    // chunk c has landed once all but the newest kCT4 DMAs of this thread have (the first chunk: all of them)
    if (c + 1 < nchunks)
      __builtin_amdgcn_s_waitcnt(0x0070 | (kCT4 & 15) | (((kCT4 >> 4) & 3) << 14));  // vmcnt(kCT4), keep lgkm/exp
    else
      __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0)
    __syncthreads();
    const unsigned char *bb = sh.bbuf + (c & 1) * kChunkBytes4;

    int cb[kCT4], cs[kCT4], ci[kCT4];
#pragma unroll
    for (int t = 0; t < kCT4; ++t) {
      cb[t] = INT_MIN;
      cs[t] = INT_MIN;
      ci[t] = 0;
    }

    for (int rb = 0; rb < nrb; ++rb) {
      // next step's targets, one step ahead (the next chunk starts over at row block 0)
      {
        const int rbn = (rb + 1 < nrb) ? rb + 1 : 0;
        if (rb + 1 < nrb || c + 1 < nchunks) load_targets(rbn, anext, hnext);
      }
      const int t0 = rb * (kWaves * kRT) + w * kRT;
      if (t0 < tT) {
        v4i bf[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bf[0][ks] = *(const v4i *)(bb + ks * 1024 + lane * 16);
#pragma unroll
        for (int t = 0; t < kCT4; ++t) {
          const int cur = t & 1;
          if (t + 1 < kCT4) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bf[cur ^ 1][ks] = *(const v4i *)(bb + (t + 1) * OSFM_TILE_BYTES + ks * 1024 + lane * 16);
          }
          v16i acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[0][0], bf[cur][0], hinit[0], 0, 0, 0);
          v16i acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[1][0], bf[cur][0], hinit[1], 0, 0, 0);
#pragma unroll
          for (int ks = 1; ks < 4; ++ks) {
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[0][ks], bf[cur][ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[1][ks], bf[cur][ks], acc1, 0, 0, 0);
          }
          // max over the 32 targets of this lane: four independent v_max3 chains, then the class update
          int m0 = max(max(acc0[0], acc0[1]), acc0[2]), m1 = max(max(acc0[8], acc0[9]), acc0[10]);
          int m2 = max(max(acc1[0], acc1[1]), acc1[2]), m3 = max(max(acc1[8], acc1[9]), acc1[10]);
          m0 = max(max(m0, acc0[3]), acc0[4]);
          m1 = max(max(m1, acc0[11]), acc0[12]);
          m2 = max(max(m2, acc1[3]), acc1[4]);
          m3 = max(max(m3, acc1[11]), acc1[12]);
          m0 = max(max(m0, acc0[5]), acc0[6]);
          m1 = max(max(m1, acc0[13]), acc0[14]);
          m2 = max(max(m2, acc1[5]), acc1[6]);
          m3 = max(max(m3, acc1[13]), acc1[14]);
          m0 = max(max(m0, acc0[7]), acc0[15]);
          m2 = max(max(m2, acc1[7]), acc1[15]);
          const int m = max(max(m0, m1), max(m2, m3));
          asm("v_med3_i32 %0, %1, %2, %3" : "=v"(cs[t]) : "v"(cb[t]), "v"(m), "v"(cs[t]));  // cs <= cb: the second-largest of {cb, m, cs}
          ci[t] = m > cb[t] ? rb : ci[t];
          cb[t] = max(cb[t], m);
        }
      }
      if (rb + 1 < nrb || c + 1 < nchunks) {
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) {
          hinit[rt] = hnext[rt];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) afrag[rt][ks] = anext[rt][ks];
        }
      }
    }  // row blocks

    // ---- end of the chunk: merge the 8 partial classes (4 waves x 2 halves) of every query, decide, re-examine ----
    __syncthreads();  // everyone is done reading the chunk: its buffer becomes the scratch
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
      if (d2hi >= kCollisionD2 && ratio >= 0.0) flag = 1;  // squared mode never takes a square root
      want = ratio_ok(d1lo, d2hi, ratio);
      if (!want) out[myf] = kNone;
    }
    unsigned long long pending = __ballot(want);
    while (pending) {
      const int src = __builtin_ctzll(pending);
      pending &= pending - 1;
      const int qf = __shfl(myf, src), qna = __shfl(myna, src), qb = __shfl(b, src), qs = __shfl(s2, src), qid = __shfl(id, src);
      bool full = (qs == qb);  // two classes tie on u: the winner may sit in either
      int win = kNone;
      if (!full) {
        const int qrb = qid >> 3, qw = (qid >> 1) & 3, qh = qid & 1;
        const int rtl = (lane >> 4) & 1, r = lane & 15;
        const int row = (qrb * (kWaves * kRT) + qw * kRT + rtl) * 32 + (r & 3) + 8 * (r >> 2) + 4 * qh;
        int v = INT_MIN;
        if (lane < 32 && row < nT) v = 2 * dot_rows8(tilesQ, qf, tilesT, row) - normT[row];
        const int m1 = wave_max(v);
        const int jwin = -wave_max(v == m1 ? -row : INT_MIN);  // lowest index among equals (cv2's rule)
        const int m2 = wave_max((lane < 32 && row != jwin) ? v : INT_MIN);
        const int sa = max(m2, 2 * qs), sb = max(m2, 2 * qs + 1);
        const bool ra = ratio_ok(qna - m1, qna - sa, ratio), rbb = ratio_ok(qna - m1, qna - sb, ratio);
        if (ra == rbb)
          win = ra ? jwin : kNone;
        else
          full = true;  // the decision hangs on the parity bit of a norm in another class
      }
      if (full) {
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
        win = ratio_ok(qna - m1, qna - m2, ratio) ? jwin : kNone;
      }
      if (lane == 0) out[qf] = (unsigned short)win;
    }
  }  // chunks
  __syncthreads();
  return flag;
}

__global__ void __launch_bounds__(kThreads, 2) match_fused4_kernel(MatchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef OSFM_MATCH_V4
  RowPassShared sh;
  sh.bbuf = smem;                                                  // [2][32 KiB]
  sh.nbuf = (int *)(smem + 2 * kChunkBytes4);                       // [2][256]
  sh.pad_norm = a.pad_norm;
#else
  QueryPassShared sh;
  sh.bbuf = smem;  // [2][32 KiB]
#endif
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

  for (int j = tid; j < a.ncap; j += kThreads) {
    resA[j] = kNone;
    resB[j] = kNone;
  }
  if (tid == 0) {
    misc[8] = 0;
    misc[9] = 0;
  }
  __syncthreads();
#ifdef OSFM_MATCH_V4
  int flag = row_pass<false>(sh, tilesA, normA, nA, nA, nullptr, tilesB, normB, nB, resA, a.ratio, tid);
  (void)hnegA;
  (void)hnegB;
#else
  int flag = query_pass<false>(sh, tilesA, normA, nA, nA, nullptr, tilesB, normB, hnegB, nB, resA, a.ratio, tid);
#endif
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
#ifdef OSFM_MATCH_V4
    if (nK > 0) flag |= row_pass<true>(sh, tilesB, normB, nB, nK, cand, tilesA, normA, nA, resB, a.ratio, tid);
#else
    if (nK > 0) flag |= query_pass<true>(sh, tilesB, normB, nB, nK, cand, tilesA, normA, hnegA, nA, resB, a.ratio, tid);
#endif
  }
  if (flag) misc[8] = 1;
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
                                unsigned short *res_u16) {
  for (int q = tid; q < nQ; q += kThreads) {
    v4i qa[8];
    load_row(tilesQ, q, qa);
    const int nq = normQ[q];
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
      const float d = ratio < 0.0 ? (float)d2i : sqrtf((float)d2i);  // squared mode: FLANN reports and compares squared distances
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
  exact_direction(tilesC, normC, nC, tilesR, normR, nR, a.ratio, tid, colBI, nullptr);
  if (a.symmetric) exact_direction(tilesR, normR, nR, tilesC, normC, nC, a.ratio, tid, nullptr, rowres);
  __syncthreads();
  emit_matches(a, p, nC, colBI, rowres, misc, tid, qs);
}

}  // namespace

size_t osfm_match4_lds_bytes(int ncap) { return (size_t)2 * kChunkBytes4 + 2 * kChunkCols4 * 4 + 64 + (size_t)ncap * 6; }

static int ensure_kernel_attributes(int device) {
  static OsfmPerDeviceOnce once;
  return once.run(device, []() -> int {
    OSFM_HIP(hipFuncSetAttribute((const void *)match_fused4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    OSFM_HIP(hipFuncSetAttribute((const void *)match_exact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
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
  a.pad_norm = store->d_norms + store->tile_off[store->n_images] * 32;  // first slack row: padding norm
  OSFM_REQUIRE(a.ncap <= OSFM_MAX_FEATURES, OSFM_E_UNSUPPORTED, "more than %d features in an image", OSFM_MAX_FEATURES);
  OSFM_REQUIRE(n_pairs < (1ll << 31), OSFM_E_INVALID, "too many pairs in one launch");
  {
    const int rc = ensure_kernel_attributes(ctx->device);
    if (rc != OSFM_OK) return rc;
  }
  if (!exact_kernel) {
    hipLaunchKernelGGL(match_fused4_kernel, dim3((unsigned)n_pairs), dim3(kThreads), osfm_match4_lds_bytes(a.ncap), stream, a);
  } else {
    const size_t lds = (size_t)a.ncap * 6 + 64;
    hipLaunchKernelGGL(match_exact_kernel, dim3((unsigned)n_pairs), dim3(kThreads), lds, stream, a, d_flags != nullptr ? 1 : 0);
  }
  OSFM_HIP(hipGetLastError());
  return OSFM_OK;
}
