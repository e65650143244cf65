// hahog.hip -- HAHOG feature extraction on the GPU: what features::hahog (opensfm/src/features/src/hahog.cc:125-206)
// computes through the vendored vlfeat (third_party/vlfeat/vl/covdet.c, scalespace.c, imopv.c, sift.c):
//
//   Gaussian scale space, first octave 0, 3 subdivisions + 2 extra levels per octave   scalespace.c:669-697,700-753,756-810, imopv.c:623-681
//   scaled determinant of the Hessian on every level                                    covdet.c:1733-1821
//   3-D local extrema above 0.8 x peak threshold, quadratic refinement, thresholds      covdet.c:1044-1117,1193-1316,1985-2030
//   selection of the strongest target_num_features                                      hahog.cc:12-28,75-96
//   orientation(s) of every feature from a 41 x 41 patch                                covdet.c:2685-2840,2153-2415
//   31 x 31 patch, polar gradient, 4 x 4 x 8 SIFT histogram, normalisation              hahog.cc:163-200, imopv.c vl_imgradient_polar_f, sift.c:1754-1898
//   square root / scaling to [0, 255] as the Python caller does                         opensfm/features.py:516-534
//
// The arithmetic follows the reference operation for operation (float where it is float, double where it promotes, products and sums
// unfused, the taps of every Gaussian in vlfeat's order, histograms accumulated in raster order) so that the scale space, the
// responses and with them the set of detected features are the reference's bit for bit; what can differ is the last bit of the few
// libm calls made per FEATURE on the device (pow for the scale, exp for the patch filter, cos / sin / atan2 of the orientation) --
// they enter float fields, where a last-bit difference of a double survives with probability ~1e-8.
//
// Kernels are plain HBM-bound stencils and per-feature workgroups (no matrix cores: there is no contraction here).
#include <cmath>
#include <cstddef>
#include <cstring>
#include <atomic>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "osfm_internal.h"

namespace {

#ifdef OSFM_DBG_PHASES
// instrumented builds only (tools/hahog_phases.py): 100 MHz ticks of thread 0 of every workgroup, summed per phase
__device__ unsigned long long g_hphase[16];
#define HTICK(var) const unsigned long long var = wall_clock64();
#define HPHASE(i, t0, t1) if (threadIdx.x == 0) atomicAdd(&g_hphase[i], (t1) - (t0));
#else
#define HTICK(var)
#define HPHASE(i, t0, t1)
#endif
constexpr int kFirstSub = -1, kLastSub = 3, kLev = kLastSub - kFirstSub + 1, kRes = 3;
constexpr int kMaxTaps = 65;
constexpr double kPi = 3.141592653589793;          // VL_PI
constexpr float kEpsF = 1.19209290E-07F;          // VL_EPSILON_F
constexpr double kEpsD = 2.220446049250313e-16;    // VL_EPSILON_D

struct Octave {
  float *gss, *css;  // kLev levels of h x w floats each, contiguous
  int w, h;
};
constexpr int kMaxOct = 16;
struct Pyramid {
  Octave oct[kMaxOct];
  int n_oct;
  double base_scale;
  double sigma[kMaxOct][kLev];  // vl_scalespace_get_level_sigma, computed by the host's libm
};

// ---- scale space ---------------------------------------------------------------------------------------------------------------
// vl_imconvcol_vf (imopv.c:120-215 / imopv_sse2.c), VL_PAD_BY_CONTINUITY: dst(y) = sum_j src(clamp(y - W + j)) * filt[2W - j], the sum
// started at 0 and taken in that order, product and sum rounded separately
__global__ void conv_v_kernel(const float *src, float *dst, int w, int h, const float *taps, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  float acc = 0.f;
  for (int j = 0; j <= 2 * W; j++) {
    int p = y - W + j;
    p = p < 0 ? 0 : (p > h - 1 ? h - 1 : p);
    acc = acc + src[(long)p * w + x] * taps[2 * W - j];
  }
  dst[(long)y * w + x] = acc;
}
__global__ void conv_h_kernel(const float *src, float *dst, int w, int h, const float *taps, int W) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const float *row = src + (long)y * w;
  float acc = 0.f;
  for (int j = 0; j <= 2 * W; j++) {
    int p = x - W + j;
    p = p < 0 ? 0 : (p > w - 1 ? w - 1 : p);
    acc = acc + row[p] * taps[2 * W - j];
  }
  dst[(long)y * w + x] = acc;
}
// The two passes in one launch (round 4): a workgroup owns a kSmX x kSmY tile of the output.  The source tile with its halo of W rows /
// columns goes to LDS once (clamped indices: the padding by continuity of both passes), the column pass fills an LDS tile kSmY rows high
// and kSmX + 2 W columns wide -- the columns the row pass of this tile reads, a clamped column being the column pass of the clamped
// column --, the row pass writes the result.  Every output value is the same sequence of float products and sums as conv_v_kernel
// followed by conv_h_kernel (bit-identical); the intermediate image never goes through HBM and each source pixel leaves L2 ~1.6 times
// instead of 2 W + 1.  src and dst must not overlap (other workgroups read the halo).
constexpr int kSmX = 64, kSmY = 32, kSmMaxW = 16, kSmR = 8;
__host__ __device__ constexpr int sm_stride(int W) { return (kSmX + 2 * W) | 1; }  // odd: the row pass has a lane per tile row
// Sliding windows (W is a template parameter so that they live in registers): a thread of the column pass owns kSmR consecutive rows of
// one column and loads the kSmR + 2 W source values once; a thread of the row pass owns kSmR consecutive columns of one row.  The sums
// stay what they were: sum_j value(j) * taps[2 W - j], j ascending from an accumulator of zero, product and sum rounded separately.
// The Hessian response of the level it has just produced comes out of the same launch (_vl_det_hessian_response, the expression of
// hessian_kernel below): the tile then carries a halo of two output pixels (the response of a border pixel is that of its nearest
// interior pixel, one further in), i.e. 60 x 28 pixels of the level per 64 x 32 tile.
constexpr int kSmHalo = 2;
template <int W>
__global__ void __launch_bounds__(256) smooth_fused_kernel(const float *src, float *dst, float *css, int w, int h, const float *taps, float factor, int src_w,
                                                           int src_mul) {  // sample (x, y) of the source image is src[(y src_mul) src_w + x src_mul]: 2 = every second pixel of
                                                                           // every second row of the octave before (copy_and_downsample, scalespace.c:497-520)
  constexpr int SW = sm_stride(W), SH = kSmY + 2 * W, NT = 2 * W + 1;
  __shared__ float S[SW * SH], T[SW * kSmY];  // source tile (later the output tile), column-pass tile
  const int x0 = blockIdx.x * (kSmX - 2 * kSmHalo) - kSmHalo, y0 = blockIdx.y * (kSmY - 2 * kSmHalo) - kSmHalo, tid = threadIdx.x;
  float tp[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) tp[j] = taps[2 * W - j];
  for (int t = tid; t < (kSmX + 2 * W) * SH; t += 256) {
    const int r = t / (kSmX + 2 * W), c = t - r * (kSmX + 2 * W);
    int gy = y0 - W + r, gx = x0 - W + c;
    gy = gy < 0 ? 0 : (gy > h - 1 ? h - 1 : gy);
    gx = gx < 0 ? 0 : (gx > w - 1 ? w - 1 : gx);
    S[r * SW + c] = src[(long)(gy * src_mul) * src_w + gx * src_mul];
  }
  __syncthreads();
  for (int t = tid; t < (kSmX + 2 * W) * (kSmY / kSmR); t += 256) {  // consecutive lanes: consecutive columns
    const int g = t / (kSmX + 2 * W), c = t - g * (kSmX + 2 * W), r0 = g * kSmR;
    float v[kSmR + 2 * W];
#pragma unroll
    for (int k = 0; k < kSmR + 2 * W; k++) v[k] = S[(r0 + k) * SW + c];
#pragma unroll
    for (int i = 0; i < kSmR; i++) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < NT; j++) acc = acc + v[i + j] * tp[j];
      T[(r0 + i) * SW + c] = acc;
    }
  }
  __syncthreads();
  {  // 32 rows x 8 column groups = 256 threads; consecutive lanes: consecutive rows (odd stride: distinct banks)
    const int r = tid % kSmY, c0 = (tid / kSmY) * kSmR;
    float v[kSmR + 2 * W];
#pragma unroll
    for (int k = 0; k < kSmR + 2 * W; k++) v[k] = T[r * SW + c0 + k];
#pragma unroll
    for (int i = 0; i < kSmR; i++) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < NT; j++) acc = acc + v[i + j] * tp[j];
      S[r * SW + c0 + i] = acc;  // the source tile is dead: every thread passed the barrier after its last read of S
    }
  }
  __syncthreads();
  // S[r][c] = the smoothed level at (x0 + c, y0 + r) wherever that is inside the image (a value outside is never read)
  for (int t = tid; t < (kSmX - 2 * kSmHalo) * (kSmY - 2 * kSmHalo); t += 256) {
    const int r = t / (kSmX - 2 * kSmHalo) + kSmHalo, c = t % (kSmX - 2 * kSmHalo) + kSmHalo;
    const int x = x0 + c, y = y0 + r;
    if (x >= w || y >= h) continue;
    dst[(long)y * w + x] = S[r * SW + c];
    const int cc = (x < 1 ? 1 : (x > w - 2 ? w - 2 : x)) - x0, rr = (y < 1 ? 1 : (y > h - 2 ? h - 2 : y)) - y0;
    const float *p = S + rr * SW + cc;
    const float p11 = p[-SW - 1], p12 = p[-SW], p13 = p[-SW + 1], p21 = p[-1], p22 = p[0], p23 = p[1], p31 = p[SW - 1], p32 = p[SW], p33 = p[SW + 1];
    const float Lxx = (-p21 + 2 * p22 - p23);
    const float Lyy = (-p12 + 2 * p22 - p32);
    const float Lxy = ((p11 - p31 - p13 + p33) / 4.0f);
    css[(long)y * w + x] = (Lxx * Lyy - Lxy * Lxy) * factor;
  }
}
typedef void (*smooth_fn)(const float *, float *, float *, int, int, const float *, float, int, int);
template <int W>
struct SmoothTable {
  static smooth_fn get(int q) { return q == W ? smooth_fused_kernel<W> : SmoothTable<W - 1>::get(q); }
};
template <>
struct SmoothTable<0> {
  static smooth_fn get(int) { return nullptr; }
};
// copy_and_downsample by one octave (scalespace.c:497-520): every second pixel of every second row
__global__ void downsample_kernel(const float *src, int w, int h, float *dst, int dw, int dh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x < dw && y < dh) dst[(long)y * dw + x] = src[(long)(2 * y) * w + 2 * x];
}
// _vl_det_hessian_response (covdet.c:1733-1821): interior from the 3 x 3 neighbourhood, the border copies the nearest interior value
__global__ void hessian_kernel(const float *im, float *out, int w, int h, float factor) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int c = x < 1 ? 1 : (x > w - 2 ? w - 2 : x), r = y < 1 ? 1 : (y > h - 2 ? h - 2 : y);
  const float *p = im + (long)r * w + c;
  const float p11 = p[-w - 1], p12 = p[-w], p13 = p[-w + 1], p21 = p[-1], p22 = p[0], p23 = p[1], p31 = p[w - 1], p32 = p[w], p33 = p[w + 1];
  const float Lxx = (-p21 + 2 * p22 - p23);
  const float Lyy = (-p12 + 2 * p22 - p32);
  const float Lxy = ((p11 - p31 - p13 + p33) / 4.0f);
  out[(long)y * w + x] = (Lxx * Lyy - Lxy * Lxy) * factor;
}

// The first level of an octave that is not smoothed (round 6, one launch instead of two): copy_and_downsample of the level of the octave before
// and the Hessian response of the result, the expressions of downsample_kernel and hessian_kernel on the same values
__global__ void downsample_hessian_kernel(const float *src, int sw, float *dst, float *css, int w, int h, float factor) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  dst[(long)y * w + x] = src[(long)(2 * y) * sw + 2 * x];
  const int c = x < 1 ? 1 : (x > w - 2 ? w - 2 : x), r = y < 1 ? 1 : (y > h - 2 ? h - 2 : y);
  auto at = [&](int rr, int cc) { return src[(long)(2 * rr) * sw + 2 * cc]; };
  const float p11 = at(r - 1, c - 1), p12 = at(r - 1, c), p13 = at(r - 1, c + 1), p21 = at(r, c - 1), p22 = at(r, c), p23 = at(r, c + 1), p31 = at(r + 1, c - 1),
              p32 = at(r + 1, c), p33 = at(r + 1, c + 1);
  const float Lxx = (-p21 + 2 * p22 - p23);
  const float Lyy = (-p12 + 2 * p22 - p32);
  const float Lxy = ((p11 - p31 - p13 + p33) / 4.0f);
  css[(long)y * w + x] = (Lxx * Lyy - Lxy * Lxy) * factor;
}

// ---- detection -----------------------------------------------------------------------------------------------------------------
struct Features {  // structure of arrays, capacity cap
  float *x, *y, *sigma, *peak, *edge;
  int *o, *s;
  unsigned long long *key;  // vlfeat's detection order: octaves from the last to the first, then z, y, x
  int *count;
  int cap;
};

__device__ __forceinline__ long vl_floor_d(double x) {
  const long xi = (long)x;
  return (x >= 0 || (double)xi == x) ? xi : xi - 1;
}
__device__ __forceinline__ long vl_floor_f(float x) {
  const long xi = (long)x;
  return (x >= 0 || (float)xi == x) ? xi : xi - 1;
}

// vl_gaussian_elimination for the 3 x 4 system (mathop.c:~860-960, column-major M[i + 3 j]); returns false when singular
__device__ bool solve3(double *M) {
#define MA(i, j) M[(i) + (j)*3]
  for (int j = 0; j < 3; j++) {
    double maxa = 0, maxabsa = 0;
    int maxi = -1;
    for (int i = j; i < 3; i++) {
      const double a = MA(i, j), absa = fabs(a);
      if (absa > maxabsa) {
        maxa = a;
        maxabsa = absa;
        maxi = i;
      }
    }
    if (maxabsa < 1e-10) return false;
    const int i = maxi;
    for (int jj = j; jj < 4; jj++) {
      const double tmp = MA(i, jj);
      MA(i, jj) = MA(j, jj);
      MA(j, jj) = tmp;
      MA(j, jj) /= maxa;
    }
    for (int ii = j + 1; ii < 3; ii++) {
      const double x = MA(ii, j);
      for (int jj = j; jj < 4; jj++) MA(ii, jj) -= x * MA(j, jj);
    }
  }
  for (int i = 2; i > 0; i--)
    for (int ii = i - 1; ii >= 0; ii--) {
      const double x = MA(ii, i);
      MA(ii, 3) -= x * MA(i, 3);
    }
#undef MA
  return true;
}

// vl_find_local_extrema_3 + vl_refine_local_extreum_3 + the thresholds of vl_covdet_detect, all three interior levels of an octave in one
// workgroup.  Round 6: two phases.
//   A  the 26 strict comparisons of a sample as ONE comparison with the largest (smallest) of its neighbours, built from row maxima: a lane owns a
//      column, takes the kExRows + 2 rows of the five levels into registers (50 coalesced loads in flight, every response read once per workgroup
//      instead of 27 times through the L1: the old kernel's 27 unaligned loads per sample were what bound it, 65 us on the first octave), gets the
//      two neighbouring columns from the lanes beside it (DPP wave shifts: lanes 0 and 63 of a wavefront are halo columns, a wavefront yields 62
//      columns) and forms max3 / min3 along x, then along y, then across the levels.  v > every neighbour <=> v > their maximum: the same
//      decisions.  The strict extrema go onto a list in LDS.
//   B  the fp64 refinement, one lane per LISTED sample (1 - 2 % of the samples of a noisy image at OpenSfM's peak threshold of 1e-5).
// The list's order does not matter: the features are put in vlfeat's order of detection by F.key afterwards.
constexpr int kExRows = 8, kExWaveCols = 62, kExCols = 4 * kExWaveCols;
__device__ __forceinline__ float lane_left(float v) {  // the value of lane - 1 (wave_shr:1)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_right(float v) {  // the value of lane + 1 (wave_shl:1)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
// (every octave in ONE launch: a workgroup's path -- fifty loads, the comparisons, a barrier, five dependent fp64 refinement steps -- is ~10 us
//  long whatever the octave's size, and seven launches one behind the other paid it seven times: 125 -> see profiles/r06_hahog_*)
struct ExtremaPlan {
  int first[kMaxOct + 1];  // first workgroup of octave k's tiles (octave 0, the largest, first); first[n_oct] = the grid
  int gx[kMaxOct];         // tiles along x
};
__global__ void __launch_bounds__(256) extrema_kernel(Pyramid py, ExtremaPlan plan, double threshold08, double peak_threshold, double edge_threshold,
                                                      Features F) {
  __shared__ int cand[kExCols * kExRows * (kLev - 2) / 2 + 64];  // level << 19 | row << 16 | x (a strict maximum has no strict maximum beside it)
  __shared__ int ncand;
  int o = 0;
  while (o + 1 < py.n_oct && (int)blockIdx.x >= plan.first[o + 1]) o++;
  const int tile = blockIdx.x - plan.first[o], tile_y = tile / plan.gx[o], tile_x = tile - tile_y * plan.gx[o];
  const float *css = py.oct[o].css;
  const int w = py.oct[o].w, h = py.oct[o].h, last_octave = py.n_oct - 1;
  const double step = (double)(1 << o), base_scale = py.base_scale;
  const long xo = 1, yo = w, zo = (long)w * h;
  const int yb = tile_y * kExRows + 1;
  if (threadIdx.x == 0) ncand = 0;
  __syncthreads();
  {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xs = tile_x * kExCols + wv * kExWaveCols + lane;  // lanes 1 .. 62 decide, lanes 0 and 63 carry the columns beside them
    const int xl = xs < w - 1 ? xs : w - 1;
    float hm[kLev][kExRows + 2], hn[kLev][kExRows + 2], vc[kLev - 2][kExRows], sm[kLev - 2][kExRows], sn[kLev - 2][kExRows];
    {
      float v[kLev][kExRows + 2];
#pragma unroll
      for (int j = 0; j < kExRows + 2; j++) {
        const int yy = yb - 1 + j, yl = yy < h - 1 ? yy : h - 1;
#pragma unroll
        for (int l = 0; l < kLev; l++) v[l][j] = css[l * zo + (long)yl * yo + xl];
      }
#pragma unroll
      for (int j = 0; j < kExRows + 2; j++)
#pragma unroll
        for (int l = 0; l < kLev; l++) {
          const float c = v[l][j], lf = lane_left(c), rt = lane_right(c);
          hm[l][j] = max3f(lf, c, rt);
          hn[l][j] = min3f(lf, c, rt);
          if (l >= 1 && l <= kLev - 2 && j >= 1 && j <= kExRows) {
            vc[l - 1][j - 1] = c;
            sm[l - 1][j - 1] = fmaxf(lf, rt);
            sn[l - 1][j - 1] = fminf(lf, rt);
          }
        }
    }
    const bool col_ok = lane >= 1 && lane <= kExWaveCols && xs <= w - 2;
#pragma unroll
    for (int j = 1; j <= kExRows; j++) {
      float vm[kLev], vn[kLev];  // the 3 x 3 window of every level around (xs, row j)
#pragma unroll
      for (int l = 0; l < kLev; l++) {
        vm[l] = max3f(hm[l][j - 1], hm[l][j], hm[l][j + 1]);
        vn[l] = min3f(hn[l][j - 1], hn[l][j], hn[l][j + 1]);
      }
#pragma unroll
      for (int l = 1; l <= kLev - 2; l++) {
        const float c = vc[l - 1][j - 1];
        const float nbmax = max3f(vm[l - 1], vm[l + 1], max3f(hm[l][j - 1], hm[l][j + 1], sm[l - 1][j - 1]));
        const float nbmin = min3f(vn[l - 1], vn[l + 1], min3f(hn[l][j - 1], hn[l][j + 1], sn[l - 1][j - 1]));
        const bool mx = (double)c >= threshold08 && c > nbmax, mn = (double)c <= -threshold08 && c < nbmin;
        if ((mx || mn) && col_ok && yb - 1 + j <= h - 2) cand[atomicAdd(&ncand, 1)] = ((l - 1) << 19) | ((j - 1) << 16) | xs;
      }
    }
  }
  __syncthreads();
  const int nc = ncand;
  for (int ci = threadIdx.x; ci < nc; ci += 256) {
  const int x0 = cand[ci] & 0xffff, y0 = yb + ((cand[ci] >> 16) & 7), z = (cand[ci] >> 19) + 1;
  // refinement
  int x = x0, y = y0, dx = 0, dy = 0;
  double Dx = 0, Dy = 0, Dz = 0, Dxx = 0, Dyy = 0, Dzz = 0, Dxy = 0, Dxz = 0, Dyz = 0, b[3] = {0, 0, 0};
  const float *pt = nullptr;
  bool ok = true;
#define AT(ddx, ddy, ddz) (*(pt + (ddx)*xo + (ddy)*yo + (ddz)*zo))
  for (int iter = 0; iter < 5; iter++) {
    x += dx;
    y += dy;
    pt = css + x * xo + y * yo + z * zo;
    Dx = 0.5 * (AT(+1, 0, 0) - AT(-1, 0, 0));
    Dy = 0.5 * (AT(0, +1, 0) - AT(0, -1, 0));
    Dz = 0.5 * (AT(0, 0, +1) - AT(0, 0, -1));
    Dxx = (AT(+1, 0, 0) + AT(-1, 0, 0) - 2.0 * AT(0, 0, 0));
    Dyy = (AT(0, +1, 0) + AT(0, -1, 0) - 2.0 * AT(0, 0, 0));
    Dzz = (AT(0, 0, +1) + AT(0, 0, -1) - 2.0 * AT(0, 0, 0));
    Dxy = 0.25 * (AT(+1, +1, 0) + AT(-1, -1, 0) - AT(-1, +1, 0) - AT(+1, -1, 0));
    Dxz = 0.25 * (AT(+1, 0, +1) + AT(-1, 0, -1) - AT(-1, 0, +1) - AT(+1, 0, -1));
    Dyz = 0.25 * (AT(0, +1, +1) + AT(0, -1, -1) - AT(0, -1, +1) - AT(0, +1, -1));
    double M[12] = {Dxx, Dxy, Dxz, Dxy, Dyy, Dyz, Dxz, Dyz, Dzz, -Dx, -Dy, -Dz};
    ok = solve3(M);
    if (!ok) {
      b[0] = b[1] = b[2] = 0;
      break;
    }
    b[0] = M[9];
    b[1] = M[10];
    b[2] = M[11];
    dx = (b[0] > 0.6 && x < w - 2 ? 1 : 0) + (b[0] < -0.6 && x > 1 ? -1 : 0);
    dy = (b[1] > 0.6 && y < h - 2 ? 1 : 0) + (b[1] < -0.6 && y > 1 ? -1 : 0);
    if (dx == 0 && dy == 0) break;
  }
  const double peakScore = AT(0, 0, 0) + 0.5 * (Dx * b[0] + Dy * b[1] + Dz * b[2]);
#undef AT
  const double alpha = (Dxx + Dyy) * (Dxx + Dyy) / (Dxx * Dyy - Dxy * Dxy);
  double edgeScore;
  if (alpha < 0)
    edgeScore = INFINITY;
  else {
    const double t = 0.25 * alpha - 1;
    edgeScore = (0.5 * alpha - 1) + sqrt((t > 0 ? t : 0) * alpha);
  }
  // VlCovDetExtremum3 keeps the refined position and the two scores as floats (covdet.c:994-1004): everything below sees those
  const float rx = (float)(x + b[0]), ry = (float)(y + b[1]), rz = (float)(z + b[2]);
  const float peakF = (float)peakScore, edgeF = (float)edgeScore;
  ok = ok && fabs(b[0]) < 1.5 && fabs(b[1]) < 1.5 && fabs(b[2]) < 1.5 && 0 <= rx && rx <= w - 1 && 0 <= ry && ry <= h - 1 && 0 <= rz &&
       rz <= kLev - 1;
  ok = ok && fabs((double)peakF) > peak_threshold;
  ok = ok && (double)edgeF < edge_threshold;
  if (!ok) continue;
  // o + (refined.z + first) / resolution: a float and integers -- C evaluates the whole exponent in float (covdet.c:2007-2009)
  const float expo = (float)o + (rz + (float)kFirstSub) / (float)kRes;
  const double sigma = base_scale * pow(2.0, (double)expo);
  const int slot = atomicAdd(F.count, 1);
  if (slot >= F.cap) continue;
  F.x[slot] = (float)(rx * step);
  F.y[slot] = (float)(ry * step);
  F.sigma[slot] = (float)sigma;
  F.o[slot] = o;
  F.s[slot] = (int)round((double)rz);
  F.peak[slot] = peakF;
  F.edge[slot] = edgeF;
  F.key[slot] = ((unsigned long long)(last_octave - o) << 56) | ((unsigned long long)z << 48) | ((unsigned long long)y0 << 24) | (unsigned long long)x0;
  }
}

__global__ void iota_kernel(int *a, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}
__global__ void gather_f_kernel(const int *idx, const float *src, float *dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
__global__ void compose_kernel(const int *outer, const int *inner, int *dst, int n) {  // dst[i] = inner[outer[i]]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = inner[outer[i]];
}

// ---- patches -------------------------------------------------------------------------------------------------------------------
// vl_lapack_dlasv2 + vl_svd2 (mathop.c:641-838), singular values only (what the patch extraction needs of an oriented frame)
__device__ __forceinline__ double sgn(double x) { return x >= 0.0 ? 1.0 : -1.0; }
__device__ void svd2_values(const double *Mx, double *d1, double *d2) {
  const double m11 = Mx[0], m21 = Mx[1], m12 = Mx[2], m22 = Mx[3];
  double cu1 = m11, su1 = m21;
  const double norm = sqrt(cu1 * cu1 + su1 * su1);
  cu1 /= norm;
  su1 /= norm;
  const double f = cu1 * m11 + su1 * m21, g = cu1 * m12 + su1 * m22, h = -su1 * m12 + cu1 * m22;
  double svt = 0, cvt = 0, sut = 0, cut = 0, ft = f, gt = g, ht = h, fa = fabs(f), ga = fabs(g), ha = fabs(h), smin = 0, smax = 0;
  int pmax = 1, swap = 0, glarge = 0;
  if (fa < ha) {
    pmax = 3;
    double tmp = ft; ft = ht; ht = tmp;
    tmp = fa; fa = ha; ha = tmp;
    swap = 1;
  }
  if (ga == 0.0) {
    smin = ha;
    smax = fa;
    cut = 1.0; sut = 0.0; cvt = 1.0; svt = 0.0;
  } else {
    if (ga > fa) {
      pmax = 2;
      if ((fa / ga) < kEpsD) {
        glarge = 1;
        smax = ga;
        if (ha > 1.0) smin = fa / (ga / ha);
        else smin = (fa / ga) * ha;
        cut = 1.0; sut = ht / gt; cvt = 1.0; svt = ft / gt;
      }
    }
    if (glarge == 0) {
      const double fmh = fa - ha;
      const double d = (fmh == fa) ? 1.0 : fmh / fa;
      const double q = gt / ft, s = 2.0 - d, dd = d * d, qq = q * q, ss = s * s;
      const double spq = sqrt(ss + qq);
      const double dpq = (d == 0.0) ? fabs(q) : sqrt(dd + qq);
      const double a = 0.5 * (spq + dpq);
      smin = ha / a;
      smax = fa * a;
      double tmp;
      if (qq == 0.0) {
        if (d == 0.0) tmp = sgn(ft) * 2 * sgn(gt);
        else tmp = gt / (sgn(ft) * fmh) + q / s;
      } else
        tmp = (q / (spq + s) + q / (dpq + d)) * (1.0 + a);
      const double tt = sqrt(tmp * tmp + 4.0);
      cvt = 2.0 / tt;
      svt = tmp / tt;
      cut = (cvt + svt * q) / a;
      sut = (ht / ft) * svt / a;
    }
  }
  double cu, su, cv, sv;
  if (swap == 1) { cu = svt; su = cvt; cv = sut; sv = cut; }
  else { cu = cut; su = sut; cv = cvt; sv = svt; }
  double tsign = 1.0;
  if (pmax == 1) tsign = sgn(cv) * sgn(cu) * sgn(f);
  if (pmax == 2) tsign = sgn(sv) * sgn(cu) * sgn(g);
  if (pmax == 3) tsign = sgn(sv) * sgn(su) * sgn(h);
  *d1 = (tsign >= 0 ? 1.0 : -1.0) * smax;
  *d2 = ((tsign * sgn(f) * sgn(h)) >= 0 ? 1.0 : -1.0) * smin;
}

// What vl_covdet_extract_patch_helper (covdet.c:2153-2415) decides once per patch: the level, the frame in level coordinates and the
// box of the padded copy it makes when the patch leaves the image.
struct PatchPlan {
  const float *level;
  int width, height;       // of the level
  double A[4], T[2];       // patch -> level coordinates (T relative to the padded copy when padded)
  bool padded, zero;       // zero: the central band is empty (the reference clears the copy)
  long x0i, y0i, padx0, pady0, padx1, pady1, pw, ph;
  double sigma_level;      // sigma_ of the level
};
// One thread plans one patch (round 6: or_plan_kernel / desc_plan_kernel, a thread per feature, in front of the per-feature kernels.  The plan
// used to be made inside them by the first wavefront of every workgroup -- sixty-four lanes on identical arguments, a lane per octave for the
// search, three wavefronts waiting at the barrier: 8 and 14 us of a 47 / 52 us workgroup.)  The octave search is the reference's loop.
__device__ double plan_patch(const Pyramid &py, double extent, double sigma, const double *A_, const double *T_, double d1, double d2, PatchPlan &P) {
  const int last_octave = py.n_oct - 1;
  const double factor = 1.0 / (d1 < d2 ? d1 : d2);
  long o, s = 0;
  double sigma_ = 0.0;
  {
    // `for (o = first + 1; o <= last; ++o) { s, sigma_ of o; if (factor * sigma_ > sigma) { o--; break; } }`, then o = min(o, last) and s, sigma_
    // of that o once more (covdet.c:2196-2216)
    auto level_of = [&](long oo) {
      long sl = vl_floor_d(log2(sigma / (factor * py.base_scale)) - oo);
      sl = sl > kFirstSub ? sl : kFirstSub;
      sl = sl < kLastSub ? sl : kLastSub;
      s = sl;
      sigma_ = py.base_scale * pow(2.0, oo + (double)sl / kRes);
    };
    for (o = 1; o <= last_octave; ++o) {
      level_of(o);
      if (factor * sigma_ > sigma) {
        o--;
        break;
      }
    }
    o = o < last_octave ? o : last_octave;
    level_of(o);
  }
  P.sigma_level = sigma_;
  const Octave &oc = py.oct[o];
  P.level = oc.gss + (long)(s - kFirstSub) * oc.w * oc.h;
  P.width = oc.w;
  P.height = oc.h;
  const double step = (double)(1L << o);
  for (int i = 0; i < 4; i++) P.A[i] = A_[i] / step;
  P.T[0] = T_[0] / step;
  P.T[1] = T_[1] / step;
  double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
  const double boxx[4] = {extent, extent, -extent, -extent}, boxy[4] = {-extent, extent, extent, -extent};
  for (int i = 0; i < 4; i++) {
    const double x = P.A[0] * boxx[i] + P.A[2] * boxy[i] + P.T[0];
    const double y = P.A[1] * boxx[i] + P.A[3] * boxy[i] + P.T[1];
    x0 = x0 < x ? x0 : x;
    x1 = x1 > x ? x1 : x;
    y0 = y0 < y ? y0 : y;
    y1 = y1 > y ? y1 : y;
  }
  P.x0i = (long)(floor(x0) - 1);
  P.y0i = (long)(floor(y0) - 1);
  const long x1i = (long)(ceil(x1) + 1), y1i = (long)(ceil(y1) + 1);
  P.padded = P.x0i < 0 || x1i > P.width - 1 || P.y0i < 0 || y1i > P.height - 1;
  P.zero = false;
  if (P.padded) {
    P.padx0 = -P.x0i > 0 ? -P.x0i : 0;
    P.pady0 = -P.y0i > 0 ? -P.y0i : 0;
    P.padx1 = x1i - (P.width - 1) > 0 ? x1i - (P.width - 1) : 0;
    P.pady1 = y1i - (P.height - 1) > 0 ? y1i - (P.height - 1) : 0;
    P.pw = x1i - P.x0i + 1;
    P.ph = y1i - P.y0i + 1;
    P.zero = !(P.pady0 < P.ph - P.pady1);
    P.T[0] -= P.x0i;
    P.T[1] -= P.y0i;
  }
  return sigma_;
}
// sample (xi, yi) of the image the bilinear interpolation reads: the level itself, or the padded copy (covdet.c:2296-2330, including
// how its rows are filled: the last two columns of a row repeat the column before them)
__device__ __forceinline__ float plan_read(const PatchPlan &P, long xi, long yi) {
  if (!P.padded) return P.level[yi * P.width + xi];
  if (P.zero) return 0.f;
  long m = yi < P.pady0 ? P.pady0 : (yi > P.ph - P.pady1 - 1 ? P.ph - P.pady1 - 1 : yi);
  const long row = P.y0i + m;
  long c0 = P.x0i < 0 ? 0 : P.x0i;
  c0 = c0 < P.width - 1 ? c0 : P.width - 1;
  long lim = P.pw - P.padx1 - 2 - P.padx0;
  lim = lim > 0 ? lim : 0;
  long k = xi - P.padx0;
  k = k < 0 ? 0 : (k > lim ? lim : k);
  return P.level[row * P.width + c0 + k];
}
// the resampling loop (covdet.c:2360-2405): patch[yyi][xxi], side = 2 resolution + 1; the running coordinates are sums, as there
// Round 4: the four samples of ALL the pixels of a thread (PER = ceil(side^2 / nthreads)) are requested before the first blend -- a plain
// loop waited for one round trip to the level per pixel (26 of the orientation workgroup's 75 us); each pixel's arithmetic is unchanged.
// Round 6: the instruction diet.  The kernels are bound by what a SIMD issues for its four to six resident wavefronts, and a pixel of this
// loop cost several hundred instructions: the running sums redone per pixel (up to 40 + 40 dependent fp64 additions), double -> int64 -> double
// conversions for the floor, 64-bit index arithmetic and the padded copy's clamps on every one of the four reads.  Now the running sums are a
// table in LDS (`hat`: entry q = -extent + stephat + ... + stephat, q additions in the reference's order, filled once per workgroup by
// patch_hat_table), the floor is v_floor_f64 (|x| < 2^31: the same integer, and x - floor(x) the same double), the indices are 32-bit, and a
// patch that lies inside its level (wave-uniform) reads the level directly.  Each pixel's arithmetic is unchanged.
__device__ __forceinline__ void patch_hat_table(double *hat, int resolution, double extent, int lane) {  // one wavefront, a lane per entry
  const int side = 2 * resolution + 1;
  const double stephat = extent / resolution;
  if (lane < side) {
    double a = -extent;
    for (int q = 0; q < lane; q++) a += stephat;
    hat[lane] = a;
  }
}
template <int PER, int G>  // G pixels in flight per thread (all seven of the orientation patch cost a fourth wave per SIMD in registers)
__device__ void sample_patch(const PatchPlan &P, const double *hat, float *patch, int resolution, int tid, int nthreads) {
  const int side = 2 * resolution + 1;
  const double A0 = P.A[0], A1 = P.A[1], A2 = P.A[2], A3 = P.A[3], T0 = P.T[0], T1 = P.T[1];
  const bool padded = P.padded;
  const float *level = P.level;
  const int width = P.width;
#pragma unroll
  for (int g = 0; g < PER; g += G) {
    double wxs[G], wys[G];
    float v00[G], v10[G], v01[G], v11[G];
#pragma unroll
    for (int u = 0; u < G; u++) {
      const int tq = tid + (g + u) * nthreads;
      const int t = (g + u < PER && tq < side * side) ? tq : 0;  // a thread past the end samples pixel 0 and drops it
      const int yyi = t / side, xxi = t - yyi * side;
      const double yhat = hat[yyi], xhat = hat[xxi];
      const double rx = A2 * yhat + T0, ry = A3 * yhat + T1;
      const double x = A0 * xhat + rx, y = A1 * xhat + ry;
      const double fx = floor(x), fy = floor(y);
      const int xi = (int)fx, yi = (int)fy;
      if (!padded) {
        const float *q = level + (yi * width + xi);
        v00[u] = q[0];
        v10[u] = q[1];
        v01[u] = q[width];
        v11[u] = q[width + 1];
      } else {
        v00[u] = plan_read(P, xi, yi);
        v10[u] = plan_read(P, xi + 1, yi);
        v01[u] = plan_read(P, xi, yi + 1);
        v11[u] = plan_read(P, xi + 1, yi + 1);
      }
      wxs[u] = x - fx;
      wys[u] = y - fy;
    }
#pragma unroll
    for (int u = 0; u < G; u++) {
      const int t = tid + (g + u) * nthreads;
      if (g + u >= PER || t >= side * side) continue;
      const double i00 = v00[u], i10 = v10[u], i01 = v01[u], i11 = v11[u], wx = wxs[u], wy = wys[u];
      patch[t] = (float)((1.0 - wy) * ((1.0 - wx) * i00 + wx * i10) + wy * ((1.0 - wx) * i01 + wx * i11));
    }
  }
}

// vl_fast_atan2_f, vl_fast_resqrt_f, vl_fast_sqrt_f, vl_mod_2pi_f (mathop.h)
__device__ __forceinline__ float fast_atan2_f(float y, float x) {
  float angle, r;
  const float c3 = 0.1821F, c1 = 0.9675F;
  const float abs_y = fabsf(y) + kEpsF;
  if (x >= 0) {
    r = (x - abs_y) / (x + abs_y);
    angle = (float)(kPi / 4);
  } else {
    r = (x + abs_y) / (abs_y - x);
    angle = (float)(3 * kPi / 4);
  }
  angle += (c3 * r * r - c1) * r;
  return (y < 0) ? -angle : angle;
}
__device__ __forceinline__ float fast_resqrt_f(float x) {
  const float xhalf = (float)0.5 * x;
  int i = __float_as_int(x);
  i = 0x5f3759df - (i >> 1);
  float u = __int_as_float(i);
  u = u * ((float)1.5 - xhalf * u * u);
  u = u * ((float)1.5 - xhalf * u * u);
  return u;
}
__device__ __forceinline__ float fast_sqrt_f(float x) { return ((double)x < 1e-8) ? 0 : x * fast_resqrt_f(x); }
__device__ __forceinline__ float mod_2pi_f(float x) {
  while (x > (float)(2 * kPi)) x -= (float)(2 * kPi);
  while (x < 0.0F) x += (float)(2 * kPi);
  return x;
}
// vl_imgradient_polar_f on a side x side patch in LDS: modulus and angle of pixel t
__device__ __forceinline__ void polar_gradient(const float *im, int side, int t, float *mod, float *ang) {
  const int y = t / side, x = t - y * side;
  const float *src = im + t;
  float gx, gy;
  if (x == 0) gx = src[1] - src[0];
  else if (x == side - 1) gx = src[0] - src[-1];
  else gx = 0.5 * (src[1] - src[-1]);
  if (y == 0) gy = src[side] - src[0];
  else if (y == side - 1) gy = src[0] - src[-side];
  else gy = 0.5 * (src[side] - src[-side]);
  *mod = fast_sqrt_f(gx * gx + gy * gy);
  *ang = mod_2pi_f((float)(fast_atan2_f(gy, gx) + 2 * kPi));
}

// ---- orientations (vl_covdet_extract_orientations_for_frame, covdet.c:2685-2840; hahog.cc:98-123) --------------------------------
constexpr int kOrRes = 20, kOrSide = 2 * kOrRes + 1, kOrBins = 36, kMaxOr = 4;
constexpr double kOrExtent = 9.0;  // VL_COVDET_AA_PATCH_EXTENT = 3 * VL_COVDET_AA_RELATIVE_INTEGRATION_SIGMA
struct Oriented {
  float *x, *y, *a11, *a21, *a12, *a22;  // capacity 4 x features
};
// vl_imsmooth_f on the 41 x 41 patch in LDS (imopv.c: columns, then rows, padding by continuity), W taps on either side
template <int W>
__device__ __forceinline__ void or_smooth(float *patch, float *tmp, const float *taps1, int tid) {
  float tp[2 * W + 1];
#pragma unroll
  for (int j = 0; j <= 2 * W; j++) tp[j] = taps1[2 * W - j];
  for (int t = tid; t < kOrSide * kOrSide; t += 256) {  // along y
    const int y = t / kOrSide, x = t - y * kOrSide;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j <= 2 * W; j++) {
      int p = y - W + j;
      p = p < 0 ? 0 : (p > kOrSide - 1 ? kOrSide - 1 : p);
      acc = acc + patch[p * kOrSide + x] * tp[j];
    }
    tmp[t] = acc;
  }
  __syncthreads();
  for (int t = tid; t < kOrSide * kOrSide; t += 256) {  // along x
    const int y = t / kOrSide, x = t - y * kOrSide;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j <= 2 * W; j++) {
      int p = x - W + j;
      p = p < 0 ? 0 : (p > kOrSide - 1 ? kOrSide - 1 : p);
      acc = acc + tmp[y * kOrSide + p] * tp[j];
    }
    patch[t] = acc;
  }
  __syncthreads();
}
struct OrPlan {  // what or_plan_kernel hands a workgroup of orientation_kernel
  PatchPlan P;
  float taps[16];  // the smoothing of the patch: W <= 7 (sd <= 1 / stephat)
  int W, pad;
};
// a thread per selected feature: its position and scale gathered from the detections, the patch plan, the taps of vl_imsmooth_f
__global__ void __launch_bounds__(64) or_plan_kernel(Pyramid py, const int *order, const float *Fx, const float *Fy, const float *Fsigma, int n, float *fx,
                                                     float *fy, float *fsigma, OrPlan *plans) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const int src = order[f];
  const float x = Fx[src], y = Fy[src], sgf = Fsigma[src];
  fx[f] = x;
  fy[f] = y;
  fsigma[f] = sgf;
  OrPlan L;
  // the detector's frames are isotropic: A = sigma I, so vl_svd2 returns D = (sigma, sigma), U = V = I and theta0 = atan2(0, 1) = 0
  const double sg = sgf;
  const double A[4] = {sg, 0.0, 0.0, sg}, T[2] = {x, y};
  const double sigma_level = plan_patch(py, kOrExtent, 1.0, A, T, sg, sg, L.P);
  const double sigma1 = sigma_level / sg;
  // vl_imsmooth_f(patch, deltaSigma1 / stephat, deltaSigma2 / stephat): one filter, both directions (sigma1 = sigma2)
  const double t = 1.0 - sigma1 * sigma1;
  const double delta = sqrt(t > 0 ? t : 0), stephat = kOrExtent / kOrRes;
  const double sd = delta / stephat;
  const int W = (int)ceil(sd * 3.0);
  double g[8];
  float mass = (float)1.0;
  for (int i = 1; i <= W && i < 8; i++) {
    const double xx = (double)i / sd;
    g[i] = exp(-0.5 * xx * xx);
    mass += g[i] + g[i];
  }
  for (int i = 0; i < 16; i++) L.taps[i] = 0.f;
  L.taps[W] = 1.0f / mass;
  for (int i = 1; i <= W && i < 8; i++) {
    L.taps[W - i] = (float)g[i] / mass;
    L.taps[W + i] = (float)g[i] / mass;
  }
  L.W = W;
  L.pad = 0;
  plans[f] = L;
}
__global__ void __launch_bounds__(256) orientation_kernel(Pyramid py, const OrPlan *plans, int n, const double *aa_mask, int *n_or,
                                                          double *or_angle /* n x 4 */) {
  // 37 KB: four workgroups per CU.  The patch lives in the space of the per-pixel records (the gradient of every pixel is taken into
  // registers first, then the patch is dead and the records are written); the bins are bytes
  __shared__ double2 hc[kOrSide * kOrSide];  // what the pixel adds to its bin and to the next one
  __shared__ float tmp[kOrSide * kOrSide];
  __shared__ unsigned char hbin[kOrSide * kOrSide];
  float *patch = reinterpret_cast<float *>(hc);
  __shared__ float taps1[kMaxTaps];
  __shared__ int W1;
  __shared__ PatchPlan P;
  __shared__ double hist[kOrBins], hist2[kOrBins], hat[kOrSide];
  constexpr int kOrChunks = (kOrSide * kOrSide + 255) / 256;
  __shared__ int start[kOrBins + 1];
  __shared__ unsigned short cwa[kOrChunks][4][kOrBins];  // pixels of (chunk, wavefront) in a bin, then their exclusive prefix
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= n) return;
  HTICK(h0)
  static_assert(sizeof(OrPlan) % 4 == 0 && offsetof(OrPlan, taps) == sizeof(PatchPlan) && sizeof(PatchPlan) % 4 == 0, "layout of OrPlan");
  if (tid < 64) {  // wave 0: the plan or_plan_kernel made
    const int *src = reinterpret_cast<const int *>(plans + f);
    for (int q = tid; q < (int)(sizeof(PatchPlan) / 4); q += 64) reinterpret_cast<int *>(&P)[q] = src[q];
    if (tid < 16) taps1[tid] = plans[f].taps[tid];
    if (tid == 0) W1 = plans[f].W;
  }
  else if (tid < 128)  // wave 1, beside wave 0's plan
    patch_hat_table(hat, kOrRes, kOrExtent, tid - 64);
  if (tid < kOrBins) hist[tid] = 0.0;
  __syncthreads();
  HTICK(h1)
  sample_patch<(kOrSide * kOrSide + 255) / 256, 4>(P, hat, patch, kOrRes, tid, 256);
  __syncthreads();
  HTICK(h2)
  switch (W1) {  // (uniform over the workgroup; W <= 7.  Round 6: the taps in registers, the tap loops unrolled -- the same sums in the same order)
    case 0: or_smooth<0>(patch, tmp, taps1, tid); break;
    case 1: or_smooth<1>(patch, tmp, taps1, tid); break;
    case 2: or_smooth<2>(patch, tmp, taps1, tid); break;
    case 3: or_smooth<3>(patch, tmp, taps1, tid); break;
    case 4: or_smooth<4>(patch, tmp, taps1, tid); break;
    case 5: or_smooth<5>(patch, tmp, taps1, tid); break;
    case 6: or_smooth<6>(patch, tmp, taps1, tid); break;
    default: or_smooth<7>(patch, tmp, taps1, tid); break;
  }
  HTICK(h3)
  // per pixel, in parallel: the bin and the two products the sequential loop adds (covdet.c:2769-2781)
  const double binExtent = 2 * kPi / kOrBins;
  constexpr int kPerThread = (kOrSide * kOrSide + 255) / 256;
  float gm[kPerThread], ga[kPerThread];
#pragma unroll
  for (int u = 0; u < kPerThread; u++) {
    const int t = tid + 256 * u;
    gm[u] = ga[u] = 0.f;
    if (t < kOrSide * kOrSide) polar_gradient(patch, kOrSide, t, &gm[u], &ga[u]);
  }
  __syncthreads();  // the last read of the patch: its space becomes the records
#pragma unroll
  for (int u = 0; u < kPerThread; u++) {
    const int t = tid + 256 * u;
    if (t >= kOrSide * kOrSide) continue;
    const float fm = gm[u], fa = ga[u];
    const double modulus = fm, angle = fa, weight = aa_mask[t];
    const double xx = angle / binExtent;
    const double fb = floor(xx);  // vl_floor_d: 0 <= angle <= 2 pi, the same integer without the double -> int64 -> double round trip ...
    const int bin = ((int)fb + kOrBins) % kOrBins;  // ... and a 32-bit remainder instead of two 64-bit ones (round 6)
    const double w2 = xx - fb, w1 = 1.0 - w2;
    hbin[t] = (unsigned char)bin;
    hc[t] = make_double2(w1 * (modulus * weight), w2 * (modulus * weight));
  }
  __syncthreads();
  HTICK(h4)
  // Bin b adds, in raster order, what the sequential loop adds to it -- and nothing else: the 2 x 1 681 records (pixel t gives record 2t
  // = its first product to bin hbin[t] and record 2t + 1 = its second product to the next bin) are sorted by bin with a STABLE counting
  // sort, so a lane walks only its own ~93 records, in the reference's order.  (Scanning all pixels per bin, even branch-free, was a
  // 190-cycle dependent step 1 681 times: 148 of the kernel's 187 us per workgroup, profiles/r03_hahog_phases_before_sort.txt.)
  // Rank of a record inside its bin = number of earlier pixels whose hbin is the bin (their first records) or the bin before it (their
  // second records).
  unsigned short *order = reinterpret_cast<unsigned short *>(tmp);  // 2 x 1 681 record ids: exactly the smoothing buffer, free by now
  // Round 6: three barriers instead of twenty-one.  Pass 1 (no barrier): every chunk's ranks inside the wavefront and the wavefront's count per bin
  // (the lanes holding a bin value come from SIX ballots, one per bit of the bin index: mask(t) = valid & AND_k (bit k of t ? B_k : ~B_k); a lane
  // needs four such masks -- its bin, the bins beside it and, as the counter of bin `lane`, the bin whose number it carries).  Then wave 0: lane b
  // turns bin b's 7 x 4 counts into their exclusive prefix (chunk-major, wave-minor: the raster order), the totals give start[] by a scan over
  // the lanes.  Pass 2: the records go to their places.  (Until round 6: a running count per bin carried from chunk to chunk between barriers, the
  // totals from 1 681 LDS atomics, start[] by one thread.)  The same ranks.
  unsigned pk[kOrChunks];
  {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int ci = 0; ci < kOrChunks; ci++) {
      const int t = ci * 256 + tid;
      const bool valid = t < kOrSide * kOrSide;
      const int b = valid ? hbin[t] : -1;
      const int bm1 = valid ? (b + kOrBins - 1) % kOrBins : -1, bp1 = valid ? (b + 1) % kOrBins : -1;
      unsigned long long Bk[6];
#pragma unroll
      for (int k = 0; k < 6; k++) Bk[k] = __ballot(valid && ((b >> k) & 1));
      const unsigned long long Vm = __ballot(valid), below = (1ull << lane) - 1ull;
      auto lanes_with = [&](int tv) {
        unsigned long long m = Vm;
#pragma unroll
        for (int k = 0; k < 6; k++) m &= ((tv >> k) & 1) ? Bk[k] : ~Bk[k];
        return m;
      };
      const int p_0 = __popcll(lanes_with(b) & below), p_m1 = __popcll(lanes_with(bm1) & below), p_p1 = __popcll(lanes_with(bp1) & below);
      pk[ci] = (unsigned)p_0 | ((unsigned)p_m1 << 8) | ((unsigned)p_p1 << 16);
      if (lane < kOrBins) cwa[ci][w][lane] = (unsigned short)__popcll(lanes_with(lane));
    }
  }
  __syncthreads();
  if (tid < 64) {  // wave 0
    int totb = 0;
    if (tid < kOrBins) {
      unsigned short *cell = &cwa[0][0][0] + tid;
      int v[kOrChunks * 4];
#pragma unroll
      for (int q = 0; q < kOrChunks * 4; q++) v[q] = cell[q * kOrBins];
#pragma unroll
      for (int q = 0; q < kOrChunks * 4; q++) {
        cell[q * kOrBins] = (unsigned short)totb;
        totb += v[q];
      }
    }
    const int prev = __shfl(totb, (tid + kOrBins - 1) % kOrBins);
    const int cnt = tid < kOrBins ? totb + prev : 0;  // records of bin b: its pixels' first products and the second products of the bin before it
    int inc = cnt;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const int o = __shfl_up(inc, m);
      if (tid >= m) inc += o;
    }
    if (tid < kOrBins) start[tid] = inc - cnt;
    if (tid == kOrBins - 1) start[kOrBins] = inc;
  }
  __syncthreads();
  {
    const int w = tid >> 6;
#pragma unroll
    for (int ci = 0; ci < kOrChunks; ci++) {
      const int t = ci * 256 + tid;
      if (t < kOrSide * kOrSide) {
        const int b = hbin[t], bm1 = (b + kOrBins - 1) % kOrBins, bp1 = (b + 1) % kOrBins;
        const int e_0 = cwa[ci][w][b] + (int)(pk[ci] & 255u), e_m1 = cwa[ci][w][bm1] + (int)((pk[ci] >> 8) & 255u),
                  e_p1 = cwa[ci][w][bp1] + (int)((pk[ci] >> 16) & 255u);
        order[start[b] + e_0 + e_m1] = (unsigned short)(2 * t);
        order[start[bp1] + e_p1 + e_0] = (unsigned short)(2 * t + 1);
      }
    }
  }
  __syncthreads();
  if (tid < kOrBins) {
    const double *rec = reinterpret_cast<const double *>(hc);
    const int k1 = start[tid + 1];
    double hsum = 0.0;
    int k = start[tid];
    for (; k + 8 <= k1; k += 8) {  // eight records in flight, added in order
      int id[8];
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) id[u] = order[k + u];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = rec[id[u]];
#pragma unroll
      for (int u = 0; u < 8; u++) hsum += v[u];
    }
    for (; k < k1; k++) hsum += rec[order[k]];
    hist[tid] = hsum;
  }
  __syncthreads();
  HTICK(h5)
  // six passes of the circular box filter (covdet.c:2786-2799).  The in-place loop reads the OLD left neighbour (prev), the old centre
  // and the old right neighbour, the last bin the old first one: a Jacobi step, a lane per bin, same operation order
  for (int iter = 0; iter < 6; iter++) {
    const double *src = (iter & 1) ? hist2 : hist;
    double *dst = (iter & 1) ? hist : hist2;
    if (tid < kOrBins) dst[tid] = (src[(tid + kOrBins - 1) % kOrBins] + src[tid] + src[(tid + 1) % kOrBins]) / 3.0;
    __syncthreads();
  }
  // peaks (covdet.c:2801-2830): a lane per bin decides, the first four in bin order are kept (the loop's break), lane 0 sorts them
  __shared__ double pk_ang[kMaxOr], pk_sc[kMaxOr];
  __shared__ int pk_cnt;
  if (tid < 64) {  // wave 0
    bool peak = false;
    double a = 0.0, h0 = 0.0;
    if (tid < kOrBins) {
      double maxPeak = 0;
#pragma unroll
      for (int i = 0; i < kOrBins; i++) maxPeak = maxPeak > hist[i] ? maxPeak : hist[i];
      const int i = tid;
      h0 = hist[i];
      const double hm = hist[(i - 1 + kOrBins) % kOrBins], hp = hist[(i + 1 + kOrBins) % kOrBins];
      peak = h0 > 0.8 * maxPeak && h0 > hm && h0 > hp;
      if (peak) {
        const double di = -0.5 * (hp - hm) / (hp + hm - 2 * h0);
        a = binExtent * (i + di) + 0.0;  // + theta0
      }
    }
    const unsigned long long mask = __ballot(peak);
    const int rank = __popcll(mask & ((1ull << tid) - 1ull));
    if (peak && rank < kMaxOr) {
      pk_ang[rank] = a;
      pk_sc[rank] = h0;
    }
    if (tid == 0) {
      const int c = __popcll(mask);
      pk_cnt = c < kMaxOr ? c : kMaxOr;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int cnt = pk_cnt;
    // qsort by decreasing score (glibc: merge sort, stable)
    for (int i = 1; i < cnt; i++)
      for (int j = i; j > 0 && pk_sc[j] > pk_sc[j - 1]; j--) {
        double t0 = pk_sc[j]; pk_sc[j] = pk_sc[j - 1]; pk_sc[j - 1] = t0;
        t0 = pk_ang[j]; pk_ang[j] = pk_ang[j - 1]; pk_ang[j - 1] = t0;
      }
    n_or[f] = cnt;
    for (int i = 0; i < cnt; i++) or_angle[(long)f * kMaxOr + i] = pk_ang[i];
  }
  HTICK(h6)
  HPHASE(0, h0, h1) HPHASE(1, h1, h2) HPHASE(2, h2, h3) HPHASE(3, h3, h4) HPHASE(4, h4, h5) HPHASE(5, h5, h6) HPHASE(6, h0, h6)
#ifdef OSFM_DBG_PHASES
  if (tid == 0) atomicAdd(&g_hphase[7], 1ull);
#endif
}
// exclusive scan of the orientation counts, one workgroup (a few thousand features)
__global__ void __launch_bounds__(1024) scan_kernel(const int *cnt, int n, int *off) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  int s = 0;
  for (int i = tid * per; i < (tid + 1) * per && i < n; i++) s += cnt[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int i = 0; i < 1024; i++) {
      const int v = part[i];
      part[i] = acc;
      acc += v;
    }
    off[n] = acc;
  }
  __syncthreads();
  int acc = part[tid];
  for (int i = tid * per; i < (tid + 1) * per && i < n; i++) {
    off[i] = acc;
    acc += cnt[i];
  }
}
// hahog.cc:98-123: one oriented copy of the frame per orientation
__global__ void orient_frames_kernel(const float *fx, const float *fy, const float *fsigma, const int *off, const int *n_or, const double *or_angle,
                                     int n, Oriented R) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const double A[4] = {fsigma[f], 0.0f, 0.0f, fsigma[f]};  // a11, a21, a12, a22
  for (int j = 0; j < n_or[f]; j++) {
    const double r1 = cos(or_angle[(long)f * kMaxOr + j]), r2 = sin(or_angle[(long)f * kMaxOr + j]);
    const int q = off[f] + j;
    R.x[q] = fx[f];
    R.y[q] = fy[f];
    R.a11[q] = (float)(+A[0] * r1 + A[2] * r2);
    R.a21[q] = (float)(+A[1] * r1 + A[3] * r2);
    R.a12[q] = (float)(-A[0] * r2 + A[2] * r1);
    R.a22[q] = (float)(-A[1] * r2 + A[3] * r1);
  }
}

// ---- descriptors (hahog.cc:163-200; sift.c:1754-1898) -----------------------------------------------------------------------------
constexpr int kDRes = 15, kDSide = 2 * kDRes + 1, kNBO = 8, kNBP = 4;
constexpr double kDExtent = 7.5;
// What a pixel of the 31 x 31 patch contributes apart from its gradient does not depend on the feature (round 6): hahog.cc describes every patch
// at its centre with angle0 = pi / 2 and one sigma, so the normalised coordinates (nx, ny), the Gaussian window, the lower spatial bins and the two
// spatial fractions of pixel t -- two fp64 divisions, the exp-table interpolation, three float -> int64 floors and eight LDS atomics per pixel --
// are the same 961 values for all ~5 000 features of an image, and so are the pixels that feed a spatial bin.  desc_table_kernel computes
// them once per context with the expressions the descriptor kernel used to evaluate per feature (the same bits); descriptor_kernel reads them.
constexpr int kDescListCap = (kDSide * kDSide + 3) / 4 * 4;
constexpr unsigned kDescPad = 0x8000u;
static_assert(kDSide * kDSide <= 1024 && (kDescListCap * 2) % 8 == 0, "list entries: ten bits of pixel, 8-byte rows");
struct DescTable {
  float4 pix[kDSide * kDSide];               // window, nx - (binx + 0.5), ny - (biny + 0.5), bits of (binx + 128) | (biny + 128) << 8
  // per spatial bin (bx + 2) + 4 (by + 2): the pixels that feed it, in raster order: t | (2 dx + dy) << 10 (dx = bx - binx, dy = by - biny, both in {0, 1}); padded to a
  // multiple of four with kDescPad
  unsigned short list[kNBP * kNBP][kDescListCap];
  int nlist[kNBP * kNBP];
};
__global__ void __launch_bounds__(256) desc_table_kernel(const double *expn_tab, double st0, double ct0, double sigma_d, DescTable *tab) {
  const int tid = threadIdx.x;
  // (sift.c:1806-1850; x = y = 15, so xi = yi = 15 and every pixel of the patch is inside the window W = 21)
  const double x0 = (double)(kDSide - 1) / 2, y0 = (double)(kDSide - 1) / 2;
  const double SBP = 3.0 * sigma_d + kEpsD;
  for (int t = tid; t < kDSide * kDSide; t += 256) {
    const int py_ = t / kDSide, px_ = t - py_ * kDSide;
    const float dx = (float)(px_ - x0), dy = (float)(py_ - y0);
    const float nx = (float)((ct0 * dx + st0 * dy) / SBP);
    const float ny = (float)((-st0 * dx + ct0 * dy) / SBP);
    const float wsigma = (float)(kNBP / 2);
    double ex = (nx * nx + ny * ny) / (2.0 * wsigma * wsigma), win_d;
    if (ex > 25.0) win_d = 0.0;
    else {
      ex *= 256 / 25.0;
      const int i = (int)vl_floor_d(ex);
      const double r = ex - i, a = expn_tab[i], b = expn_tab[i + 1];
      win_d = a + r * (b - a);
    }
    const float win = (float)win_d;
    const int binx = (int)vl_floor_f((float)(nx - 0.5)), biny = (int)vl_floor_f((float)(ny - 0.5));
    tab->pix[t] = make_float4(win, (float)(nx - (binx + 0.5)), (float)(ny - (biny + 0.5)), __int_as_float((binx + 128) | ((biny + 128) << 8)));
  }
  __syncthreads();
  if (tid < kNBP * kNBP) {
    const int bx = tid % kNBP - kNBP / 2, by = tid / kNBP - kNBP / 2;
    int n = 0;
    for (int t = 0; t < kDSide * kDSide; t++) {  // (the pix entries of this workgroup's other threads: the barrier above made them visible)
      const int cw = __float_as_int(tab->pix[t].w);
      const int dbinx = bx - ((cw & 255) - 128), dbiny = by - (((cw >> 8) & 255) - 128);
      if ((unsigned)(dbinx | dbiny) <= 1u) tab->list[tid][n++] = (unsigned short)(t | ((2 * dbinx + dbiny) << 10));
    }
    while (n % 4) tab->list[tid][n++] = (unsigned short)kDescPad;
    tab->nlist[tid] = n;
  }
}
// a thread per oriented feature: the keypoint row (hahog.cc:188-196) and the plan of its 31 x 31 patch
__global__ void __launch_bounds__(64) desc_plan_kernel(Pyramid py, Oriented R, int n, float *points, PatchPlan *plans) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const float a11 = R.a11[f], a21 = R.a21[f], a12 = R.a12[f], a22 = R.a22[f];
  {
    const float det = a11 * a22 - a12 * a21;
    const float size = sqrtf(fabsf(det));
    const float angle = (float)(atan2f(a21, a11) * 180.0f / M_PI);
    points[4 * (long)f + 0] = R.x[f];
    points[4 * (long)f + 1] = R.y[f];
    points[4 * (long)f + 2] = size;
    points[4 * (long)f + 3] = angle;
  }
  const double A[4] = {a11, a21, a12, a22}, T[2] = {R.x[f], R.y[f]};
  double d1, d2;
  svd2_values(A, &d1, &d2);
  PatchPlan L;
  plan_patch(py, kDExtent, 1.0, A, T, d1, d2, L);
  plans[f] = L;
}
__global__ void __launch_bounds__(256) descriptor_kernel(const PatchPlan *plans, int n, const DescTable *tab, int flags, float *desc) {
  __shared__ float patch[kDSide * kDSide];
  // Round 6: what a pixel gives each of the 2 x 2 spatial bins it feeds -- (window x modulus) |1 - dx - rx| |1 - dy - ry|, the first three factors of the
  // reference's product in its order -- is formed once per pixel, and a bin walks the LIST of the pixels that feed it (DescTable::list: the same for every
  // feature), not a rectangle of candidates it has to test: ~19 instructions a walked pixel instead of 41, and the walk is nearly all this kernel issues
  __shared__ float sbase[4 * kDSide * kDSide];  // at [4 t + 2 dx + dy]
  __shared__ float snt[kDSide * kDSide];        // the orientation coordinate nt: lower bin floor(nt), fraction nt - floor(nt)
  __shared__ float descr[kNBO * kNBP * kNBP];
  __shared__ float snorm;
  __shared__ PatchPlan P;
  __shared__ double hat[kDSide];
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= n) return;
  if (tid >= 128 && tid < 192) patch_hat_table(hat, kDRes, kDExtent, tid - 128);  // wave 2, beside wave 0's plan
  HTICK(h0)
  if (tid < (int)(sizeof(PatchPlan) / 4)) reinterpret_cast<int *>(&P)[tid] = reinterpret_cast<const int *>(plans + f)[tid];  // desc_plan_kernel's
  __syncthreads();
  HTICK(h1)
  sample_patch<(kDSide * kDSide + 255) / 256, 2>(P, hat, patch, kDRes, tid, 256);
  __syncthreads();
  HTICK(h2)
  // per pixel: window x modulus, the lower bin of the 2 x 2 x 2 it feeds and the three fractions (sift.c:1806-1850): the gradient and its
  // orientation bin here, everything else from the table
  for (int t = tid; t < kDSide * kDSide; t += 256) {
    float mod, angle;
    polar_gradient(patch, kDSide, t, &mod, &angle);
    const float theta = mod_2pi_f((float)(angle - (kPi / 2)));
    const float nt = (float)(kNBO * theta / (2 * kPi));
    const float4 c = tab->pix[t];
    const float wm = c.x * mod, rx = c.y, ry = c.z;
    const float ax0 = fabsf(1 - 0 - rx), ax1 = fabsf(1 - 1 - rx), ay0 = fabsf(1 - 0 - ry), ay1 = fabsf(1 - 1 - ry);
    *reinterpret_cast<float4 *>(sbase + 4 * t) = make_float4(wm * ax0 * ay0, wm * ax0 * ay1, wm * ax1 * ay0, wm * ax1 * ay1);
    snt[t] = nt;
  }
  __syncthreads();
  HTICK(h3)
  // Bin (bx, by, bt) adds, in raster order, what the sequential loop adds to it.  A pixel feeds bin (bx, by, .) only with binx in
  // {bx - 1, bx} and biny in {by - 1, by}: the lane walks, in raster order, only the rows and the columns whose measured ranges of binx
  // and biny admit that (a superset for any patch rotation; ~13 x 13 of the 31 x 31 pixels).  Inside the walk everything is branch-free
  // with unconditional loads (two per pixel), so that the steps pipeline: a pixel that does not feed the bin adds +0.0f, which leaves a
  // sum of non-negative terms unchanged; a pixel feeds a bin through at most one of its two orientation bins.  (All 961 pixels per bin
  // were 130 of the kernel's 166 us per workgroup, profiles/r03_hahog_phases_before_sort.txt.)
  if (tid < kNBO * kNBP * kNBP) {
    const int bt = tid % kNBO;
    const unsigned short *lst = tab->list[tid / kNBO];
    const int n = tab->nlist[tid / kNBO];
    float acc = 0.f;
    uint2 nxt = *reinterpret_cast<const uint2 *>(lst);  // four entries; the next four are requested a step ahead
    for (int k = 0; k < n; k += 4) {
      const uint2 e4 = nxt;
      if (k + 4 < n) nxt = *reinterpret_cast<const uint2 *>(lst + k + 4);
      const unsigned e[4] = {e4.x & 0xffffu, e4.x >> 16, e4.y & 0xffffu, e4.y >> 16};
      float base[4], nt[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {  // four pixels of the walk in flight (a padding entry adds +0.0f), added in raster order
        const int t = (int)(e[u] & 1023u);
        base[u] = sbase[4 * t + (int)((e[u] >> 10) & 3u)];
        nt[u] = snt[t];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool on = !(e[u] & kDescPad);
        const float fl = floorf(nt[u]);  // vl_floor_f: 0 <= nt <= 8
        const float rt = nt[u] - fl;
        const unsigned sb = (unsigned)fl;  // 0 .. 8: the remainders below are masks
        const int b0 = (int)(sb & (kNBO - 1)), b1 = (int)((sb + 1) & (kNBO - 1));
        static_assert((kNBO & (kNBO - 1)) == 0, "kNBO is a power of two");
        const float v0 = base[u] * fabsf(1 - 0 - rt);
        const float v1 = base[u] * fabsf(1 - 1 - rt);
        acc += (on && b0 == bt) ? v0 : ((on && b1 == bt) ? v1 : 0.0f);
      }
    }
    descr[tid] = acc;
  }
  __syncthreads();
  HTICK(h4)
  // normalise, clamp at 0.2, normalise (sift.c:1872-1896, norm_thresh = 0): the sum in index order by one lane, the divisions by all
  for (int pass = 0; pass < 2; pass++) {
    if (tid == 0) {
      float norm = 0.0f;
#pragma unroll
      for (int i = 0; i < 128; i++) norm += descr[i] * descr[i];
      snorm = fast_sqrt_f(norm) + kEpsF;
    }
    __syncthreads();
    if (tid < 128) {
      float v = descr[tid] / snorm;
      if (pass == 0 && (double)v > 0.2) v = (float)0.2;
      descr[tid] = v;
    }
    __syncthreads();
  }
  if (tid < 128) {
    float v = descr[tid];
    if (flags & OSFM_HAHOG_ROOT) v = sqrtf(v);                       // np.sqrt (features.py:526)
    if (flags & OSFM_HAHOG_UCHAR) {                                  // (uchar_scaling * desc).clip(0, 255).round()  (features.py:527-534)
      v = ((flags & OSFM_HAHOG_ROOT) ? 362.0f : 512.0f) * v;
      v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
      v = rintf(v);
    }
    desc[128 * (long)f + tid] = v;
  }
  HTICK(h5)
  HPHASE(8, h0, h1) HPHASE(9, h1, h2) HPHASE(10, h2, h3) HPHASE(11, h3, h4) HPHASE(12, h4, h5) HPHASE(13, h0, h5)
#ifdef OSFM_DBG_PHASES
  if (tid == 0) atomicAdd(&g_hphase[14], 1ull);
#endif
}

// _vl_new_gaussian_fitler_f (imopv.c:623-643)
std::vector<float> gaussian_taps(double sigma, int *W) {
  const long width = (long)std::ceil(sigma * 3.0);
  std::vector<float> filt((size_t)(2 * width + 1));
  float mass = (float)1.0;
  filt[(size_t)width] = 1.0f;
  for (long i = 1; i <= width; ++i) {
    const double x = (double)i / sigma;
    const double g = std::exp(-0.5 * x * x);
    mass += g + g;
    filt[(size_t)(width - i)] = (float)g;
    filt[(size_t)(width + i)] = (float)g;
  }
  for (size_t i = 0; i < filt.size(); i++) filt[i] /= mass;
  *W = (int)width;
  return filt;
}

// grey levels 0 .. 255 -> float32 in [0, 1]: numpy's image.astype(np.float32) / 255 (one correctly rounded float32 division per pixel)
__global__ void __launch_bounds__(256) u8_to_unit_float_kernel(const unsigned char *src, float *dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (float)src[i] / 255.0f;
}

// Device memory of one call: two slabs from the context's cache of blocks (the pyramid and fixed-size buffers; what depends on the number
// of detections), handed out in 256-byte steps -- a dozen hipMalloc / hipFree pairs per image cost more than the kernels
struct Slab {
  OsfmPoolBuf buf;
  size_t used = 0;
  bool overflow = false;
  template <typename T>
  T *take(size_t n) {
    const size_t bytes = ((n ? n : 1) * sizeof(T) + 255) / 256 * 256;
    if (used + bytes > buf.bytes) {
      overflow = true;
      return nullptr;
    }
    T *p = (T *)((char *)buf.p + used);
    used += bytes;
    return p;
  }
};
inline size_t padded(size_t n, size_t elem) { return ((n ? n : 1) * elem + 255) / 256 * 256; }

inline dim3 grid2(int w, int h, int z = 1) { return dim3((unsigned)((w + 255) / 256), (unsigned)h, (unsigned)z); }

}  // namespace

#ifdef OSFM_DBG_PHASES
extern "C" int osfm_dbg_hahog_phases(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hphase), sizeof(g_hphase)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_hphase), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
// The tables that do not depend on the image (round 6: once per context, not once per image -- 1 938 libm exponentials on the host, a pageable
// upload and the table kernel were ~40 us of every call): the orientation mask (covdet.c:1536-1548) and fast_expn's table (sift.c:714-720) from the
// host's libm, then the descriptor's per-pixel table by desc_table_kernel.  Made under the context's hahog_mu and complete when this returns.
// The Gaussians of the scale space, from the host's libm: octave o, level l -> slot o (kLev + 1) + l (vl_scalespace: the first level of an octave
// from the image or from the previous octave's level, scalespace.c:741-752 / :797-809; the others from the level before, :675-693 with its sqrtf)
struct TapPlan {
  std::vector<float> taps;          // kMaxTaps per slot
  int W[(kLev + 1) * kMaxOct];      // half-width per slot, -1: none
  bool first[kMaxOct];              // the octave's first level is smoothed
  int octaves_ok = 0;               // octaves whose filters fit kMaxTaps
};
static const TapPlan &tap_plan() {
  static const TapPlan plan = []() {
    TapPlan t;
    t.taps.assign((size_t)kMaxTaps * (kLev + 1) * kMaxOct, 0.f);
    for (int &w : t.W) w = -1;
    const double base_scale = 1.6 * std::pow(2.0, 1.0 / kRes);
    auto sigma_of = [&](int o, int s) { return base_scale * std::pow(2.0, o + (double)s / kRes); };
    auto set_taps = [&](int slot, double sigma) -> bool {
      int W;
      const std::vector<float> f = gaussian_taps(sigma, &W);
      if ((int)f.size() > kMaxTaps) return false;
      memcpy(t.taps.data() + (size_t)slot * kMaxTaps, f.data(), f.size() * sizeof(float));
      t.W[slot] = W;
      return true;
    };
    bool ok = true;
    for (int o = 0; o < kMaxOct && ok; o++) {
      const double step = std::pow(2.0, o);
      const int base = o * (kLev + 1);
      const double sigma = sigma_of(o, kFirstSub);
      const double prev = o == 0 ? 0.5 : sigma_of(o - 1, std::min(kFirstSub + kRes, kLastSub));
      t.first[o] = sigma > prev;
      if (t.first[o]) ok = ok && set_taps(base, std::sqrt(sigma * sigma - prev * prev) / step);
      for (int s = kFirstSub + 1; s <= kLastSub && ok; s++) {
        const double sg = sigma_of(o, s), pv = sigma_of(o, s - 1);
        const double delta = sqrtf(sg * sg - pv * pv);
        ok = ok && set_taps(base + (s - kFirstSub), delta / step);
      }
      if (ok) t.octaves_ok = o + 1;
    }
    return t;
  }();
  return plan;
}
static int hahog_tables(osfm_ctx *ctx, hipStream_t st, double **d_tab, DescTable **d_dtab, const float **d_taps) {
  constexpr size_t n_tab = (size_t)kOrSide * kOrSide + 257, tab_bytes = (n_tab * sizeof(double) + 255) / 256 * 256;
  constexpr size_t dt_bytes = (sizeof(DescTable) + 255) / 256 * 256;
  const TapPlan &tp = tap_plan();
  std::lock_guard<std::mutex> g(ctx->hahog_mu);
  if (!ctx->d_hahog_tables) {
    void *p = nullptr;
    OSFM_REQUIRE(osfm_malloc_retry(ctx, &p, tab_bytes + dt_bytes + tp.taps.size() * sizeof(float)) == hipSuccess, OSFM_E_NOMEM, "osfm_hahog_extract: device memory for the tables");
    std::vector<double> h_tab(n_tab);
    {
      const int w = kOrRes;
      const double step = (2.0 * kOrExtent) / (2 * w + 1), sigma = 3;
      for (int j = -w; j <= w; ++j)
        for (int i = -w; i <= w; ++i) {
          const double dx = i * step / sigma, dy = j * step / sigma;
          h_tab[(size_t)((i + w) + (2 * w + 1) * (j + w))] = (double)(float)std::exp(-0.5 * (dx * dx + dy * dy));  // `float aaMask[]` (covdet.c:1471)
        }
      for (int k = 0; k < 257; ++k) h_tab[(size_t)kOrSide * kOrSide + k] = std::exp(-(double)k * (25.0 / 256));
    }
    const double patchStep = (double)kDExtent / kDRes;
    const double sigma_d = (double)kDExtent / (3.0 * (4 + 1) / 2) / patchStep;
    hipError_t e = hipMemcpyAsync(p, h_tab.data(), n_tab * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync((char *)p + tab_bytes + dt_bytes, tp.taps.data(), tp.taps.size() * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(desc_table_kernel, dim3(1), dim3(256), 0, st, (const double *)p + (size_t)kOrSide * kOrSide, std::sin(kPi / 2), std::cos(kPi / 2), sigma_d,
                         (DescTable *)((char *)p + tab_bytes));
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      (void)hipFree(p);
      osfm_set_error("osfm_hahog_extract: the constant tables: %s", hipGetErrorString(e));
      return OSFM_E_HIP;
    }
    ctx->d_hahog_tables = p;
  }
  *d_tab = (double *)ctx->d_hahog_tables;
  *d_dtab = (DescTable *)((char *)ctx->d_hahog_tables + tab_bytes);
  *d_taps = (const float *)((char *)ctx->d_hahog_tables + tab_bytes + dt_bytes);
  return OSFM_OK;
}
// one image on one stream (the caller holds the context lock; the block cache has its own)
static int hahog_extract_on_stream(osfm_ctx *ctx, hipStream_t st, const float *image, int rows, int cols, float peak_threshold, float edge_threshold,
                                   int target_num_features, int flags, float *points, float *desc, int capacity, int *n_features) {
  OSFM_REQUIRE(ctx && image && n_features, OSFM_E_INVALID, "osfm_hahog_extract: null argument");
  OSFM_REQUIRE(rows > 0 && cols > 0, OSFM_E_INVALID, "osfm_hahog_extract: image %d x %d", rows, cols);
  OSFM_REQUIRE(target_num_features >= 0 && capacity >= 0, OSFM_E_INVALID, "osfm_hahog_extract: negative count");
  *n_features = 0;
  if (rows < 17 || cols < 17)  // smaller than one 16-pixel octave (covdet.c:1686 gives lastOctave < firstOctave: vlfeat has no scale space
    return OSFM_OK;            // to search): a thumbnail or a masked crop yields no features instead of stopping a pipeline
  const int W0 = cols, H0 = rows;
  // vl_covdet_put_image (covdet.c:1670-1725): (minOctaveSize - 1) 2^lastOctave <= min(width, height) - 1
  const int last_octave = (int)std::floor(std::log2(std::min((double)W0 - 1, (double)H0 - 1) / 15.0));
  OSFM_REQUIRE(last_octave >= 0 && last_octave < kMaxOct, OSFM_E_UNSUPPORTED, "osfm_hahog_extract: %d octaves", last_octave + 1);
  Pyramid py;
  memset(&py, 0, sizeof(py));
  py.n_oct = last_octave + 1;
  py.base_scale = 1.6 * std::pow(2.0, 1.0 / kRes);
  Slab A, B;
  constexpr int kFeatureCap = 1 << 20;
  {
    size_t need = padded((size_t)W0 * H0, 4) + padded((size_t)W0 * H0, 1) + 5 * padded(kFeatureCap, 4) + 2 * padded(kFeatureCap, 4) +
                  padded(kFeatureCap, 8) + padded(4, 4);
    for (int o = 0; o <= last_octave; o++) need += 2 * padded((size_t)(W0 >> o) * (H0 >> o) * kLev, 4);
    OSFM_REQUIRE(A.buf.alloc(ctx, need) == hipSuccess, OSFM_E_NOMEM, "osfm_hahog_extract: %zu bytes of device memory", need);
  }
  for (int o = 0; o <= last_octave; o++) {
    py.oct[o].w = W0 >> o;
    py.oct[o].h = H0 >> o;
    const size_t n = (size_t)py.oct[o].w * py.oct[o].h * kLev;
    py.oct[o].gss = A.take<float>(n);
    py.oct[o].css = A.take<float>(n);
    for (int s = kFirstSub; s <= kLastSub; s++) py.sigma[o][s - kFirstSub] = py.base_scale * std::pow(2.0, o + (double)s / kRes);
  }
  float *d_tmp = A.take<float>((size_t)W0 * H0);
  unsigned char *d_u8 = A.take<unsigned char>((size_t)W0 * H0);  // OSFM_HAHOG_IMAGE_U8: the caller's grey levels, converted on the device
  // every Gaussian of the pyramid (they depend on the octave and the level, not on the image): made once, tap_plan / hahog_tables
  const TapPlan &tp = tap_plan();
  OSFM_REQUIRE(tp.octaves_ok > last_octave, OSFM_E_UNSUPPORTED, "osfm_hahog_extract: Gaussian wider than %d taps", kMaxTaps);
  const int *tapW = tp.W;
  const bool *first_smooth = tp.first;
  const float *d_taps = nullptr;
  double *d_tab = nullptr;  // the constant tables of the context (hahog_tables)
  DescTable *d_dtab = nullptr;
  {
    const int rct = hahog_tables(ctx, st, &d_tab, &d_dtab, &d_taps);
    if (rct != OSFM_OK) return rct;
  }
  const bool two_pass = getenv("OSFM_HAHOG_TWO_PASS") != nullptr;  // measurement / test knob: the separate column and row kernels
  // one level from the previous one; returns 1 when the level's Hessian response (css, factor) came out of the same launch
  auto fused_ok = [&](int slot, int w, int h) { return !(tapW[slot] > kSmMaxW || tapW[slot] < 1 || two_pass || w < 3 || h < 3); };
  auto smooth = [&](const float *src, float *dst, int w, int h, int slot, float *css, float factor, int src_w = 0, int src_mul = 1) -> int {
    const int W = tapW[slot];
    if (!fused_ok(slot, w, h)) {
      if (src == d_tmp) {  // a staged first level on the two-kernel path (a filter wider than the fused kernel takes): the level is the other buffer
        hipLaunchKernelGGL(conv_v_kernel, grid2(w, h), dim3(256), 0, st, src, dst, w, h, d_taps + (size_t)slot * kMaxTaps, W);
        hipLaunchKernelGGL(conv_h_kernel, grid2(w, h), dim3(256), 0, st, (const float *)dst, d_tmp, w, h, d_taps + (size_t)slot * kMaxTaps, W);
        if (hipMemcpyAsync(dst, d_tmp, (size_t)w * h * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
        return 0;
      }
      hipLaunchKernelGGL(conv_v_kernel, grid2(w, h), dim3(256), 0, st, src, d_tmp, w, h, d_taps + (size_t)slot * kMaxTaps, W);
      hipLaunchKernelGGL(conv_h_kernel, grid2(w, h), dim3(256), 0, st, (const float *)d_tmp, dst, w, h, d_taps + (size_t)slot * kMaxTaps, W);
      return 0;
    }
    float *out = src == dst ? d_tmp : dst;  // in place (the first level of an octave): through the scratch image
    constexpr int sx = kSmX - 2 * kSmHalo, sy = kSmY - 2 * kSmHalo;
    hipLaunchKernelGGL(SmoothTable<kSmMaxW>::get(W), dim3((unsigned)((w + sx - 1) / sx), (unsigned)((h + sy - 1) / sy)), dim3(256), 0, st, src, out, css, w, h,
                       d_taps + (size_t)slot * kMaxTaps, factor, src_w > 0 ? src_w : w, src_mul);
    if (out != dst && hipMemcpyAsync(dst, out, (size_t)w * h * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
    return 1;
  };
  // vl_scalespace_put_image
  // the first level of an octave is smoothed from a staging image (the caller's image / the decimated level of the octave before) straight
  // into its place; without a smoothing step (never at vlfeat's defaults) the staging image is the level
  const bool staged0 = first_smooth[0] && !two_pass;
  if (flags & OSFM_HAHOG_IMAGE_U8) {  // features.py:524: image.astype(np.float32) / 255 -- the same float32 division, a quarter of the bytes over PCIe
    const unsigned char *src = reinterpret_cast<const unsigned char *>(image);
    if (!(flags & OSFM_HAHOG_IMAGE_ON_DEVICE)) {
      OSFM_HIP(hipMemcpyAsync(d_u8, src, (size_t)W0 * H0, hipMemcpyHostToDevice, st));
      src = d_u8;
    }
    const long npx0 = (long)W0 * H0;
    hipLaunchKernelGGL(u8_to_unit_float_kernel, dim3((unsigned)((npx0 + 255) / 256)), dim3(256), 0, st, src, staged0 ? d_tmp : py.oct[0].gss, npx0);
  } else
    OSFM_HIP(hipMemcpyAsync(staged0 ? d_tmp : py.oct[0].gss, image, (size_t)W0 * H0 * sizeof(float),
                            (flags & OSFM_HAHOG_IMAGE_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  for (int o = 0; o <= last_octave; o++) {
    const Octave &oc = py.oct[o];
    const size_t npx = (size_t)oc.w * oc.h;
    const int base = o * (kLev + 1);
    const bool staged = first_smooth[o] && !two_pass;
    const double step = std::pow(2.0, o);
    auto factor_of = [&](int l) { return (float)std::pow(py.sigma[o][l] / step, 4.0); };
    bool have_css[kLev] = {};
    bool first_done = false;
    if (o > 0) {
      // copy_and_downsample of the level of the octave before.  Round 6: not a launch of its own when the first level goes through the fused
      // smoothing (which then reads every second pixel of every second row of that level) or is not smoothed at all (downsample_hessian_kernel)
      const Octave &pv = py.oct[o - 1];
      const int pl = std::min(kFirstSub + kRes, kLastSub) - kFirstSub;
      const float *plevel = pv.gss + (size_t)pl * pv.w * pv.h;
      if (first_smooth[o] && fused_ok(base, oc.w, oc.h)) {
        const int rcs = smooth(plevel, oc.gss, oc.w, oc.h, base, oc.css, factor_of(0), pv.w, 2);
        OSFM_REQUIRE(rcs >= 0, OSFM_E_HIP, "osfm_hahog_extract: device copy failed");
        have_css[0] = rcs == 1;
        first_done = true;
      } else if (!first_smooth[o] && !two_pass) {
        hipLaunchKernelGGL(downsample_hessian_kernel, grid2(oc.w, oc.h), dim3(256), 0, st, plevel, pv.w, oc.gss, oc.css, oc.w, oc.h, factor_of(0));
        have_css[0] = true;
        first_done = true;
      } else {
        hipLaunchKernelGGL(downsample_kernel, grid2(oc.w, oc.h), dim3(256), 0, st, plevel, pv.w, pv.h, staged ? d_tmp : oc.gss, oc.w, oc.h);
      }
    }
    if (first_smooth[o] && !first_done) {
      const int rcs = smooth(staged ? (const float *)d_tmp : (const float *)oc.gss, oc.gss, oc.w, oc.h, base, oc.css, factor_of(0));
      OSFM_REQUIRE(rcs >= 0, OSFM_E_HIP, "osfm_hahog_extract: device copy failed");
      have_css[0] = rcs == 1;
    }
    for (int l = 1; l < kLev; l++) {
      const int rcs = smooth(oc.gss + (size_t)(l - 1) * npx, oc.gss + (size_t)l * npx, oc.w, oc.h, base + l, oc.css + (size_t)l * npx, factor_of(l));
      OSFM_REQUIRE(rcs >= 0, OSFM_E_HIP, "osfm_hahog_extract: device copy failed");
      have_css[l] = rcs == 1;
    }
    for (int l = 0; l < kLev; l++)
      if (!have_css[l])
        hipLaunchKernelGGL(hessian_kernel, grid2(oc.w, oc.h), dim3(256), 0, st, (const float *)(oc.gss + (size_t)l * npx), oc.css + (size_t)l * npx, oc.w,
                           oc.h, factor_of(l));
  }
  // detection
  Features F;
  F.cap = kFeatureCap;
  F.x = A.take<float>((size_t)F.cap); F.y = A.take<float>((size_t)F.cap); F.sigma = A.take<float>((size_t)F.cap);
  F.peak = A.take<float>((size_t)F.cap); F.edge = A.take<float>((size_t)F.cap);
  F.o = A.take<int>((size_t)F.cap); F.s = A.take<int>((size_t)F.cap);
  F.key = A.take<unsigned long long>((size_t)F.cap);
  F.count = A.take<int>(4);
  OSFM_REQUIRE(!A.overflow, OSFM_E_NOMEM, "osfm_hahog_extract: internal: slab A too small");
  OSFM_HIP(hipMemsetAsync(F.count, 0, 4 * sizeof(int), st));
  {
    ExtremaPlan plan;
    int nwg = 0;
    for (int o = 0; o <= last_octave; o++) {
      const Octave &oc = py.oct[o];
      plan.first[o] = nwg;
      plan.gx[o] = 1;
      if (oc.w < 3 || oc.h < 3) continue;
      plan.gx[o] = (oc.w - 2 + kExCols - 1) / kExCols;
      nwg += plan.gx[o] * ((oc.h - 2 + kExRows - 1) / kExRows);
    }
    for (int o = last_octave + 1; o <= kMaxOct; o++) plan.first[o] = nwg;
    for (int o = last_octave + 1; o < kMaxOct; o++) plan.gx[o] = 1;
    if (nwg > 0)
      hipLaunchKernelGGL(extrema_kernel, dim3((unsigned)nwg), dim3(256), 0, st, py, plan, 0.8 * (double)peak_threshold, (double)peak_threshold,
                         (double)edge_threshold, F);
  }
  int n0 = 0;
  OSFM_HIP(hipMemcpyAsync(&n0, F.count, sizeof(int), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  OSFM_REQUIRE(n0 <= F.cap, OSFM_E_UNSUPPORTED, "osfm_hahog_extract: %d extrema exceed the buffer of %d", n0, F.cap);
  if (n0 == 0 || target_num_features == 0) return OSFM_OK;  // (hahog.cc:24-27 with target 0: the sorted list is cut to nothing)
  // vlfeat's order of detection, then hahog.cc's selection (stable sorts: glibc's qsort is a merge sort)
  const int nblk0 = (n0 + 255) / 256;
  size_t tb1 = 0, tb2 = 0;
  {
    int *nul = nullptr;
    unsigned long long *nulk = nullptr;
    float *nulf = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tb1, nulk, nulk, nul, nul, (size_t)n0, 0u, 64u, st);
    (void)rocprim::radix_sort_pairs_desc(nullptr, tb2, nulf, nulf, nul, nul, (size_t)n0, 0u, 32u, st);
  }
  const int n1max = std::min(n0, target_num_features), n2max = kMaxOr * n1max;
  {
    const size_t need = 4 * padded((size_t)n0, 4) + padded((size_t)n0, 8) + 2 * padded((size_t)n0, 4) + padded(std::max(tb1, tb2) + 256, 1) +
                        3 * padded((size_t)n1max, 4) + padded((size_t)n1max, 4) + padded((size_t)n1max + 1, 4) + padded((size_t)n1max * kMaxOr, 8) +
                        6 * padded((size_t)n2max, 4) + padded((size_t)4 * n2max, 4) + padded((size_t)128 * n2max, 4) + padded((size_t)n1max * sizeof(OrPlan), 1) +
                        padded((size_t)n2max * sizeof(PatchPlan), 1);
    // sizes rounded up so that images of one series reuse the cached block
    OSFM_REQUIRE(B.buf.alloc(ctx, (need + ((size_t)1 << 20)) >> 20 << 20) == hipSuccess, OSFM_E_NOMEM, "osfm_hahog_extract: %zu bytes of device memory", need);
  }
  int *d_iota = B.take<int>((size_t)n0), *d_ord = B.take<int>((size_t)n0), *d_ord2 = B.take<int>((size_t)n0), *d_sel = B.take<int>((size_t)n0);
  unsigned long long *d_keys = B.take<unsigned long long>((size_t)n0);
  float *d_sc = B.take<float>((size_t)n0), *d_sc2 = B.take<float>((size_t)n0);
  unsigned char *d_tmpsort = B.take<unsigned char>(std::max(tb1, tb2) + 256);
  OSFM_REQUIRE(!B.overflow, OSFM_E_NOMEM, "osfm_hahog_extract: internal: slab B too small");
  hipLaunchKernelGGL(iota_kernel, dim3(nblk0), dim3(256), 0, st, d_iota, n0);
  OSFM_HIP(rocprim::radix_sort_pairs(d_tmpsort, tb1, F.key, d_keys, d_iota, d_ord, (size_t)n0, 0u, 64u, st));
  int n1 = n0;
  const int *d_order = d_ord;  // feature i of the selection = detected feature d_order[i]
  if (n0 > target_num_features) {  // (hahog.cc:75-96: n0 > 3 target / 2: sort, keep that many; then n > target: sort again, keep target)
    hipLaunchKernelGGL(gather_f_kernel, dim3(nblk0), dim3(256), 0, st, (const int *)d_ord, (const float *)F.peak, d_sc, n0);
    OSFM_HIP(rocprim::radix_sort_pairs_desc(d_tmpsort, tb2, d_sc, d_sc2, d_iota, d_ord2, (size_t)n0, 0u, 32u, st));
    hipLaunchKernelGGL(compose_kernel, dim3(nblk0), dim3(256), 0, st, (const int *)d_ord2, (const int *)d_ord, d_sel, n0);
    d_order = d_sel;
    n1 = std::min(n0, target_num_features);
  }
  // selected features, gathered
  float *sx = B.take<float>((size_t)n1), *sy = B.take<float>((size_t)n1), *ssg = B.take<float>((size_t)n1);
  int *d_nor = B.take<int>((size_t)n1), *d_off = B.take<int>((size_t)n1 + 1);
  double *d_ang = B.take<double>((size_t)n1 * kMaxOr);
  OrPlan *d_orplan = (OrPlan *)B.take<double>((size_t)n1 * (sizeof(OrPlan) / 8));
  OSFM_REQUIRE(!B.overflow, OSFM_E_NOMEM, "osfm_hahog_extract: internal: slab B too small");
  const int nblk1 = (n1 + 255) / 256;
  hipLaunchKernelGGL(or_plan_kernel, dim3((unsigned)((n1 + 63) / 64)), dim3(64), 0, st, py, d_order, (const float *)F.x, (const float *)F.y,
                     (const float *)F.sigma, n1, sx, sy, ssg, d_orplan);
  hipLaunchKernelGGL(orientation_kernel, dim3(n1), dim3(256), 0, st, py, (const OrPlan *)d_orplan, n1, (const double *)d_tab, d_nor, d_ang);
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, (const int *)d_nor, n1, d_off);
  int n2 = 0;
  OSFM_HIP(hipMemcpyAsync(&n2, d_off + n1, sizeof(int), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  *n_features = n2;
  if (n2 == 0) return OSFM_OK;
  OSFM_REQUIRE(points && desc && n2 <= capacity, OSFM_E_INVALID,
               "osfm_hahog_extract: %d features do not fit the capacity of %d (4 x target_num_features always does)", n2, capacity);
  Oriented R;
  R.x = B.take<float>((size_t)n2); R.y = B.take<float>((size_t)n2);
  R.a11 = B.take<float>((size_t)n2); R.a21 = B.take<float>((size_t)n2); R.a12 = B.take<float>((size_t)n2); R.a22 = B.take<float>((size_t)n2);
  float *d_points = B.take<float>((size_t)4 * n2), *d_desc = B.take<float>((size_t)128 * n2);
  PatchPlan *d_dplan = (PatchPlan *)B.take<double>((size_t)n2 * (sizeof(PatchPlan) / 8));
  OSFM_REQUIRE(!B.overflow, OSFM_E_NOMEM, "osfm_hahog_extract: internal: slab B too small");
  hipLaunchKernelGGL(orient_frames_kernel, dim3(nblk1), dim3(256), 0, st, (const float *)sx, (const float *)sy, (const float *)ssg, (const int *)d_off,
                     (const int *)d_nor, (const double *)d_ang, n1, R);
  // hahog.cc:168-199: the descriptor's scale in patch pixels and the orientation pi / 2, with the host's libm
  hipLaunchKernelGGL(desc_plan_kernel, dim3((unsigned)((n2 + 63) / 64)), dim3(64), 0, st, py, R, n2, d_points, d_dplan);
  hipLaunchKernelGGL(descriptor_kernel, dim3(n2), dim3(256), 0, st, (const PatchPlan *)d_dplan, n2, (const DescTable *)d_dtab, flags, d_desc);
  OSFM_HIP(hipGetLastError());
  OSFM_HIP(hipMemcpyAsync(points, d_points, (size_t)4 * n2 * sizeof(float), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipMemcpyAsync(desc, d_desc, (size_t)128 * n2 * sizeof(float), hipMemcpyDeviceToHost, st));
  OSFM_HIP(hipStreamSynchronize(st));
  return OSFM_OK;
}

extern "C" int osfm_hahog_extract(osfm_ctx *ctx, const float *image, int rows, int cols, float peak_threshold, float edge_threshold,
                                  int target_num_features, int flags, float *points, float *desc, int capacity, int *n_features) {
  OSFM_REQUIRE(ctx && image && n_features, OSFM_E_INVALID, "osfm_hahog_extract: null argument");
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  return hahog_extract_on_stream(ctx, ctx->stream, image, rows, cols, peak_threshold, edge_threshold, target_num_features, flags, points, desc, capacity,
                                 n_features);
}

// Several images per call.  One image is ~130 short launches and two host round trips (the number of detections sizes the sorts, the
// number of oriented frames the descriptor launch): a single stream leaves the chip idle between them.  Here up to `concurrency` images
// are in flight, each on its own stream and host thread, so that the launches and round trips of one image run under the kernels of the
// others.  Results are those of osfm_hahog_extract image by image.
extern "C" int osfm_hahog_extract_batch(osfm_ctx *ctx, int n_images, const float *const *images, const int *rows, const int *cols, float peak_threshold,
                                        float edge_threshold, int target_num_features, int flags, float *const *points, float *const *desc,
                                        const int *capacities, int *n_features, int concurrency) {
  OSFM_REQUIRE(ctx && n_images >= 0 && (n_images == 0 || (images && rows && cols && points && desc && capacities && n_features)), OSFM_E_INVALID,
               "osfm_hahog_extract_batch: null argument");
  if (n_images == 0) return OSFM_OK;
  OSFM_CTX_LOCK(ctx);
  OSFM_HIP(hipSetDevice(ctx->device));
  const int K = std::max(1, std::min(n_images, concurrency > 0 ? std::min(concurrency, 16) : 4));
  while ((int)ctx->aux_streams.size() < K) {
    hipStream_t s = nullptr;
    OSFM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    ctx->aux_streams.push_back(s);
  }
  std::atomic<int> next{0};
  std::vector<int> rc((size_t)K, OSFM_OK);
  std::vector<std::string> err((size_t)K);
  auto work = [&](int k) {
    if (hipSetDevice(ctx->device) != hipSuccess) {
      rc[(size_t)k] = OSFM_E_HIP;
      err[(size_t)k] = "hipSetDevice failed in a worker thread";
      return;
    }
    // (worker code may only take device memory through OsfmPoolBuf::alloc(ctx, ...) -- the block cache has its own mutex -- and never
    // anything that locks the context: the caller holds ctx->mu while it waits in join())
    for (;;) {
      if (rc[(size_t)k] != OSFM_OK) break;  // before the next index is taken: an image must not be skipped by a worker that has stopped
      const int i = next.fetch_add(1);
      if (i >= n_images) break;
      const int r = hahog_extract_on_stream(ctx, ctx->aux_streams[(size_t)k], images[i], rows[i], cols[i], peak_threshold, edge_threshold,
                                            target_num_features, flags, points[i], desc[i], capacities[i], &n_features[i]);
      if (r != OSFM_OK) {
        rc[(size_t)k] = r;
        err[(size_t)k] = std::string("image ") + std::to_string(i) + ": " + osfm_last_error();  // this thread's message, for the caller's
      }
    }
  };
  std::vector<std::thread> th;
  int started = 1;
  try {
    for (int k = 1; k < K; k++) {
      th.emplace_back(work, k);
      started++;
    }
  } catch (const std::exception &) {  // no more threads: the ones that exist share the images between them
  }
  (void)started;
  work(0);
  for (auto &t : th) t.join();
  for (int k = 0; k < K; k++)
    if (rc[(size_t)k] != OSFM_OK) {
      osfm_set_error("osfm_hahog_extract_batch: %s", err[(size_t)k].c_str());
      return rc[(size_t)k];
    }
  return OSFM_OK;
}
