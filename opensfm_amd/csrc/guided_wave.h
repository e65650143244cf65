// guided_wave.h -- masked (guided) brute-force matching: per-lane numerics + wavefront orchestration.
//
// reference: match_brute_force[_symmetric](f1, f2, config, maskij) (opensfm/matching.py:723-777: cv2 knnMatch(k = 2, masks) + Lowe
// ratio, both directions with the transposed mask + set intersection) and the mask of guided matching,
// compute_inliers_bearing_epipolar (matching.py:847-868) = geometry::EpipolarAngleTwoBearingsMany
// (opensfm/src/geometry/src/triangulation.cc:195-219).
//
// First-correct design (not the hot path: guided matching only runs when poses are known): one wavefront per QUERY descriptor,
// lanes stride over the train descriptors; integer squared distances from u8 dot products (norm_a + norm_b - 2 a.b, exact), the
// float32 distance sqrtf(d2) and the top-2 insertion of OpenCV's batchDistance per lane, partial top-2 merged by lane 0.
// Written against the wave policy of relpose_wave.h (single / parallel_for) so that a host emulation runs the same code
// (tests/native/guided_host.cpp) bit for bit against the CPU oracle.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define OSFM_GHD __host__ __device__ inline
#else
#define OSFM_GHD inline
#endif

namespace osfm_gm {

constexpr int kDim = 128;
constexpr int kLanes = 64;

// ---- epipolar geometry (doubles on float32-rounded bearings, as the reference) --------------------------------------
OSFM_GHD void normalized3(double* v) {  // Eigen::normalized(): left unchanged when the norm is 0
  const double z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (z > 0) {
    const double n = sqrt(z);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
  }
}
OSFM_GHD void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
// per feature of the FIRST image: [x (3), e1 = (t^ x x)^ (3)]; of the SECOND image: [w = R y (3), e2 = (t^ x w)^ (3)]
OSFM_GHD void epipolar_precompute(int second, const float* b, const double* R, const double* tn, double* out6) {
  double v[3] = {(double)b[0], (double)b[1], (double)b[2]};
  if (second) {
    const double y[3] = {v[0], v[1], v[2]};
    for (int a = 0; a < 3; a++) v[a] = y[0] * R[3 * a] + y[1] * R[3 * a + 1] + y[2] * R[3 * a + 2];
  }
  double e[3];
  cross3(tn, v, e);
  normalized3(e);
  for (int a = 0; a < 3; a++) {
    out6[a] = v[a];
    out6[3 + a] = e[a];
  }
}
// first6 = [x, e1] of feature i of image 1, second6 = [w, e2] of feature j of image 2
OSFM_GHD double epipolar_angle(const double* first6, const double* second6) {
  const double a = fabs(first6[3] * second6[0] + first6[4] * second6[1] + first6[5] * second6[2]);
  const double b = fabs(first6[0] * second6[3] + first6[1] * second6[4] + first6[2] * second6[5]);
  return M_PI / 2.0 - acos((a + b) / 2.0);
}

// which (query, train) combinations may be ranked
struct Allowed {
  const uint8_t* mask;  // explicit maskij (n1 x n2, row = feature of image 1) or nullptr
  const double* first6;   // n1 x 6 (when mask == nullptr)
  const double* second6;  // n2 x 6
  double threshold;
  int n2;
  OSFM_GHD bool operator()(int i1, int i2) const {  // indices in image 1 / image 2, whatever the matching direction
    if (mask) return mask[(size_t)i1 * n2 + i2] != 0;
    return epipolar_angle(first6 + 6 * (size_t)i1, second6 + 6 * (size_t)i2) < threshold;
  }
};

// ---- distances ------------------------------------------------------------------------------------------------------
OSFM_GHD int dot_u8_128(const uint8_t* a, const uint8_t* b) {
  int acc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t* pa = (const uint32_t*)a;
  const uint32_t* pb = (const uint32_t*)b;
  for (int k = 0; k < kDim / 4; k++) acc = (int)__builtin_amdgcn_udot4(pa[k], pb[k], (uint32_t)acc, false);
#else
  for (int k = 0; k < kDim; k++) acc += (int)a[k] * (int)b[k];
#endif
  return acc;
}
struct Top2 {
  float d0, d1;  // the two smallest float32 distances seen (d1 = INFINITY until a second candidate arrives)
  int j0, n;     // index of d0 (lowest index among equal distances), number of candidates seen (saturates at 2)
};
OSFM_GHD Top2 top2_empty() { return Top2{INFINITY, INFINITY, -1, 0}; }
// batchDistance's insertion (K = 2): `if (d < dist[1])`, shift while `dist[k] > d`
OSFM_GHD void top2_insert(Top2& t, float d, int j) {
  if (t.n < 2) t.n++;
  if (d < t.d1) {
    if (t.d0 > d) {
      t.d1 = t.d0;
      t.d0 = d;
      t.j0 = j;
    } else {
      t.d1 = d;
    }
  }
}
// merge of two partial results over DISJOINT candidate sets; `b` holds larger indices than `a` only if its j0 is larger,
// so ties on d0 resolve to the lower index explicitly
OSFM_GHD Top2 top2_merge(const Top2& a, const Top2& b) {
  Top2 r;
  r.n = a.n + b.n > 2 ? 2 : a.n + b.n;
  const bool a_first = a.d0 < b.d0 || (a.d0 == b.d0 && (b.j0 < 0 || (a.j0 >= 0 && a.j0 < b.j0)));
  const Top2& f = a_first ? a : b;
  const Top2& s = a_first ? b : a;
  r.d0 = f.d0;
  r.j0 = f.j0;
  r.d1 = f.d1 < s.d0 ? f.d1 : s.d0;  // second smallest of the union
  return r;
}

struct GuidedShared {
  Top2 part[kLanes];
};

// One query against all train descriptors.  desc: u8, 128 per row; norms: sum of squares per row.
// `first_is_query`: the query belongs to image 1 (direction i -> j), otherwise to image 2 (the transposed mask).
// Returns the matched train index or -1 (fewer than two allowed candidates, or Lowe's ratio fails).
template <class W>
OSFM_GHD int match_query_wave(W& w, GuidedShared& s, const uint8_t* q, int qnorm, int qi, const uint8_t* train, const int* tnorm,
                              int n_train, const Allowed& allowed, int first_is_query, double ratio) {
  w.parallel_for(kLanes, [&](int l) { s.part[l] = top2_empty(); });
  w.parallel_for(n_train, [&](int j) {
    if (!(first_is_query ? allowed(qi, j) : allowed(j, qi))) return;
    const int d2 = qnorm + tnorm[j] - 2 * dot_u8_128(q, train + (size_t)j * kDim);
    top2_insert(s.part[j % kLanes], sqrtf((float)d2), j);
  });
  Top2 t = s.part[0];  // every lane merges redundantly (uniform data): no broadcast needed
  for (int l = 1; l < kLanes; l++) t = top2_merge(t, s.part[l]);
  if (t.n < 2) return -1;
  return ((double)t.d0 < ratio * (double)t.d1) ? t.j0 : -1;
}

}  // namespace osfm_gm
