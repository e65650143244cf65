// Host emulation of the fundamental-matrix RANSAC kernels (opensfm_amd/csrc/ransac.hip): the same fransac_core.h functions with loops
// in place of lanes.  mode 0 = the batched path: draw kernel (16 subsets from a 192-value table), solve kernel (64 lanes side by
// side), decide kernel (lazy scoring), then the long-run kernel's rounds (32, 64, 64, ... hypotheses, 512 raw values).  mode 1 = the
// single-problem kernel of the leaf: rounds of 8, 16, 32, 64, ... hypotheses, eager scoring.
// tests/test_fransac_host.py compares F, the inlier mask and the iteration count with the CPU oracle bit for bit.
#include <cstring>
#include <vector>

#include "../../opensfm_amd/csrc/fransac_core.h"

using namespace fransac;

namespace {
struct HostEx {
  template <class F>
  void one(F f) { f(); }
  template <class F>
  void par(int n, F f) {
    for (int i = n - 1; i >= 0; --i) f(i);  // any order must do: run the lanes backwards
  }
  template <class PTS>
  void score(int nsub, const double (*models)[27], const unsigned char *nmodels, const PTS &pts, int n, float t, int (*good)[3]) {
    for (int b = 0; b < nsub; ++b)
      for (int k = 0; k < nmodels[b]; ++k) {
        int g = 0;
        for (int i = 0; i < n; ++i) {
          const Pt4 q = pts(i);
          g += epi_error(models[b] + 9 * k, (double)q.x, (double)q.y, (double)q.z, (double)q.w) <= t;
        }
        good[b][k] = g;
      }
  }
};
struct Pts {
  const Pt4 *p;
  Pt4 operator()(int k) const { return p[k]; }
};
}  // namespace

extern "C" int fransac_host_run(const double *p1, const double *p2, int n, double thr, double conf, int max_iters, int raw_cap_long,
                                int mode, double *F, uint8_t *mask, int *iters, long long *scored, int *rounds) {
  std::vector<Pt4> pv((size_t)n);
  for (int i = 0; i < n; ++i) pv[i] = Pt4{(float)p1[2 * i], (float)p1[2 * i + 1], (float)p2[2 * i], (float)p2[2 * i + 1]};
  const Pts pts{pv.data()};
  if (thr <= 0) thr = 3;
  if (conf < 2.220446049250313e-16 || conf > 1 - 2.220446049250313e-16) conf = 0.99;
  const float t = (float)(thr * thr);
  PairState st;
  state_init(st, max_iters);
  HostEx ex;
  DrawOut O;
  int nr = 0;
  bool done = false;
  int lmax = 8;
  static unsigned short subset[64][8];
  static double models[64][27];
  static unsigned char nmodels[64];
  static int good[64][3];
  static double priv[81 * 64];
  static int ipriv[9 * 64];
  if (mode == 0) {
    {  // fransac_draw_kernel
      static DrawBuf<192, 16> D;
      round_draw(ex, st, D, O, subset, pts, n, 16);
    }
    // fransac_solve_kernel: the pair's subsets sit in lanes 16 .. 31 of a wave that also holds other pairs' problems
    for (int b = O.nsub - 1; b >= 0; --b) {
      double ms1[14], ms2[14];
      subset_points(subset[b], pts, ms1, ms2);
      nmodels[b] = (unsigned char)run_7point<64>(ms1, ms2, models[b], priv + 16 + b, ipriv + 16 + b);
    }
    // fransac_decide_kernel
    done = decide_lazy(st, n, conf, O.nsub, O.fail, &models[0][0], nmodels, [&](const double *Fm) {
      int g = 0;
      for (int i = 0; i < n; ++i) g += epi_error(Fm, (double)pv[i].x, (double)pv[i].y, (double)pv[i].z, (double)pv[i].w) <= t;
      return g;
    });
    ++nr;
    lmax = 32;
  }
  while (!done) {  // fransac_rest_kernel / ransac_single_kernel
    static DrawBuf<512, 64> D;
    // raw_cap_long < 512: test knob that starves the table so that short rounds and the sequential fallback occur
    done = fransac_round<512, 64, 64>(ex, st, D, O, subset, models, nmodels, good, priv, ipriv, pts, n, t, conf, lmax,
                                      raw_cap_long > 0 ? raw_cap_long : 512);
    ++nr;
    lmax = lmax * 2 < 64 ? lmax * 2 : 64;
    if (nr > 100000) return -2;
  }
  *iters = st.iters;
  *scored = (long long)st.scored;
  *rounds = nr;
  memset(mask, 0, (size_t)n);
  if (st.max_good <= 0) return 0;
  memcpy(F, st.best, sizeof(st.best));
  for (int i = 0; i < n; ++i) mask[i] = epi_error(st.best, (double)pv[i].x, (double)pv[i].y, (double)pv[i].z, (double)pv[i].w) <= t;
  return 1;
}
