/* Stage-level views of the reference's HAHOG pipeline (vlfeat's covdet, compiled from /root/reference/opensfm/src/third_party/vlfeat
 * where it lies): one Gaussian scale-space level / its Hessian response, and the detected features before selection and orientation.
 * TEST INFRASTRUCTURE ONLY (linked into oracle/_ref/libhahog_ref.so); used to localise a difference while the GPU extractor is brought
 * up -- the row's parity statement itself is against features::hahog (hahog_ref.cc). */
#include <string.h>

#include <vl/covdet.h>
#include <vl/scalespace.h>

static VlCovDet *run(const float *image, long rows, long cols, double peak, double edge) {
  VlCovDet *covdet = vl_covdet_new(VL_COVDET_METHOD_HESSIAN);
  vl_covdet_set_first_octave(covdet, 0);
  vl_covdet_set_peak_threshold(covdet, peak);
  vl_covdet_set_edge_threshold(covdet, edge);
  vl_covdet_put_image(covdet, image, cols, rows);
  vl_covdet_set_non_extrema_suppression_threshold(covdet, 0);
  vl_covdet_detect(covdet, ((vl_size)1) << 40);  /* (vl_size is signed in this vlfeat: -1 would stop after the first octave) */
  return covdet;
}

/* geometry: out[0] = last octave, out[1] = first subdivision, out[2] = last subdivision */
long covdet_ref_geometry(long rows, long cols, long *out) {
  float z = 0.f;
  (void)z;
  VlCovDet *covdet = vl_covdet_new(VL_COVDET_METHOD_HESSIAN);
  vl_covdet_set_first_octave(covdet, 0);
  float *tmp = (float *)vl_calloc((size_t)rows * cols, sizeof(float));
  vl_covdet_put_image(covdet, tmp, cols, rows);
  VlScaleSpaceGeometry g = vl_scalespace_get_geometry(vl_covdet_get_gss(covdet));
  out[0] = g.lastOctave;
  out[1] = g.octaveFirstSubdivision;
  out[2] = g.octaveLastSubdivision;
  vl_free(tmp);
  vl_covdet_delete(covdet);
  return 0;
}

/* level (o, s) of the Gaussian scale space (gss) and of the Hessian response (css): (rows >> o) x (cols >> o) floats each */
long covdet_ref_level(const float *image, long rows, long cols, long o, long s, float *gss, float *css) {
  VlCovDet *covdet = run(image, rows, cols, 1e-5, 10.0);
  VlScaleSpaceOctaveGeometry og = vl_scalespace_get_octave_geometry(vl_covdet_get_gss(covdet), o);
  const size_t n = (size_t)og.width * og.height;
  memcpy(gss, vl_scalespace_get_level(vl_covdet_get_gss(covdet), o, s), n * sizeof(float));
  memcpy(css, vl_scalespace_get_level(vl_covdet_get_css(covdet), o, s), n * sizeof(float));
  vl_covdet_delete(covdet);
  return (long)n;
}

/* detected features in vlfeat's order: out[7 i ..] = x, y, sigma, peakScore, edgeScore, o, s */
long covdet_ref_detect(const float *image, long rows, long cols, double peak, double edge, float *out, long capacity) {
  VlCovDet *covdet = run(image, rows, cols, peak, edge);
  const long n = (long)vl_covdet_get_num_features(covdet);
  const VlCovDetFeature *f = (const VlCovDetFeature *)vl_covdet_get_features(covdet);
  for (long i = 0; i < n && i < capacity; i++) {
    out[7 * i + 0] = f[i].frame.x;
    out[7 * i + 1] = f[i].frame.y;
    out[7 * i + 2] = f[i].frame.a11;
    out[7 * i + 3] = f[i].peakScore;
    out[7 * i + 4] = f[i].edgeScore;
    out[7 * i + 5] = (float)f[i].o;
    out[7 * i + 6] = (float)f[i].s;
  }
  vl_covdet_delete(covdet);
  return n;
}

/* orientations of the frame (x, y, sigma I): angles[4], scores[4]; returns their number */
long covdet_ref_orientations(const float *image, long rows, long cols, const float *xys, long nframes, double *angles, double *scores, long *counts) {
  VlCovDet *covdet = vl_covdet_new(VL_COVDET_METHOD_HESSIAN);
  vl_covdet_set_first_octave(covdet, 0);
  vl_covdet_put_image(covdet, image, cols, rows);
  for (long i = 0; i < nframes; i++) {
    VlFrameOrientedEllipse fr;
    fr.x = xys[3 * i];
    fr.y = xys[3 * i + 1];
    fr.a11 = xys[3 * i + 2];
    fr.a12 = 0;
    fr.a21 = 0;
    fr.a22 = xys[3 * i + 2];
    vl_size n = 0;
    VlCovDetFeatureOrientation *o = vl_covdet_extract_orientations_for_frame(covdet, &n, fr);
    counts[i] = (long)n;
    for (vl_size j = 0; j < n; j++) {
      angles[4 * i + j] = o[j].angle;
      scores[4 * i + j] = o[j].score;
    }
  }
  vl_covdet_delete(covdet);
  return nframes;
}
