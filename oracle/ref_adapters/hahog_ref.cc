// C entry point around the reference's own features::hahog (opensfm/src/features/src/hahog.cc, compiled from where it lies together
// with the vendored vlfeat sources covdet.c / sift.c / scalespace.c / imopv.c ...): the oracle of the HAHOG extraction row
// (SURVEY.md 8f-4).  TEST INFRASTRUCTURE ONLY; built into oracle/_ref/libhahog_ref.so by `make -C oracle ref`.
#include <cstring>

#include <features/hahog.h>

// image: rows x cols float32 in [0, 1].  Returns the number of features (-1: the reference returned None); the first call with
// points == nullptr only counts.  points: n x 4 (x, y, size, angle in degrees), desc: n x 128.
extern "C" long hahog_ref(const float *image, long rows, long cols, float peak_threshold, float edge_threshold, int target_num_features,
                          float *points, float *desc, long capacity) {
  foundation::pyarray_f im;
  im.ptr = image;
  im.rows = rows;
  im.cols = cols;
  py::tuple t = features::hahog(im, peak_threshold, edge_threshold, target_num_features);
  if (t.is_none) return -1;
  const long n = (long)t.first.rows;
  if (points && desc && n <= capacity) {
    std::memcpy(points, t.first.v.data(), sizeof(float) * 4 * (size_t)n);
    std::memcpy(desc, t.second.v.data(), sizeof(float) * 128 * (size_t)n);
  }
  return n;
}
