// Declarations-only stand-in for the reference's <foundation/python_types.h> (pybind11 + numpy types), so that
// opensfm/src/features/src/hahog.cc compiles from where it lies without pybind11 / OpenCV: just enough of py::tuple / py::none /
// py::gil_scoped_release / foundation::pyarray_f / py_array_from_data for that one translation unit.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <limits>
#include <utility>
#include <vector>

namespace foundation {
struct pyarray_f {
  const float *ptr = nullptr;
  long rows = 0, cols = 0;
  std::size_t size() const { return (std::size_t)(rows * cols); }
  const float *data() const { return ptr; }
  long shape(int axis) const { return axis == 0 ? rows : cols; }
};
struct array_out {
  std::vector<float> v;
  std::size_t rows = 0, cols = 0;
};
template <class T>
array_out py_array_from_data(const T *data, std::size_t rows, std::size_t cols) {
  array_out a;
  a.rows = rows;
  a.cols = cols;
  a.v.assign(data, data + rows * cols);
  return a;
}
}  // namespace foundation

namespace py {
struct none {};
struct gil_scoped_release {};
struct tuple {
  bool is_none = false;
  foundation::array_out first, second;
  tuple() = default;
  tuple(none) : is_none(true) {}
};
inline tuple make_tuple(foundation::array_out a, foundation::array_out b) {
  tuple t;
  t.first = std::move(a);
  t.second = std::move(b);
  return t;
}
}  // namespace py
