// rocblas.h -- host stand-in for the rocBLAS calls of the bundle-adjustment sources (test infrastructure, see ../hip/hip_runtime.h)
#pragma once
#include <hip/hip_runtime.h>

typedef struct hipemu_rocblas_handle_ *rocblas_handle;
enum rocblas_status { rocblas_status_success = 0, rocblas_status_invalid_value = 1 };
enum rocblas_operation { rocblas_operation_none = 111, rocblas_operation_transpose = 112 };
enum rocblas_fill { rocblas_fill_upper = 121, rocblas_fill_lower = 122 };
static inline rocblas_status rocblas_create_handle(rocblas_handle *h) {
  *h = (rocblas_handle)malloc(8);
  return rocblas_status_success;
}
static inline rocblas_status rocblas_destroy_handle(rocblas_handle h) {
  free(h);
  return rocblas_status_success;
}
static inline rocblas_status rocblas_set_stream(rocblas_handle, hipStream_t) { return rocblas_status_success; }
// C = alpha op(A) op(B) + beta C, column-major, `batch` problems `stride` apart
static inline rocblas_status rocblas_dgemm_strided_batched(rocblas_handle, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const double *alpha,
                                                           const double *A, int lda, long sa, const double *B, int ldb, long sb, const double *beta, double *C,
                                                           int ldc, long sc, int batch) {
  for (int q = 0; q < batch; q++) {
    const double *a = A + q * sa, *b = B + q * sb;
    double *c = C + q * sc;
    for (int j = 0; j < n; j++)
      for (int i = 0; i < m; i++) {
        double acc = 0.0;
        for (int l = 0; l < k; l++) {
          const double av = ta == rocblas_operation_none ? a[i + (long)l * lda] : a[l + (long)i * lda];
          const double bv = tb == rocblas_operation_none ? b[l + (long)j * ldb] : b[j + (long)l * ldb];
          acc += av * bv;
        }
        c[i + (long)j * ldc] = *alpha * acc + (*beta == 0.0 ? 0.0 : *beta * c[i + (long)j * ldc]);
      }
  }
  return rocblas_status_success;
}
