// rocsolver.h -- host stand-in (test infrastructure, see ../hip/hip_runtime.h)
#pragma once
#include <rocblas/rocblas.h>

#include <cmath>
static inline rocblas_status rocsolver_dpotrf(rocblas_handle, rocblas_fill, int n, double *A, int lda, int *info) {
  *info = 0;
  for (int j = 0; j < n; j++) {
    double d = A[j + (long)j * lda];
    for (int k = 0; k < j; k++) d -= A[j + (long)k * lda] * A[j + (long)k * lda];
    if (!(d > 0)) {
      *info = j + 1;
      return rocblas_status_success;
    }
    d = std::sqrt(d);
    A[j + (long)j * lda] = d;
    for (int i = j + 1; i < n; i++) {
      double v = A[i + (long)j * lda];
      for (int k = 0; k < j; k++) v -= A[i + (long)k * lda] * A[j + (long)k * lda];
      A[i + (long)j * lda] = v / d;
    }
  }
  return rocblas_status_success;
}
static inline rocblas_status rocsolver_dpotrs(rocblas_handle, rocblas_fill, int n, int nrhs, const double *A, int lda, double *B, int ldb) {
  for (int q = 0; q < nrhs; q++) {
    double *b = B + (long)q * ldb;
    for (int i = 0; i < n; i++) {
      double v = b[i];
      for (int k = 0; k < i; k++) v -= A[i + (long)k * lda] * b[k];
      b[i] = v / A[i + (long)i * lda];
    }
    for (int i = n - 1; i >= 0; i--) {
      double v = b[i];
      for (int k = i + 1; k < n; k++) v -= A[k + (long)i * lda] * b[k];
      b[i] = v / A[i + (long)i * lda];
    }
  }
  return rocblas_status_success;
}
