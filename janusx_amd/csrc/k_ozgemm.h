// Sliced int8 GEMM (k_ozgemm.hip): images and products.
#pragma once
#include "jx_common.h"

namespace jx {

struct OzImage {
    int8_t *q = nullptr;      // nrb * nks * planes * 4096 bytes
    double *scale = nullptr;  // nrb * 128 row maxima (0 rows -> 1.0 at use)
    int rows = 0, k = 0, nrb = 0, nks = 0, planes = 0;
};

// digit planes per operand: JXGPU_OZ_PLANES (4 .. 6), default 6 = 21 int8 products, ~4e-14 relative (5: 15 products, 1e-11:
// the eigenvectors then leave the 1e-10 orthogonality bar after the ~25 products in a row of one decomposition)
int oz_planes();
// the calling thread's override (0: none) -- handed on to worker threads of a decomposition
int oz_planes_override_get();
void oz_planes_override_set(int planes);
size_t oz_image_bytes(int rows, int k, int planes = 0);
OzImage oz_image_at(void *mem, int rows, int k, int planes = 0);
// operand element (r, k) = x[r * rs + k * cs], rs == 1 or cs == 1
int oz_slice(hipStream_t st, const double *x, int64_t rs, int64_t cs, const OzImage &im);
// C (m x n, column-major, ldc) = alpha A B' + beta C from images (a: >= m rows, b: >= n rows, same k).
// mode 0: all tiles; 1: tiles with column tile >= row tile only; 2: A[r][k] = 0 for k < 128 (r / 128)
int oz_mm(hipStream_t st, const OzImage &a, const OzImage &b, int m, int n, double alpha, double beta, double *c, int64_t ldc,
          int mode);
int oz_dgemm(hipStream_t st, bool ta, bool tb, int m, int n, int k, double alpha, const double *a, int64_t lda, const double *b,
             int64_t ldb, double beta, double *c, int64_t ldc, void *work, size_t work_bytes);

}  // namespace jx
