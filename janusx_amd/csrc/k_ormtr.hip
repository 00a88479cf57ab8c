// Back-transformation C <- Q C of the eigenvectors of the tridiagonal matrix, Q = H_0 H_1 ... H_{n-2} from
// sytrd_lower (LAPACK dsytrd(lower) storage: v_j below the sub-diagonal of column j, tau_j) -- LAPACK dormtr
// (left, lower, no-transpose), the last stage behind src/math/eigh.rs:1422-1528.
// rocSOLVER's dormtr spends 32 ms here at n = 5000 (64-column blocks: ~80 larft recurrences of tiny kernels and
// k = 64 GEMMs).  This form uses wide blocks and BLAS-3 only: for a block of nb reflectors with explicit V
//   I - V T V',   T^-1 = strict_upper(V'V) + diag(1 / tau)            (compact WY, inverse-T form)
// so the block needs one Gram product, one triangular solve with n right-hand sides and two GEMMs with k = nb.
#include <rocblas/rocblas.h>

#include <stdlib.h>

#include "jx_common.h"

namespace jx {

constexpr int OT_NB = 1024;

// vc (rows, nbk) column-major = explicit reflectors of columns jb .. jb+nbk-1 restricted to rows jb+1 .. n-1
// (zero above the unit entry; a reflector with tau = 0 is the identity and is stored as a zero column)
// `off` = row offset of the unit entry below the diagonal: 1 for dsytrd (k_sytrd.hip), the band width for the band
// reduction (k_sy2sb.hip)
__global__ __launch_bounds__(256) void ot_extract_v_kernel(const double *__restrict__ a, int n, int jb, int nbk,
                                                          const double *__restrict__ tau, double *__restrict__ vc,
                                                          int rows, int off) {
    const int k = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (k >= nbk || r >= rows) return;
    const int j = jb + k;             // reflector / column index
    const int row = jb + off + r;     // matrix row
    double v = 0.0;
    if (tau[j] != 0.0) {
        if (row == j + off) v = 1.0;
        else if (row > j + off) v = a[(int64_t)j * n + row];
    }
    vc[(int64_t)k * rows + r] = v;
}

// m (nb x nb, column-major, ld = nb) holds G = V'V: keep the strict upper triangle, put 1/tau on the diagonal
// (1 for identity reflectors), zero the strict lower triangle.
__global__ void ot_fix_m_kernel(double *__restrict__ m, int nbk, int ld, const double *__restrict__ tau, int jb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= nbk || r >= nbk) return;
    double *p = m + (int64_t)c * ld + r;
    if (r > c) *p = 0.0;
    else if (r == c) {
        const double t = tau[jb + c];
        *p = (t != 0.0) ? 1.0 / t : 1.0;
    }
}

int ormtr_lower_off(rocblas_handle h, hipStream_t st, const double *d_a, int n, int off, int nref, const double *d_tau,
                    double *d_c, int ncols);

// d_a: (n,n) column-major after sytrd_lower; d_tau (n-1); d_c (n,n) column-major, overwritten with Q C.
int ormtr_lower(rocblas_handle h, hipStream_t st, const double *d_a, int n, const double *d_tau, double *d_c) {
    return ormtr_lower_off(h, st, d_a, n, 1, n - 1, d_tau, d_c, n);
}

// General form: reflector j (j = 0 .. nref-1) = [0 (j + off rows); 1; A(j+off+1 : n, j)] with factor d_tau[j];
// d_c (n, ncols), ld = n (ncols < n: a rank's share of the eigenvector columns).
int ormtr_lower_off(rocblas_handle h, hipStream_t st, const double *d_a, int n, int off, int nref, const double *d_tau,
                    double *d_c, int ncols) {
    if (n < 2 || nref < 1 || ncols < 1) return 0;
    // wider blocks for large n (measured at n = 20000: 512 -> 378 ms, 1024 -> 315, 2048 -> 281, 4096 -> 327)
    const int nb = (getenv("JXGPU_ORMTR_NB") && atoi(getenv("JXGPU_ORMTR_NB")) > 0) ? atoi(getenv("JXGPU_ORMTR_NB"))
                                                                                    : (n >= 12000 ? 2 * OT_NB : OT_NB);
    ScratchLease ws;   // vc (n x nb) | mm (nb x nb) | w (nb x n)
    const size_t nvc = (size_t)n * nb, nmm = (size_t)nb * nb, nw = (size_t)nb * ncols;
    if (ws.take(3, sizeof(double) * (nvc + nmm + nw))) return 1;
    double *const vc = ws.as<double>(), *const mm = vc + nvc, *const w = mm + nmm;
    const double one = 1.0, zero = 0.0, minus1 = -1.0;
    const int nblocks = (nref + nb - 1) / nb;
    for (int b = nblocks - 1; b >= 0; --b) {
        const int jb = b * nb;
        const int nbk = (nref - jb < nb) ? (nref - jb) : nb;
        const int rows = n - jb - off;
        hipLaunchKernelGGL(ot_extract_v_kernel, dim3((rows + 255) / 256, nbk), dim3(256), 0, st, d_a, n, jb, nbk, d_tau,
                           vc, rows, off);
        JX_LAUNCH_CHECK();
        rocblas_status rs = rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, nbk, nbk, rows, &one,
                                          vc, rows, vc, rows, &zero, mm, nb);
        if (rs != rocblas_status_success) return fail("ormtr: Gram dgemm failed: " + std::to_string((int)rs));
        hipLaunchKernelGGL(ot_fix_m_kernel, dim3((nbk + 63) / 64, nbk), dim3(64), 0, st, mm, nbk, nb, d_tau,
                           jb);
        JX_LAUNCH_CHECK();
        double *csub = d_c + (jb + off);              // rows jb+off .. n-1 of every column
        rs = rocblas_dgemm(h, rocblas_operation_transpose, rocblas_operation_none, nbk, ncols, rows, &one, vc,
                           rows, csub, n, &zero, w, nb);
        if (rs != rocblas_status_success) return fail("ormtr: V'C dgemm failed: " + std::to_string((int)rs));
        rs = rocblas_dtrsm(h, rocblas_side_left, rocblas_fill_upper, rocblas_operation_none, rocblas_diagonal_non_unit, nbk,
                           ncols, &one, mm, nb, w, nb);
        if (rs != rocblas_status_success) return fail("ormtr: dtrsm failed: " + std::to_string((int)rs));
        rs = rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, rows, ncols, nbk, &minus1, vc, rows,
                           w, nb, &one, csub, n);
        if (rs != rocblas_status_success) return fail("ormtr: update dgemm failed: " + std::to_string((int)rs));
    }
    JX_HIP(hipStreamSynchronize(st));   // work buffers are released on return
    return 0;
}

}  // namespace jx
