// Stage 1 of the two-stage symmetric eigensolver behind src/math/eigh.rs:1422-1528 (the reference calls LAPACK dsyevd):
// dense symmetric A (lower) -> symmetric band of half bandwidth SB by blocked Householder transformations, every
// O(n^3) piece an f64-MFMA product of k_dgemm.hip.  Per panel of SB columns (P = the nt x SB block below the band):
//   1. P = Q R by shifted CholeskyQR3 (three Gram products + Cholesky + row-parallel triangular solves; Fukaya et al.
//      2020: stable up to cond(P) ~ 1/eps), then Householder reconstruction (Ballard et al. 2014: modified LU of
//      Q - [S; 0]) turns the explicit Q into the compact-WY pair (V unit lower trapezoidal, T upper triangular) with
//      (I - V T V')' P = [S R; 0].  No column-by-column reflector chain: the panel costs ~12 launches whatever nt is.
//   2. two-sided update of the trailing matrix A22 <- Q' A22 Q:  Z = A22 V (symmetric product, lower storage),
//      M = T' (V'Z) T,  W = Z T - V M / 2,  A22 -= V W' + W V' (lower tiles only).
// V stays in A below the band (LAPACK layout, unit entry of column j at row j + SB) with tau_j = T_jj for the
// back-transformation; the band is copied out for stage 2 (k_sb2st.hip).  A panel that CholeskyQR cannot factor
// (exactly rank-deficient columns other than an all-zero panel) raises a device flag; the caller then falls back to the
// one-stage reduction (k_sytrd.hip).
#include <stdlib.h>

#include "jx_common.h"

namespace jx {

int dgemm(hipStream_t st, bool ta, bool tb, int m, int n, int k, double alpha, const double *a, int64_t lda,
          const double *b, int64_t ldb, double beta, double *c, int64_t ldc, int ksplit);
int dsymm_lower(hipStream_t st, int m, int n, double alpha, const double *a, int64_t lda, const double *b, int64_t ldb,
                double beta, double *c, int64_t ldc);
int dsyr2k_lower_nt(hipStream_t st, int m, int k, double alpha, const double *a, int64_t lda, const double *b, int64_t ldb,
                    double beta, double *c, int64_t ldc);

constexpr int SB = 64;            // half bandwidth of the intermediate band matrix
constexpr int SB_P = SB + 1;      // LDS pitch of the SB x SB work matrices
constexpr int SB_FLAG_FAIL = 0, SB_FLAG_ZERO = 1;

// ---- tiny single-workgroup kernels on SB x SB matrices (256 threads, LDS) ----------------------------------------

// in-LDS Cholesky G = R'R of the leading pw x pw block of s (upper triangle on exit, strict lower part untouched);
// indices >= pw become the identity.  Returns false (all threads) when a pivot is not positive and finite.
__device__ bool sb_chol_upper(double (*s)[SB_P], int pw, int *bad_sh) {
    const int t = threadIdx.x;
    if (t == 0) *bad_sh = 0;
    __syncthreads();
    for (int j = 0; j < pw; ++j) {
        const double d = s[j][j];
        __syncthreads();
        if (!(d > 0.0) || !(d < 1e300)) {
            if (t == 0) *bad_sh = 1;
        }
        const double r = (d > 0.0) ? sqrt(d) : 1.0;
        const double ri = 1.0 / r;
        for (int i = j + t; i < pw; i += 256) s[j][i] = (i == j) ? r : s[j][i] * ri;
        __syncthreads();
        // trailing update of the upper triangle: s[i][k] -= s[j][i] s[j][k], j < i <= k < pw
        const int rem = pw - j - 1;
        for (int e = t; e < rem * rem; e += 256) {
            const int i = j + 1 + e / rem, k = j + 1 + e % rem;
            if (k >= i) s[i][k] -= s[j][i] * s[j][k];
        }
        __syncthreads();
    }
    for (int e = t; e < SB * SB; e += 256) {
        const int i = e / SB, k = e % SB;
        if (i >= pw || k >= pw) s[i][k] = (i == k) ? 1.0 : 0.0;
        else if (k < i) s[i][k] = 0.0;
    }
    __syncthreads();
    return *bad_sh == 0;
}

// rtot <- r * rtot (both upper triangular SB x SB; rtot row-major in global memory, ld = SB), or rtot <- r; the product
// is also left in tmp (LDS)
__device__ void sb_accumulate_r(const double (*r)[SB_P], double *__restrict__ rtot, bool first, double (*tmp)[SB_P]) {
    const int t = threadIdx.x;
    for (int e = t; e < SB * SB; e += 256) tmp[e / SB][e % SB] = first ? ((e / SB == e % SB) ? 1.0 : 0.0) : rtot[e];
    __syncthreads();
    double acc[SB * SB / 256];
#pragma unroll
    for (int u = 0; u < SB * SB / 256; ++u) {
        const int e = u * 256 + t;
        const int i = e / SB, k = e % SB;
        double a = 0.0;
        if (k >= i)
            for (int q = i; q <= k; ++q) a += r[i][q] * tmp[q][k];
        acc[u] = a;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SB * SB / 256; ++u) {
        const int e = u * 256 + t;
        tmp[e / SB][e % SB] = acc[u];
        rtot[e] = acc[u];
    }
    __syncthreads();
}

// passes 1 and 2: g (pw x pw Gram matrix, column-major ld = SB) -> r_out (SB x SB row-major upper, identity-padded),
// rtot updated.  shift_coef > 0: G + shift_coef * trace(G) * I (first pass).  flags[SB_FLAG_ZERO] is set when the
// panel is exactly zero (trace == 0 in the first pass) and honoured by the later passes.
__global__ __launch_bounds__(256) void sb_chol_kernel(const double *__restrict__ g, int pw, double shift_coef, int first,
                                                      double *__restrict__ r_out, double *__restrict__ rtot,
                                                      int *__restrict__ flags, int *__restrict__ panel_zero) {
    __shared__ double s[SB][SB_P];
    __shared__ double tmp[SB][SB_P];
    __shared__ int bad;
    __shared__ int zero_sh;
    __shared__ double tr_sh;
    const int t = threadIdx.x;
    for (int e = t; e < SB * SB; e += 256) {
        const int c = e / SB, r = e % SB;          // column-major source
        s[r][c] = (r < pw && c < pw) ? g[r + c * SB] : 0.0;
    }
    __syncthreads();
    if (t == 0) {
        double tr = 0.0;
        for (int i = 0; i < pw; ++i) tr += s[i][i];
        tr_sh = tr;
        int z = *panel_zero;
        if (first) {
            z = (tr == 0.0) ? 1 : 0;
            *panel_zero = z;
        }
        zero_sh = z;
    }
    __syncthreads();
    const bool zero = (zero_sh != 0);
    if (zero) {
        for (int e = t; e < SB * SB; e += 256) s[e / SB][e % SB] = (e / SB == e % SB) ? 1.0 : 0.0;
        __syncthreads();
    } else {
        if (shift_coef > 0.0) {
            const double sh = shift_coef * tr_sh;
            for (int i = t; i < pw; i += 256) s[i][i] += sh;
            __syncthreads();
        }
        if (!sb_chol_upper(s, pw, &bad)) {
            if (t == 0) atomicOr(flags + SB_FLAG_FAIL, first ? 1 : 2);
        }
    }
    for (int e = t; e < SB * SB; e += 256) {     // the row solve multiplies by the reciprocal diagonal
        const int i = e / SB, k = e % SB;
        r_out[e] = (i == k) ? 1.0 / s[i][k] : s[i][k];
    }
    sb_accumulate_r(s, rtot, first != 0, tmp);
}

// pass 3 + Householder reconstruction (one workgroup):
//   g3 -> R3, rtot <- R3 rtot;  Qtop <- Qtop R3^-1 (the pw x pw top block of the panel, in A);  modified LU of
//   Qtop - S -> L1 (unit lower), U (upper), S = -sign(diag);  T = -U S L1^-T;  writes
//   A top block = [S rtot in the upper triangle incl. diagonal | L1 strictly below], the two V copies of the panel
//   buffer (rows 0 .. pw-1: unit lower triangle), T (column-major, ld = SB), tau = diag T, R3 and U (row-major) for the
//   row solve of the remaining panel rows.
__global__ __launch_bounds__(256) void sb_recon_kernel(const double *__restrict__ g3, int pw, double *__restrict__ atop,
                                                       int64_t lda, double *__restrict__ r3_out,
                                                       double *__restrict__ u_out, double *__restrict__ rtot,
                                                       double *__restrict__ t_out, double *__restrict__ tau,
                                                       double *__restrict__ pan_v1, double *__restrict__ pan_v2,
                                                       int64_t ldp, int *__restrict__ flags,
                                                       const int *__restrict__ panel_zero) {
    __shared__ double s[SB][SB_P];     // R3, later T
    __shared__ double w[SB][SB_P];     // Qtop -> LU
    __shared__ double tmp[SB][SB_P];
    __shared__ double sgn[SB];
    __shared__ int bad;
    const int t = threadIdx.x;
    const bool zero = (*panel_zero != 0);
    if (zero) {
        // all-zero panel: identity transformation (tau = 0, V = 0); keep the buffers the next kernels read well-defined
        for (int e = t; e < SB * SB; e += 256) {
            const int i = e / SB, k = e % SB;
            r3_out[e] = (i == k) ? 1.0 : 0.0;
            u_out[e] = (i == k) ? 1.0 : 0.0;
            t_out[e] = 0.0;
            pan_v1[i + (int64_t)k * ldp] = 0.0;      // rows < SB of the panel buffer always exist? (guarded below)
        }
        __syncthreads();
        for (int e = t; e < SB * SB; e += 256) {
            const int i = e / SB, k = e % SB;
            pan_v2[i + (int64_t)k * ldp] = 0.0;
            if (i < pw && k < pw) atop[i + (int64_t)k * lda] = 0.0;
        }
        for (int i = t; i < pw; i += 256) tau[i] = 0.0;
        return;
    }
    for (int e = t; e < SB * SB; e += 256) {
        const int c = e / SB, r = e % SB;
        s[r][c] = (r < pw && c < pw) ? g3[r + c * SB] : 0.0;
        w[r][c] = (r < pw && c < pw) ? atop[r + (int64_t)c * lda] : ((r == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (!sb_chol_upper(s, pw, &bad)) {
        if (t == 0) atomicOr(flags + SB_FLAG_FAIL, 4);
    }
    for (int e = t; e < SB * SB; e += 256) {
        const int i = e / SB, k = e % SB;
        r3_out[e] = (i == k) ? 1.0 / s[i][k] : s[i][k];
    }
    if (t == 0) {
        // G3 = Q2'Q2 must be the identity to working accuracy (R3 = I), or the panel was too ill-conditioned for
        // three CholeskyQR passes
        double worst = 0.0;
        for (int i = 0; i < pw; ++i) worst = fmax(worst, fabs(s[i][i] - 1.0));
        if (!(worst < 1e-8)) atomicOr(flags + SB_FLAG_FAIL, 8);
    }
    sb_accumulate_r(s, rtot, false, tmp);      // tmp <- R3 R2 R1
    // Qtop <- Qtop R3^-1: row i, forward over columns (thread = row)
    if (t < pw) {
        for (int c = 0; c < pw; ++c) {
            double acc = w[t][c];
            for (int k = 0; k < c; ++k) acc -= w[t][k] * s[k][c];
            w[t][c] = acc / s[c][c];
        }
    }
    __syncthreads();
    // modified LU (no pivoting; |L| <= 1 by the sign choice)
    for (int j = 0; j < pw; ++j) {
        if (t == 0) {
            const double d = w[j][j];
            const double sj = (d >= 0.0) ? -1.0 : 1.0;
            sgn[j] = sj;
            w[j][j] = d - sj;
        }
        __syncthreads();
        const double piv = w[j][j];
        for (int i = j + 1 + t; i < pw; i += 256) w[i][j] /= piv;
        __syncthreads();
        const int rem = pw - j - 1;
        for (int e = t; e < rem * rem; e += 256) {
            const int i = j + 1 + e / rem, k = j + 1 + e % rem;
            w[i][k] -= w[i][j] * w[j][k];
        }
        __syncthreads();
    }
    // A top block: S rtot (upper incl. diagonal) | L1 strictly below
    for (int e = t; e < pw * pw; e += 256) {
        const int i = e / pw, k = e % pw;
        atop[i + (int64_t)k * lda] = (k >= i) ? sgn[i] * tmp[i][k] : w[i][k];
    }
    __syncthreads();
    // T L1' = -U S  (row i of T, forward over columns; T upper triangular).  tmp <- T
    if (t < pw) {
        for (int c = 0; c < pw; ++c) {
            double acc = (c >= t) ? -w[t][c] * sgn[c] : 0.0;
            for (int k = t; k < c; ++k) acc -= tmp[t][k] * w[c][k];
            tmp[t][c] = (c >= t) ? acc : 0.0;
        }
    }
    __syncthreads();
    for (int e = t; e < SB * SB; e += 256) {
        const int i = e / SB, k = e % SB;     // (row, column)
        const bool in = (i < pw && k < pw);
        // U (row-major, identity-padded) for the row solve
        u_out[e] = in ? ((k > i) ? w[i][k] : ((k == i) ? 1.0 / w[i][k] : 0.0)) : ((i == k) ? 1.0 : 0.0);
        // T column-major
        t_out[i + k * SB] = in ? tmp[i][k] : 0.0;
        // V copies of the panel buffer, rows 0 .. SB-1 (rows >= pw of the top block belong to the row solve, which
        // writes them afterwards; columns >= pw are zero)
        if (i < pw) {
            const double v = (k < pw) ? ((i == k) ? 1.0 : ((i > k) ? w[i][k] : 0.0)) : 0.0;
            pan_v1[i + (int64_t)k * ldp] = v;
            pan_v2[i + (int64_t)k * ldp] = v;
        }
    }
    for (int i = t; i < pw; i += 256) tau[i] = tmp[i][i];
}

// M = T' N1 T  ->  TM (2 SB x SB, column-major, ld = 2 SB): rows 0 .. SB-1 = T, rows SB .. = -M / 2
__global__ __launch_bounds__(256) void sb_tm_kernel(const double *__restrict__ tmat, const double *__restrict__ n1, int pw,
                                                    double *__restrict__ tm) {
    __shared__ double ts[SB][SB_P];    // T[row][col]
    __shared__ double x[SB][SB_P];
    __shared__ double y[SB][SB_P];
    const int t = threadIdx.x;
    for (int e = t; e < SB * SB; e += 256) {
        const int c = e / SB, r = e % SB;
        ts[r][c] = tmat[r + c * SB];
        x[r][c] = (r < pw && c < pw) ? n1[r + c * SB] : 0.0;
    }
    __syncthreads();
    for (int e = t; e < SB * SB; e += 256) {          // y = N1 T
        const int i = e / SB, k = e % SB;
        double acc = 0.0;
        for (int q = 0; q <= k; ++q) acc += x[i][q] * ts[q][k];
        y[i][k] = acc;
    }
    __syncthreads();
    for (int e = t; e < SB * SB; e += 256) {          // M = T' y
        const int c = e / SB, r = e % SB;
        double acc = 0.0;
        for (int q = 0; q <= r; ++q) acc += ts[q][r] * y[q][c];
        tm[r + c * (2 * SB)] = ts[r][c];
        tm[SB + r + c * (2 * SB)] = -0.5 * acc;
    }
}

// ---- row-parallel triangular solves of the panel ------------------------------------------------------------------
// q <- q R^-1 for every panel row (thread = row; R upper triangular SB x SB, row-major, identity-padded), optionally
// followed by U^-1 (reconstruction: v = q R3^-1 U^-1) with the result also written to the two V copies of the panel
// buffer.  R entries are wave-uniform: the compiler keeps them on the scalar path.
template <bool RECON>
__global__ __launch_bounds__(128) void sb_row_solve_kernel(double *__restrict__ p, int64_t ldp, int row0, int rows,
                                                           const double *__restrict__ r, const double *__restrict__ u,
                                                           double *__restrict__ v1, double *__restrict__ v2,
                                                           int64_t ldv, int pw) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= rows) return;
    double q[SB];
    double *row = p + row0 + i;
#pragma unroll
    for (int c = 0; c < SB; ++c) q[c] = (c < pw) ? row[(int64_t)c * ldp] : 0.0;
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        double acc = q[c];
#pragma unroll
        for (int k = 0; k < c; ++k) acc -= q[k] * r[k * SB + c];
        q[c] = acc * r[c * SB + c];      // diagonal stored as its reciprocal
    }
    if (RECON) {
#pragma unroll
        for (int c = 0; c < SB; ++c) {
            double acc = q[c];
#pragma unroll
            for (int k = 0; k < c; ++k) acc -= q[k] * u[k * SB + c];
            q[c] = acc * u[c * SB + c];
        }
    }
#pragma unroll
    for (int c = 0; c < SB; ++c) {
        if (c < pw) row[(int64_t)c * ldp] = q[c];
        if (RECON) {
            const double v = (c < pw) ? q[c] : 0.0;
            v1[row0 + i + (int64_t)c * ldv] = v;
            v2[row0 + i + (int64_t)c * ldv] = v;
        }
    }
}

// band (d = 0 .. SB) of the reduced matrix -> compact storage ab (ldab x n, column-major): ab[d + j ldab] = A[j + d, j],
// zero for SB < d < ldab (room for the bulges of stage 2)
__global__ __launch_bounds__(256) void sb_extract_band_kernel(const double *__restrict__ a, int n, double *__restrict__ ab,
                                                              int ldab) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * ldab) return;
    const int d = (int)(e % ldab), j = (int)(e / ldab);
    double v = 0.0;
    if (d <= SB && j + d < n) v = a[(int64_t)j * n + j + d];
    ab[e] = v;
}

size_t sy2sb_work_doubles(int n) { return (size_t)n * 4 * SB + 8 * SB * SB + 2 * SB * 2 * SB + 64; }
int sy2sb_bandwidth() { return SB; }

// d_a (n x n, column-major, symmetric, lower referenced) -> band form in place (V below the band, see above);
// d_tau (n): tau of every stage-1 reflector (0 beyond the last eliminated column); d_ab (ldab x n): band copy for
// stage 2.  d_flags[0] != 0 on return (after the caller's synchronisation) means a panel could not be factored.
int sy2sb_lower(hipStream_t st, double *d_a, int n, double *d_tau, double *d_ab, int ldab, double *d_work,
                int *d_flags) {
    const int ncol = n - SB - 1;                     // columns with entries below the band
    JX_HIP(hipMemsetAsync(d_tau, 0, sizeof(double) * (size_t)n, st));
    JX_HIP(hipMemsetAsync(d_flags, 0, sizeof(int) * 4, st));
    if (ncol > 0) {
        double *pan = d_work;                        // (n, 4 SB), ld = n: [Z | V | W | V]
        double *p = d_work + (size_t)n * 4 * SB;
        double *g = p; p += SB * SB;                 // Gram / N1 (column-major, ld = SB)
        double *rmat = p; p += SB * SB;              // current R (row-major)
        double *umat = p; p += SB * SB;
        double *rtot = p; p += SB * SB;
        double *tmat = p; p += SB * SB;              // T of the current panel (column-major)
        double *tm = p; p += 2 * SB * SB;            // [T; -M/2]
        double *tmp1 = p; p += SB * SB;
        double *tmp2 = p; p += SB * SB;
        JX_HIP(hipMemsetAsync(pan, 0, sizeof(double) * (size_t)n * 4 * SB, st));
        const int64_t ld = n;
        const double eps = 2.220446049250313e-16;
        for (int j0 = 0; j0 < ncol; j0 += SB) {
            const int pw = (ncol - j0 < SB) ? (ncol - j0) : SB;
            const int nt = n - j0 - SB;
            double *pp = d_a + (j0 + SB) + (int64_t)j0 * ld;          // panel
            double *a22 = d_a + (j0 + SB) + (int64_t)(j0 + SB) * ld;  // trailing matrix
            double *zc = pan, *v1 = pan + (size_t)SB * ld, *wc = pan + (size_t)2 * SB * ld, *v2 = pan + (size_t)3 * SB * ld;
            int *pz = d_flags + 2;
            // CholeskyQR passes 1 and 2
            for (int pass = 0; pass < 2; ++pass) {
                if (dgemm(st, true, false, pw, pw, nt, 1.0, pp, ld, pp, ld, 0.0, g, SB, 0)) return 1;
                const double shift = (pass == 0) ? 11.0 * ((double)nt * pw + (double)pw * (pw + 1)) * eps : 0.0;
                hipLaunchKernelGGL(sb_chol_kernel, dim3(1), dim3(256), 0, st, g, pw, shift, pass == 0 ? 1 : 0, rmat, rtot,
                                   d_flags, pz);
                hipLaunchKernelGGL(sb_row_solve_kernel<false>, dim3(ceil_div(nt, 128)), dim3(128), 0, st, pp, ld, 0, nt,
                                   rmat, umat, v1, v2, ld, pw);
                JX_LAUNCH_CHECK();
            }
            // pass 3 + reconstruction
            if (dgemm(st, true, false, pw, pw, nt, 1.0, pp, ld, pp, ld, 0.0, g, SB, 0)) return 1;
            hipLaunchKernelGGL(sb_recon_kernel, dim3(1), dim3(256), 0, st, g, pw, pp, ld, rmat, umat, rtot, tmat,
                               d_tau + j0, v1, v2, ld, d_flags, pz);
            if (nt > pw)
                hipLaunchKernelGGL(sb_row_solve_kernel<true>, dim3(ceil_div(nt - pw, 128)), dim3(128), 0, st, pp, ld, pw,
                                   nt - pw, rmat, umat, v1, v2, ld, pw);
            JX_LAUNCH_CHECK();
            if (pw < SB) {
                // last, narrower panel: the columns j0 + pw .. j0 + SB - 1 of the block row see Q' from the left only
                const int nc = SB - pw;
                double *cb = d_a + (j0 + SB) + (int64_t)(j0 + pw) * ld;
                if (dgemm(st, true, false, pw, nc, nt, 1.0, v1, ld, cb, ld, 0.0, tmp1, SB, 1)) return 1;
                if (dgemm(st, true, false, pw, nc, pw, 1.0, tmat, SB, tmp1, SB, 0.0, tmp2, SB, 1)) return 1;
                if (dgemm(st, false, false, nt, nc, pw, -1.0, v1, ld, tmp2, SB, 1.0, cb, ld, 1)) return 1;
            }
            // two-sided update of the trailing matrix
            if (dsymm_lower(st, nt, pw, 1.0, a22, ld, v1, ld, 0.0, zc, ld)) return 1;
            if (dgemm(st, true, false, pw, pw, nt, 1.0, v1, ld, zc, ld, 0.0, g, SB, 0)) return 1;
            hipLaunchKernelGGL(sb_tm_kernel, dim3(1), dim3(256), 0, st, tmat, g, pw, tm);
            JX_LAUNCH_CHECK();
            if (dgemm(st, false, false, nt, pw, 2 * SB, 1.0, zc, ld, tm, 2 * SB, 0.0, wc, ld, 1)) return 1;
            if (dsyr2k_lower_nt(st, nt, 2 * SB, -1.0, v1, ld, wc, ld, 1.0, a22, ld)) return 1;
        }
    }
    hipLaunchKernelGGL(sb_extract_band_kernel, dim3((unsigned)(((int64_t)n * ldab + 255) / 256)), dim3(256), 0, st, d_a, n,
                       d_ab, ldab);
    JX_LAUNCH_CHECK();
    return 0;
}

}  // namespace jx
