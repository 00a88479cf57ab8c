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

#include <algorithm>

#include "jx_common.h"

namespace jx {

int dgemm(hipStream_t st, bool ta, bool tb, int m, int n, int k, double alpha, const double *a, int64_t lda,
          const double *b, int64_t ldb, double beta, double *c, int64_t ldc, int ksplit, double *ws, size_t ws_doubles);
int dsymm_lower(hipStream_t st, int m, int n, double alpha, const double *a, int64_t lda, const double *b, int64_t ldb,
                double beta, double *c, int64_t ldc, double *ws, size_t ws_doubles);
int dsyr2k_lower_nt(hipStream_t st, int m, int k, double alpha, const double *a, int64_t lda, const double *b, int64_t ldb,
                    double beta, double *c, int64_t ldc);

constexpr int SB = 64;            // half bandwidth of the intermediate band matrix
constexpr int SB_P = SB + 1;      // LDS pitch of the SB x SB work matrices
constexpr int SB_FLAG_FAIL = 0;

// ---- tiny single-workgroup kernels on SB x SB matrices ------------------------------------------------------------
// 256 threads hold a 64 x 64 matrix in registers, 4 x 4 elements each, cyclically: thread (ti, tk) = (t / 16, t % 16)
// owns the elements (ti + 16 a, tk + 16 b).  The sequential eliminations (Cholesky, LU, triangular solves) are fully
// unrolled right-looking loops: per step the owners of the pivot row / column publish it through a double-buffered LDS
// vector, ONE barrier, and every thread updates its 16 elements from 8 LDS reads.
static_assert(SB == 64, "the register-cyclic tiny kernels are written for SB = 64");

struct Tiny {
    double v[4][4];
};

__device__ __forceinline__ void tiny_load_lds(Tiny &m, const double (*s)[SB_P]) {
    const int ti = threadIdx.x >> 4, tk = threadIdx.x & 15;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) m.v[a][b] = s[ti + 16 * a][tk + 16 * b];
}
__device__ __forceinline__ void tiny_store_lds(const Tiny &m, double (*s)[SB_P]) {
    const int ti = threadIdx.x >> 4, tk = threadIdx.x & 15;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s[ti + 16 * a][tk + 16 * b] = m.v[a][b];
}

// The elimination loops run over j at run time (a fully unrolled form is ~0.5 MB of code and runs at the speed of the
// instruction fetch).  Register arrays are only ever indexed with compile-time constants: the owners of the pivot row /
// column publish all four of their 16-blocks and the readers pick block j / 16 by its LDS address.
typedef double TinyVec[4][SB];      // [16-block][position]

// G (symmetric, full, identity-padded) -> R upper triangular (G = R'R), strict lower part zero.  Returns 0 when a pivot
// is not positive and finite (the factor is then meaningless; the caller raises the failure flag).
// The step loops are split by the 16-block of the pivot (compile time) x its position in the block (run time): the blocks above /
// left of the pivot's are finished, so a step of block ja touches only the (4 - ja)^2 register blocks a, b >= ja, the row /
// column comparisons survive only in the blocks a == ja or b == ja, and the pivot owners publish m.v[ja][..] directly.  Same
// operations on the same values as the plain 64-step form (the skipped ones were predicated off or subtracted exact zeros).
__device__ __forceinline__ int tiny_chol_upper(Tiny &m, TinyVec *vec) {
    const int ti = threadIdx.x >> 4, tk = threadIdx.x & 15;
    int ok = 1;
#pragma unroll
    for (int ja = 0; ja < 4; ++ja) {
#pragma unroll 1
        for (int jr = 0; jr < 16; ++jr) {
            const int j = 16 * ja + jr, p = j & 1;
            if (ti == jr) {
#pragma unroll
                for (int b = 0; b < 4; ++b) vec[p][ja][tk + 16 * b] = m.v[ja][b];    // row j = row jr of block ja
            }
            __syncthreads();
            const double *rowj = vec[p][ja];
            double d = rowj[j];
            if (!(d > 0.0) || !(d < 1e300)) {
                ok = 0;
                d = 1.0;
            }
            const double ri = 1.0 / sqrt(d);
            double rk[4], rr[4];
#pragma unroll
            for (int b = ja; b < 4; ++b) rk[b] = rowj[tk + 16 * b] * ri;
#pragma unroll
            for (int a = ja; a < 4; ++a) rr[a] = rowj[ti + 16 * a] * ri;
#pragma unroll
            for (int a = ja; a < 4; ++a) {
#pragma unroll
                for (int b = ja; b < 4; ++b) {
                    const bool i_gt = (a > ja) || ti > jr, i_eq = (a == ja) && ti == jr;
                    const bool k_gt = (b > ja) || tk > jr, k_ge = (b > ja) || tk >= jr;
                    if (i_gt && k_gt) m.v[a][b] -= rr[a] * rk[b];
                    else if (i_eq) m.v[a][b] = k_ge ? rk[b] : 0.0;
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (ti + 16 * a > tk + 16 * b) m.v[a][b] = 0.0;
    return ok;
}

// X <- X U^-1, U upper triangular in LDS (u[c][k], c <= k) with its diagonal stored as the reciprocal.
// UPPER_X: X is upper triangular itself (an inverse or a product of upper triangular factors built from the identity): column c
// is zero below row c, so only the row blocks a <= cb take part (half the multiply-adds; the skipped entries stay exact zeros).
template <bool UPPER_X = false>
__device__ __forceinline__ void tiny_trsm_right_upper(Tiny &x, const double (*u)[SB_P], TinyVec *vec) {
    const int ti = threadIdx.x >> 4, tk = threadIdx.x & 15;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        constexpr int A_ALL = 4;
        const int na = UPPER_X ? cb + 1 : A_ALL;             // compile time inside the unrolled loop
#pragma unroll 1
        for (int cr = 0; cr < 16; ++cr) {
            const int c = 16 * cb + cr, p = c & 1;
            if (tk == cr) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (a < na) vec[p][cb][ti + 16 * a] = x.v[a][cb];                // column c = column cr of block cb
            }
            __syncthreads();
            const double *colc = vec[p][cb];
            const double rd = u[c][c];
            double xc[4], uk[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < na) xc[a] = colc[ti + 16 * a] * rd;
#pragma unroll
            for (int b = cb; b < 4; ++b) uk[b] = ((b > cb) || tk > cr) ? u[c][tk + 16 * b] : 0.0;
#pragma unroll
            for (int b = cb; b < 4; ++b) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < na) {
                        if (b == cb && tk == cr) x.v[a][b] = xc[a];
                        else x.v[a][b] -= xc[a] * uk[b];
                    }
                }
            }
        }
    }
}

// modified LU of W (Householder reconstruction): on exit the strict lower part holds L (unit diagonal implied), the
// upper part incl. diagonal U; sgn[j] = the sign subtracted from the j-th pivot
__device__ __forceinline__ void tiny_lu_modified(Tiny &w, TinyVec *vec, TinyVec *vec2, double *sgn) {
    const int ti = threadIdx.x >> 4, tk = threadIdx.x & 15;
#pragma unroll
    for (int ja = 0; ja < 4; ++ja) {
#pragma unroll 1
        for (int jr = 0; jr < 16; ++jr) {
            const int j = 16 * ja + jr, p = j & 1;
            if (ti == jr) {
#pragma unroll
                for (int b = 0; b < 4; ++b) vec[p][ja][tk + 16 * b] = w.v[ja][b];     // row j
            }
            if (tk == jr) {
#pragma unroll
                for (int a = 0; a < 4; ++a) vec2[p][ja][ti + 16 * a] = w.v[a][ja];    // column j
            }
            __syncthreads();
            const double *rowj = vec[p][ja], *colj = vec2[p][ja];
            const double d = rowj[j];
            const double sj = (d >= 0.0) ? -1.0 : 1.0;
            const double piv = d - sj;
            const double pinv = 1.0 / piv;
            double li[4], uk[4];
#pragma unroll
            for (int a = ja; a < 4; ++a) li[a] = ((a > ja) || ti > jr) ? colj[ti + 16 * a] * pinv : 0.0;
#pragma unroll
            for (int b = ja; b < 4; ++b) uk[b] = ((b > ja) || tk > jr) ? rowj[tk + 16 * b] : 0.0;
#pragma unroll
            for (int a = ja; a < 4; ++a) {
#pragma unroll
                for (int b = ja; b < 4; ++b) {
                    const bool k_eq = (b == ja) && tk == jr;
                    const bool i_gt = (a > ja) || ti > jr, i_eq = (a == ja) && ti == jr;
                    if (k_eq && i_gt) w.v[a][b] = li[a];
                    else if (k_eq && i_eq) w.v[a][b] = piv;
                    else w.v[a][b] -= li[a] * uk[b];
                }
            }
            if (threadIdx.x == 0) sgn[j] = sj;
        }
    }
    __syncthreads();
}

// C = A B (64 x 64 each; A, B in LDS)
__device__ __forceinline__ void tiny_matmul(Tiny &c, const double (*a)[SB_P], const double (*b)[SB_P]) {
    const int ti = threadIdx.x >> 4, tk = threadIdx.x & 15;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) c.v[x][y] = 0.0;
#pragma unroll 8
    for (int q = 0; q < SB; ++q) {
        double ai[4], bk[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) ai[x] = a[ti + 16 * x][q];
#pragma unroll
        for (int y = 0; y < 4; ++y) bk[y] = b[q][tk + 16 * y];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) c.v[x][y] += ai[x] * bk[y];
    }
}

// shared by the tiny kernels: global (column-major, ld = SB) pw x pw block -> LDS row-major, identity- or zero-padded
__device__ __forceinline__ void tiny_fetch_colmajor(double (*s)[SB_P], const double *__restrict__ g, int pw, bool ident_pad) {
    for (int e = threadIdx.x; e < SB * SB; e += 256) {
        const int c = e / SB, r = e % SB;
        s[r][c] = (r < pw && c < pw) ? g[r + c * SB] : ((ident_pad && r == c) ? 1.0 : 0.0);
    }
}

// passes 1 and 2: g (pw x pw Gram matrix, column-major ld = SB) -> r_out = R^-1 (SB x SB column-major, identity-padded;
// G = R'R), rtot <- R rtot (or R in the first pass).  shift_coef > 0:
// G + shift_coef * trace(G) * I.  *panel_zero is set in the first pass when the panel is exactly zero (trace == 0) and
// honoured by the later passes.
__global__ __launch_bounds__(256) void sb_chol_kernel(const double *__restrict__ g, int pw, double shift_coef, int first,
                                                      double *__restrict__ r_out, double *__restrict__ rtot,
                                                      int *__restrict__ flags, int *__restrict__ panel_zero) {
    __shared__ double s[SB][SB_P];
    __shared__ double s2[SB][SB_P];
    __shared__ TinyVec vec[2];
    __shared__ int zero_sh;
    __shared__ double tr_sh;
    const int t = threadIdx.x;
    tiny_fetch_colmajor(s, g, pw, true);
    if (!first) {
        for (int e = t; e < SB * SB; e += 256) s2[e / SB][e % SB] = rtot[e];
    }
    __syncthreads();
    if (t == 0) {
        double tr = 0.0;
        for (int i = 0; i < pw; ++i) tr += s[i][i];
        int z = *panel_zero;
        if (first) {
            z = (tr == 0.0) ? 1 : 0;
            *panel_zero = z;
        }
        zero_sh = z;
        tr_sh = tr;
    }
    __syncthreads();
    const bool zero = (zero_sh != 0);
    const double tr = tr_sh;
    Tiny m;
    tiny_load_lds(m, s);
    const int ti = t >> 4, tk = t & 15;
    if (zero) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) m.v[a][b] = (ti + 16 * a == tk + 16 * b) ? 1.0 : 0.0;
    } else {
        if (shift_coef > 0.0) {
            const double sh = shift_coef * tr;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (ti + 16 * a == tk + 16 * b && ti + 16 * a < pw) m.v[a][b] += sh;
        }
        if (!tiny_chol_upper(m, vec)) atomicOr(flags + SB_FLAG_FAIL, first ? 1 : 2);
    }
    __syncthreads();
    tiny_store_lds(m, s);                       // R
    __syncthreads();
    if (first) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) rtot[(ti + 16 * a) * SB + tk + 16 * b] = m.v[a][b];
    } else {
        Tiny prod;
        tiny_matmul(prod, s, s2);               // R * rtot
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) rtot[(ti + 16 * a) * SB + tk + 16 * b] = prod.v[a][b];
    }
    // R^-1 for the panel product Q = P R^-1 (an f64-MFMA GEMM): X = I, X <- X R^-1
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (ti + 16 * a == tk + 16 * b) s[ti + 16 * a][tk + 16 * b] = 1.0 / m.v[a][b];
    Tiny x;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) x.v[a][b] = (ti + 16 * a == tk + 16 * b) ? 1.0 : 0.0;
    __syncthreads();
    tiny_trsm_right_upper<true>(x, s, vec);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r_out[(ti + 16 * a) + (tk + 16 * b) * SB] = x.v[a][b];      // column-major
}

// pass 3 + Householder reconstruction (one workgroup):
//   g3 -> R3, rtot <- R3 rtot;  Qtop <- Qtop R3^-1 (the pw x pw top block of the panel, in A);  modified LU of
//   Qtop - S -> L1 (unit lower), U (upper), S = -sign(diag);  T = -U S L1^-T;  writes
//   A top block = [S rtot in the upper triangle incl. diagonal | L1 strictly below], the two V copies of the panel
//   buffer (rows 0 .. pw-1: unit lower triangle), T (column-major, ld = SB), tau = diag T, and u_out = R3^-1 U^-1
//   (column-major) for the remaining panel rows: v = q R3^-1 U^-1 is one GEMM.
__global__ __launch_bounds__(256) void sb_recon_kernel(const double *__restrict__ g3, int pw, double *__restrict__ atop,
                                                       int64_t lda, double *__restrict__ r3_out,
                                                       double *__restrict__ u_out, double *__restrict__ rtot,
                                                       double *__restrict__ t_out, double *__restrict__ tau,
                                                       double *__restrict__ pan_v1, double *__restrict__ pan_v2,
                                                       int64_t ldp, int *__restrict__ flags,
                                                       const int *__restrict__ panel_zero, double skip_tol) {
    __shared__ double s[SB][SB_P];
    __shared__ double s2[SB][SB_P];
    __shared__ TinyVec vec[2];
    __shared__ TinyVec vec2[2];
    __shared__ double sgn[SB];
    const int t = threadIdx.x;
    const int ti = t >> 4, tk = t & 15;
    if (*panel_zero != 0) {
        // all-zero panel: identity transformation (tau = 0, V = 0); keep the buffers the next kernels read well-defined
        for (int e = t; e < SB * SB; e += 256) {
            const int i = e / SB, k = e % SB;
            u_out[e] = (i == k) ? 1.0 : 0.0;
            t_out[e] = 0.0;
            pan_v1[i + (int64_t)k * ldp] = 0.0;
            pan_v2[i + (int64_t)k * ldp] = 0.0;
            if (i < pw && k < pw) atop[i + (int64_t)k * lda] = 0.0;
        }
        for (int i = t; i < pw; i += 256) tau[i] = 0.0;
        return;
    }
    tiny_fetch_colmajor(s, g3, pw, true);
    for (int e = t; e < SB * SB; e += 256) s2[e / SB][e % SB] = rtot[e];
    __syncthreads();
    // Is the panel orthonormal already after two passes?  |G3 - I| at the rounding level of the Gram product itself means a
    // third pass would only factor noise: R3 = I then (the usual case; saves three of the six elimination loops).
    Tiny r3;
    tiny_load_lds(r3, s);
    bool off = false;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double dev = fabs(r3.v[a][b] - ((ti + 16 * a == tk + 16 * b) ? 1.0 : 0.0));
            if (!(dev <= skip_tol)) off = true;
        }
    const bool third = __syncthreads_or(off ? 1 : 0) != 0;
    Tiny rt;
    if (third) {
        if (!tiny_chol_upper(r3, vec)) atomicOr(flags + SB_FLAG_FAIL, 4);
        {
            // G3 = Q2'Q2 must be close to the identity (R3 ~ I), or the panel was too ill-conditioned for three passes
            bool bad = false;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (ti + 16 * a == tk + 16 * b && !(fabs(r3.v[a][b] - 1.0) < 1e-8)) bad = true;
            if (bad) atomicOr(flags + SB_FLAG_FAIL, 8);
        }
        __syncthreads();
        tiny_store_lds(r3, s);                      // s = R3
        __syncthreads();
        tiny_matmul(rt, s, s2);                     // rtot = R3 * rtot
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) rtot[(ti + 16 * a) * SB + tk + 16 * b] = rt.v[a][b];
        // s <- R3 with reciprocal diagonal (operand of the triangular solve)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (ti + 16 * a == tk + 16 * b) s[ti + 16 * a][tk + 16 * b] = 1.0 / r3.v[a][b];
    } else {
        tiny_load_lds(rt, s2);                      // rtot unchanged
    }
    // W = Qtop (identity-padded)
    Tiny w;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ti + 16 * a, k = tk + 16 * b;
            w.v[a][b] = (i < pw && k < pw) ? atop[i + (int64_t)k * lda] : ((i == k) ? 1.0 : 0.0);
        }
    __syncthreads();
    if (third) {
        tiny_trsm_right_upper(w, s, vec);       // Qtop R3^-1
        __syncthreads();
    }
    tiny_lu_modified(w, vec, vec2, sgn);
    // A top block: S rtot (upper incl. diagonal) | L1 strictly below;  V copies: unit lower triangle;  U for the row solve
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ti + 16 * a, k = tk + 16 * b;
            const bool in = (i < pw && k < pw);
            if (in) atop[i + (int64_t)k * lda] = (k >= i) ? sgn[i] * rt.v[a][b] : w.v[a][b];
            if (i < pw) {
                const double vv = (k < pw) ? ((i == k) ? 1.0 : ((i > k) ? w.v[a][b] : 0.0)) : 0.0;
                pan_v1[i + (int64_t)k * ldp] = vv;
                pan_v2[i + (int64_t)k * ldp] = vv;
            }
        }
    // T L1' = -U S:  X = -U S (upper), then X <- X (L1')^-1 with L1' unit upper = transpose of the strict lower part of w
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ti + 16 * a, k = tk + 16 * b;
            s2[k][i] = (i > k) ? w.v[a][b] : ((i == k) ? 1.0 : 0.0);       // s2 = L1' (unit diagonal = its own reciprocal)
        }
    Tiny x;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ti + 16 * a, k = tk + 16 * b;
            x.v[a][b] = (k >= i && i < pw && k < pw) ? -w.v[a][b] * sgn[k] : 0.0;
        }
    __syncthreads();
    tiny_trsm_right_upper<true>(x, s2, vec);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ti + 16 * a, k = tk + 16 * b;
            const bool in = (i < pw && k < pw && k >= i);
            t_out[i + k * SB] = in ? x.v[a][b] : 0.0;
            if (i == k && i < pw) tau[i] = x.v[a][b];
        }
    // X = R3^-1 U^-1 for the rows below the top block (v = q X, one GEMM): I <- I R3^-1 (s still holds R3 with its
    // reciprocal diagonal), then <- (.) U^-1 with U = upper part of w
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = ti + 16 * a, k = tk + 16 * b;
            const bool in = (i < pw && k < pw);
            s2[i][k] = in ? ((k > i) ? w.v[a][b] : ((k == i) ? 1.0 / w.v[a][b] : 0.0)) : ((i == k) ? 1.0 : 0.0);
            x.v[a][b] = (i == k) ? 1.0 : 0.0;
        }
    __syncthreads();
    if (third) {
        tiny_trsm_right_upper<true>(x, s, vec);
        __syncthreads();
    }
    tiny_trsm_right_upper<true>(x, s2, vec);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) u_out[(ti + 16 * a) + (tk + 16 * b) * SB] = x.v[a][b];      // column-major
}

// M = T' N1 T  ->  TM (2 SB x SB, column-major, ld = 2 SB): rows 0 .. SB-1 = T, rows SB .. = -M / 2
__global__ __launch_bounds__(256) void sb_tm_kernel(const double *__restrict__ tmat, const double *__restrict__ n1, int pw,
                                                    double *__restrict__ tm) {
    __shared__ double ts[SB][SB_P];    // T[row][col]
    __shared__ double tt[SB][SB_P];    // T'
    __shared__ double x[SB][SB_P];
    const int t = threadIdx.x;
    const int ti = t >> 4, tk = t & 15;
    for (int e = t; e < SB * SB; e += 256) {
        const int c = e / SB, r = e % SB;
        const double tv = tmat[r + c * SB];
        ts[r][c] = tv;
        tt[c][r] = tv;
        x[r][c] = (r < pw && c < pw) ? n1[r + c * SB] : 0.0;
    }
    __syncthreads();
    Tiny y;
    tiny_matmul(y, x, ts);             // N1 T
    __syncthreads();
    tiny_store_lds(y, x);
    __syncthreads();
    Tiny m;
    tiny_matmul(m, tt, x);             // T' (N1 T)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = ti + 16 * a, c = tk + 16 * b;
            tm[r + c * (2 * SB)] = ts[r][c];
            tm[SB + r + c * (2 * SB)] = -0.5 * m.v[a][b];
        }
}

// rows [row0, row0 + rows) of the first V copy of the panel buffer -> A's panel (LAPACK storage of the reflectors) and the
// second V copy; columns >= pw of both copies are zeroed
__global__ __launch_bounds__(256) void sb_copy_v_kernel(double *__restrict__ v1, double *__restrict__ v2, int64_t ldv,
                                                        double *__restrict__ p, int64_t ldp, int row0, int rows, int pw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (i >= rows) return;
    const int64_t o = row0 + i + (int64_t)c * ldv;
    double v = 0.0;
    if (c < pw) {
        v = v1[o];
        p[row0 + i + (int64_t)c * ldp] = v;
    } else {
        v1[o] = 0.0;
    }
    v2[o] = v;
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

constexpr size_t SB_WS = (size_t)7 << 20;   // split-K workspace of one launch chain (7 M doubles >= 128 * 64 * 768)
size_t sy2sb_work_doubles(int n) { return (size_t)n * 8 * SB + 12 * SB * SB + 64 + 2 * SB_WS; }
int sy2sb_bandwidth() { return SB; }

// ---- trailing matrix sharded over ranks (jxg_eigh_set_band_dist) --------------------------------------------------------
// Ownership by ABSOLUTE block rows of `block` samples, dealt cyclically: rank r keeps the lower-triangle entries of the rows
// [b block, (b + 1) block), b mod world == r, current and never reads anybody else's.  Per panel two collectives through
// `allreduce(user, count)` on the staging buffer (sum over the ranks of its first `count` doubles, on the eigensolver's stream):
//   (1) Z = A22 V as partial sums over the owned block rows (rectangle left of the diagonal block from both sides + the
//       diagonal block) into a zeroed n x 64 buffer: the sum is the same bits on every rank, and everything derived from it
//       (M, W, the next panel's factorisation chain) is replicated arithmetic on identical inputs;
//   (2) the block column of the NEXT panel, updated by the owners of its rows, gathered as a sum with zeros and written back on
//       every rank (it also holds the reflectors the Q1 back-transformation reads later).
// The rest of the rank-2k update touches owned rows only.  From a trailing size <= 2 `block` on the trailing square is gathered
// once and the last panels run replicated.
struct BandDist {
    int rank = 0, world = 1;
    int (*allreduce)(void *, int64_t) = nullptr;
    void *user = nullptr;
    double *staging = nullptr;
    int64_t staging_doubles = 0;
    int min_n = 8192;
    int block = 2048;
};
static BandDist g_band;

int sy2sb_set_band_dist(int rank, int world, int (*allreduce)(void *, int64_t), void *user, double *staging,
                        int64_t staging_doubles, int min_n, int block) {
    if (world < 1 || rank < 0 || rank >= world) return fail("sy2sb_set_band_dist: bad rank / world");
    g_band.rank = rank;
    g_band.world = world;
    g_band.allreduce = allreduce;
    g_band.user = user;
    g_band.staging = staging;
    g_band.staging_doubles = staging_doubles;
    g_band.min_n = min_n > 0 ? min_n : 8192;
    g_band.block = block >= 2 * SB ? (block / SB) * SB : 2048;
    return 0;
}
int64_t sy2sb_band_staging_doubles(int n) {
    const int64_t b = g_band.block + 2 * SB;
    return std::max<int64_t>((int64_t)n * SB, (2 * b) * (2 * b));
}
int sy2sb_band_dist_active(int n) {
    const bool force_single = getenv("JXGPU_DIST_EIGH_FORCE") && atoi(getenv("JXGPU_DIST_EIGH_FORCE")) != 0;
    return ((g_band.world > 1 || force_single) && g_band.allreduce && n >= g_band.min_n &&
            g_band.staging_doubles >= sy2sb_band_staging_doubles(n)) ? 1 : 0;
}

// d_a (n x n, column-major, symmetric, lower referenced) -> band form in place (V below the band, see above);
// d_tau (n): tau of every stage-1 reflector (0 beyond the last eliminated column); d_ab (ldab x n): band copy for
// stage 2.  d_flags[0] != 0 on return (after the caller's synchronisation) means a panel could not be factored.
//
// Look-ahead: the trailing update of panel p is split into the block column of panel p + 1 (a narrow GEMM) and the
// rest; the factorisation chain of panel p + 1 (~15 small launches, latency-bound) then runs on a side stream beside the
// rest of the update, which is what fills the chip.  Panel buffers, T and the zero-panel flag alternate by panel parity.
int sy2sb_lower(hipStream_t st, double *d_a, int n, double *d_tau, double *d_ab, int ldab, double *d_work,
                int *d_flags, bool allow_shard) {
    const int ncol = n - SB - 1;                     // columns with entries below the band
    JX_HIP(hipMemsetAsync(d_tau, 0, sizeof(double) * (size_t)n, st));
    JX_HIP(hipMemsetAsync(d_flags, 0, sizeof(int) * 4, st));
    if (ncol > 0) {
        double *pan[2] = {d_work, d_work + (size_t)n * 4 * SB};     // (n, 4 SB) each, ld = n: [Z | V | W | V]
        double *p = d_work + (size_t)n * 8 * SB;
        double *gq = p; p += SB * SB;                // Gram matrices of the factorisation chain (column-major, ld = SB)
        double *gu = p; p += SB * SB;                // N1 = V'Z of the update chain
        double *rmat = p; p += SB * SB;              // R^-1 of the current pass (column-major)
        double *umat = p; p += SB * SB;              // R3^-1 U^-1
        double *rtot = p; p += SB * SB;
        double *tmat[2];
        tmat[0] = p; p += SB * SB;                   // T of the panel (column-major), by parity
        tmat[1] = p; p += SB * SB;
        double *tm = p; p += 2 * SB * SB;            // [T; -M/2]
        double *tmp1 = p; p += SB * SB;
        double *tmp2 = p; p += SB * SB;
        double *ws_main = p; p += SB_WS;             // split-K slices of the update chain (main stream) ...
        double *ws_side = p; p += SB_WS;             // ... and of the factorisation chain (side stream)
        JX_HIP(hipMemsetAsync(d_work, 0, sizeof(double) * (size_t)n * 8 * SB, st));
        const int64_t ld = n;
        const double eps = 2.220446049250313e-16;
        static hipStream_t side = nullptr;
        static hipEvent_t ev_a[2] = {nullptr, nullptr}, ev_qr[2] = {nullptr, nullptr};
        static const bool lookahead = !(getenv("JXGPU_SY2SB_LOOKAHEAD") && atoi(getenv("JXGPU_SY2SB_LOOKAHEAD")) == 0);
        if (!side) {
            // (a high-priority side stream was measured with the persistent update kernel: no effect on the band reduction -- what the
            // chain needs is free CUs, k_dgemm.hip dsyr2k_lower_nt -- and 16 ms more in the divide and conquer; not kept)
            JX_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            for (int q = 0; q < 2; ++q) {
                JX_HIP(hipEventCreateWithFlags(&ev_a[q], hipEventDisableTiming));
                JX_HIP(hipEventCreateWithFlags(&ev_qr[q], hipEventDisableTiming));
            }
        }
        // factorisation chain of the panel at column j0 on stream s
        auto panel_qr = [&](hipStream_t s, int j0, int par, double *ws) -> int {
            const int pw = (ncol - j0 < SB) ? (ncol - j0) : SB;
            const int nt = n - j0 - SB;
            double *pp = d_a + (j0 + SB) + (int64_t)j0 * ld;
            double *v1 = pan[par] + (size_t)SB * ld, *v2 = pan[par] + (size_t)3 * SB * ld;
            int *pz = d_flags + 2 + par;
            // CholeskyQR passes 1 and 2: G = P'P, R = chol(G), P <- P R^-1 (in place: a workgroup of the product reads only
            // the rows it writes, and all of them before its epilogue)
            for (int pass = 0; pass < 2; ++pass) {
                if (dgemm(s, true, false, pw, pw, nt, 1.0, pp, ld, pp, ld, 0.0, gq, SB, 0, ws, SB_WS)) return 1;
                const double shift = (pass == 0) ? 11.0 * ((double)nt * pw + (double)pw * (pw + 1)) * eps : 0.0;
                hipLaunchKernelGGL(sb_chol_kernel, dim3(1), dim3(256), 0, s, gq, pw, shift, pass == 0 ? 1 : 0, rmat, rtot,
                                   d_flags, pz);
                JX_LAUNCH_CHECK();
                if (dgemm(s, false, false, nt, pw, pw, 1.0, pp, ld, rmat, SB, 0.0, pp, ld, 1, nullptr, 0)) return 1;
            }
            // pass 3 + reconstruction
            if (dgemm(s, true, false, pw, pw, nt, 1.0, pp, ld, pp, ld, 0.0, gq, SB, 0, ws, SB_WS)) return 1;
            hipLaunchKernelGGL(sb_recon_kernel, dim3(1), dim3(256), 0, s, gq, pw, pp, ld, rmat, umat, rtot, tmat[par],
                               d_tau + j0, v1, v2, ld, d_flags, pz, 8.0 * eps * sqrt((double)nt));
            JX_LAUNCH_CHECK();
            if (nt > pw) {
                if (dgemm(s, false, false, nt - pw, pw, pw, 1.0, pp + pw, ld, umat, SB, 0.0, v1 + pw, ld, 1, nullptr, 0)) return 1;
                hipLaunchKernelGGL(sb_copy_v_kernel, dim3(ceil_div(nt - pw, 256), SB), dim3(256), 0, s, v1, v2, ld, pp, ld,
                                   pw, nt - pw, pw);
                JX_LAUNCH_CHECK();
            }
            return 0;
        };
        // ---- sharded trailing matrix (BandDist above) ---------------------------------------------------------------
        bool sharded = allow_shard && sy2sb_band_dist_active(n) != 0;
        const int BR = g_band.block;
        // owned block rows of the trailing matrix that starts at absolute row s0, in trailing coordinates, rows >= lo only
        auto for_owned = [&](int s0, int lo, auto &&fn) -> int {
            for (int b = s0 / BR; (int64_t)b * BR < n; ++b) {
                if (b % g_band.world != g_band.rank) continue;
                const int a0 = std::max(b * BR, s0 + lo), a1 = std::min((b + 1) * BR, n);
                if (a1 <= a0) continue;
                if (fn(a0 - s0, a1 - a0)) return 1;
            }
            return 0;
        };
        auto reduce_staging = [&](int64_t count) -> int {
            if (count > g_band.staging_doubles) return fail("sy2sb: staging buffer of the band collectives too small");
            if (g_band.allreduce(g_band.user, count)) return fail("sy2sb: the all-reduce callback failed");
            return 0;
        };
        if (panel_qr(st, 0, 0, ws_main)) return 1;
        int par = 0;
        for (int j0 = 0; j0 < ncol; j0 += SB, par ^= 1) {
            const int pw = (ncol - j0 < SB) ? (ncol - j0) : SB;
            const int nt = n - j0 - SB;
            double *a22 = d_a + (j0 + SB) + (int64_t)(j0 + SB) * ld;  // trailing matrix
            double *zc = pan[par], *v1 = pan[par] + (size_t)SB * ld, *wc = pan[par] + (size_t)2 * SB * ld;
            const bool has_next = j0 + SB < ncol;
            if (j0 > 0 && lookahead) JX_HIP(hipStreamWaitEvent(st, ev_qr[par], 0));     // this panel's chain ran on the side stream
            if (sharded && nt <= 2 * BR) {
                // the last panels run replicated: one gather of the trailing square (owners' rows, zeros elsewhere, summed)
                double *stg = g_band.staging;
                JX_HIP(hipMemsetAsync(stg, 0, sizeof(double) * (size_t)nt * nt, st));
                if (for_owned(j0 + SB, 0, [&](int r0, int rl) -> int {
                        JX_HIP(hipMemcpy2DAsync(stg + r0, sizeof(double) * (size_t)nt, a22 + r0, sizeof(double) * (size_t)ld,
                                                sizeof(double) * (size_t)rl, (size_t)(r0 + rl), hipMemcpyDeviceToDevice, st));
                        return 0;
                    }))
                    return 1;
                if (reduce_staging((int64_t)nt * nt)) return 1;
                JX_HIP(hipMemcpy2DAsync(a22, sizeof(double) * (size_t)ld, stg, sizeof(double) * (size_t)nt,
                                        sizeof(double) * (size_t)nt, (size_t)nt, hipMemcpyDeviceToDevice, st));
                sharded = false;
            }
            if (pw < SB) {
                // last, narrower panel: the columns j0 + pw .. j0 + SB - 1 of the block row see Q' from the left only
                const int nc = SB - pw;
                double *cb = d_a + (j0 + SB) + (int64_t)(j0 + pw) * ld;
                if (dgemm(st, true, false, pw, nc, nt, 1.0, v1, ld, cb, ld, 0.0, tmp1, SB, 1, nullptr, 0)) return 1;
                if (dgemm(st, true, false, pw, nc, pw, 1.0, tmat[par], SB, tmp1, SB, 0.0, tmp2, SB, 1, nullptr, 0)) return 1;
                if (dgemm(st, false, false, nt, nc, pw, -1.0, v1, ld, tmp2, SB, 1.0, cb, ld, 1, nullptr, 0)) return 1;
            }
            // two-sided update of the trailing matrix
            if (!sharded) {
                if (dsymm_lower(st, nt, pw, 1.0, a22, ld, v1, ld, 0.0, zc, ld, ws_main, SB_WS)) return 1;
            } else {
                // Z = A22 V from the owned block rows: the rectangle left of the diagonal block acts from both sides
                JX_HIP(hipMemsetAsync(zc, 0, sizeof(double) * (size_t)ld * SB, st));
                if (for_owned(j0 + SB, 0, [&](int r0, int rl) -> int {
                        if (r0 > 0) {
                            if (dgemm(st, false, false, rl, pw, r0, 1.0, a22 + r0, ld, v1, ld, 1.0, zc + r0, ld, 0, ws_main, SB_WS)) return 1;
                            if (dgemm(st, true, false, r0, pw, rl, 1.0, a22 + r0, ld, v1 + r0, ld, 1.0, zc, ld, 0, ws_main, SB_WS)) return 1;
                        }
                        return dsymm_lower(st, rl, pw, 1.0, a22 + r0 + (int64_t)r0 * ld, ld, v1 + r0, ld, 1.0, zc + r0, ld, ws_main, SB_WS);
                    }))
                    return 1;
                const int64_t cnt = (int64_t)(pw - 1) * ld + nt;
                JX_HIP(hipMemcpyAsync(g_band.staging, zc, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToDevice, st));
                if (reduce_staging(cnt)) return 1;
                JX_HIP(hipMemcpyAsync(zc, g_band.staging, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToDevice, st));
            }
            if (dgemm(st, true, false, pw, pw, nt, 1.0, v1, ld, zc, ld, 0.0, gu, SB, 0, ws_main, SB_WS)) return 1;
            hipLaunchKernelGGL(sb_tm_kernel, dim3(1), dim3(256), 0, st, tmat[par], gu, pw, tm);
            JX_LAUNCH_CHECK();
            if (dgemm(st, false, false, nt, pw, 2 * SB, 1.0, zc, ld, tm, 2 * SB, 0.0, wc, ld, 1, nullptr, 0)) return 1;
            if (has_next) {
                // block column of the next panel first: A22[:, 0:SB] -= [V | W] ([W | V][0:SB, :])'
                const int nb = SB;                                      // has_next => nt > SB
                if (!sharded) {
                    if (dgemm(st, false, true, nt, nb, 2 * SB, -1.0, v1, ld, wc, ld, 1.0, a22, ld, 1, nullptr, 0)) return 1;
                } else {
                    // owners update their rows of the block column; gathered (sum with zeros) and written back everywhere
                    double *stg = g_band.staging;
                    JX_HIP(hipMemsetAsync(stg, 0, sizeof(double) * (size_t)nt * nb, st));
                    if (for_owned(j0 + SB, 0, [&](int r0, int rl) -> int {
                            if (dgemm(st, false, true, rl, nb, 2 * SB, -1.0, v1 + r0, ld, wc, ld, 1.0, a22 + r0, ld, 1, nullptr, 0)) return 1;
                            JX_HIP(hipMemcpy2DAsync(stg + r0, sizeof(double) * (size_t)nt, a22 + r0, sizeof(double) * (size_t)ld,
                                                    sizeof(double) * (size_t)rl, (size_t)nb, hipMemcpyDeviceToDevice, st));
                            return 0;
                        }))
                        return 1;
                    if (reduce_staging((int64_t)nt * nb)) return 1;
                    JX_HIP(hipMemcpy2DAsync(a22, sizeof(double) * (size_t)ld, stg, sizeof(double) * (size_t)nt,
                                            sizeof(double) * (size_t)nt, (size_t)nb, hipMemcpyDeviceToDevice, st));
                }
                if (lookahead) {
                    JX_HIP(hipEventRecord(ev_a[par], st));
                    JX_HIP(hipStreamWaitEvent(side, ev_a[par], 0));
                    if (panel_qr(side, j0 + SB, par ^ 1, ws_side)) return 1;
                    JX_HIP(hipEventRecord(ev_qr[par ^ 1], side));
                }
                // the rest of the trailing update runs beside that chain
                if (!sharded) {
                    if (dsyr2k_lower_nt(st, nt - nb, 2 * SB, -1.0, v1 + nb, ld, wc + nb, ld, 1.0, a22 + nb + (int64_t)nb * ld, ld))
                        return 1;
                } else if (for_owned(j0 + SB, nb, [&](int r0, int rl) -> int {
                               // owned rows only: the rectangle between the next panel's block column and the diagonal block, then
                               // the diagonal block
                               if (r0 > nb &&
                                   dgemm(st, false, true, rl, r0 - nb, 2 * SB, -1.0, v1 + r0, ld, wc + nb, ld, 1.0,
                                         a22 + r0 + (int64_t)nb * ld, ld, 1, nullptr, 0))
                                   return 1;
                               return dsyr2k_lower_nt(st, rl, 2 * SB, -1.0, v1 + r0, ld, wc + r0, ld, 1.0,
                                                      a22 + r0 + (int64_t)r0 * ld, ld);
                           })) {
                    return 1;
                }
                if (!lookahead && panel_qr(st, j0 + SB, par ^ 1, ws_main)) return 1;
            } else {
                if (dsyr2k_lower_nt(st, nt, 2 * SB, -1.0, v1, ld, wc, ld, 1.0, a22, ld)) return 1;
            }
        }
    }
    hipLaunchKernelGGL(sb_extract_band_kernel, dim3((unsigned)(((int64_t)n * ldab + 255) / 256)), dim3(256), 0, st, d_a, n,
                       d_ab, ldab);
    JX_LAUNCH_CHECK();
    return 0;
}

}  // namespace jx
