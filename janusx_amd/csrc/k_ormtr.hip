// Back-transformation C <- Q C of the eigenvectors of the tridiagonal (or band) matrix, Q = H_0 H_1 ... from sytrd_lower
// (LAPACK dsytrd(lower) storage: v_j below the sub-diagonal of column j, tau_j) or from the band reduction (k_sy2sb.hip: unit
// entry `off` rows below the diagonal) -- LAPACK dormtr (left, lower, no-transpose), the last stage behind
// src/math/eigh.rs:1422-1528.
// Wide compact-WY blocks, BLAS-3 only: for a block of nb reflectors with explicit V
//   I - V T V',   T^-1 = strict_upper(V'V) + diag(1 / tau)
// the block needs one Gram product, the inverse of a triangular nb x nb matrix (64 x 64 diagonal blocks in one workgroup each,
// then log2(nb / 64) levels of  T12 = -T11 (M12 T22)), the product V T, and two products with the eigenvectors:
//   W = V' C (k = rows),   C -= (V T) W (k = nb)          -- 2 n^3 flop over the blocks, everything else O(nb n^2).
// From JXGPU_OZ_MIN_N rows on every large product runs on the int8 matrix pipes (k_ozgemm.hip: operands sliced into
// base-254 digit planes, exact i32 digit products, f64 combination; 160 - 170 TFLOP/s-equivalent against the 78.6 TFLOP/s f64
// MFMA roof; V and V T are sliced once per block, C once per block); below, and for the small triangular inverse, the own f64
// MFMA GEMM (k_dgemm.hip).  No vendor BLAS (rounds 1 - 3: rocBLAS dgemm / dtrsm, 284 ms of the 1.40 s at n = 20 000).
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "k_ozgemm.h"

namespace jx {

int dgemm(hipStream_t st, bool ta, bool tb, int m, int n, int k, double alpha, const double *a, int64_t lda, const double *b,
          int64_t ldb, double beta, double *c, int64_t ldc, int ksplit, double *ws, size_t ws_doubles);

constexpr int OT_NB = 1024;
constexpr int OT_TB = 64;     // diagonal block of the triangular inverse

// vc (rows, nbk) column-major = explicit reflectors of columns jb .. jb+nbk-1 restricted to rows jb+off .. n-1
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

// m (nb x nb, column-major, ld = nb) holds G = V'V in its leading nbk x nbk upper triangle: keep the strict upper triangle,
// put 1/tau on the diagonal (1 for identity reflectors), zero the strict lower triangle; rows / columns >= nbk: identity.
__global__ void ot_fix_m_kernel(double *__restrict__ m, int nbk, int nb, const double *__restrict__ tau, int jb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= nb || r >= nb) return;
    double *p = m + (int64_t)c * nb + r;
    if (r >= nbk || c >= nbk) *p = (r == c) ? 1.0 : 0.0;
    else if (r > c) *p = 0.0;
    else if (r == c) {
        const double t = tau[jb + c];
        *p = (t != 0.0) ? 1.0 / t : 1.0;
    }
}

// inverse of the 64 x 64 upper triangular diagonal blocks of m into t (both ld): thread c solves M x = e_c upwards
__global__ __launch_bounds__(OT_TB) void ot_triinv_diag_kernel(const double *__restrict__ m, double *__restrict__ t, int ld) {
    __shared__ double s[OT_TB][OT_TB + 1];
    const int o = blockIdx.x * OT_TB, c = threadIdx.x;
    for (int j = 0; j < OT_TB; ++j) s[c][j] = m[(int64_t)(o + j) * ld + o + c];   // s[row][col], coalesced over rows
    __syncthreads();
    double x[OT_TB];
#pragma unroll
    for (int i = OT_TB - 1; i >= 0; --i) {
        double acc = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = i + 1; k < OT_TB; ++k) acc -= s[i][k] * x[k];
        x[i] = (i <= c) ? acc / s[i][i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < OT_TB; ++i) t[(int64_t)(o + c) * ld + o + i] = x[i];
}

// t (nb x nb, ld = nb) = inverse of the upper triangular m (nb x nb; rows / columns >= nbk are identity); nb = 64 2^q.
// xw: nb * nb / 4 doubles; gws: split workspace of the GEMMs
static int ot_triinv_upper(hipStream_t st, const double *m, double *t, int nb, int nbk, double *xw, double *gws, size_t gws_doubles) {
    JX_HIP(hipMemsetAsync(t, 0, sizeof(double) * (size_t)nb * nb, st));
    hipLaunchKernelGGL(ot_triinv_diag_kernel, dim3(nb / OT_TB), dim3(OT_TB), 0, st, m, t, nb);
    JX_LAUNCH_CHECK();
    for (int s = OT_TB; s < nb; s *= 2) {
        for (int o = 0; o + s < nbk; o += 2 * s) {
            const double *m12 = m + (int64_t)(o + s) * nb + o;
            const double *t11 = t + (int64_t)o * nb + o, *t22 = t + (int64_t)(o + s) * nb + (o + s);
            double *t12 = t + (int64_t)(o + s) * nb + o;
            double *x = xw + (size_t)(o / (2 * s)) * s * s;
            if (dgemm(st, false, false, s, s, s, 1.0, m12, nb, t22, nb, 0.0, x, s, 0, gws, gws_doubles)) return 1;
            if (dgemm(st, false, false, s, s, s, -1.0, t11, nb, x, s, 0.0, t12, nb, 0, gws, gws_doubles)) return 1;
        }
    }
    return 0;
}

static int ot_round_nb(int nb) {
    int r = OT_TB;
    while (r * 2 <= nb) r *= 2;
    return r;
}

int ormtr_oz_min_n() {
    static const int v = getenv("JXGPU_OZ_MIN_N") ? atoi(getenv("JXGPU_OZ_MIN_N")) : 3000;
    return v;
}

// ---- plan: everything that does not depend on C (V, its images, T, V T), prepared ahead on a stream of its own ------------
struct OrmtrBlock {
    int jb = 0, nbk = 0, rows = 0;
    size_t o_vt = 0, o_vtt = 0;      // byte offsets into `store`: oz images of V' and V T, or the f64 blocks V and V T
};
struct OrmtrPlan {
    int n = 0, off = 0, nref = 0, nb = 0;
    bool use_oz = false;
    std::vector<OrmtrBlock> blocks;  // in application order (last reflector block first)
    ScratchLease work;               // f64 temporaries of the preparation
    ScratchLease store;              // per-block operands kept for the application
};

OrmtrPlan *ormtr_plan_new() { return new OrmtrPlan(); }
void ormtr_plan_free(OrmtrPlan *p) { delete p; }

// Prepares the plan on stream `st` (asynchronous: the caller orders the application behind it).
int ormtr_prepare(hipStream_t st, const double *d_a, int n, int off, int nref, const double *d_tau, OrmtrPlan &P) {
    P.n = n;
    P.off = off;
    P.nref = nref;
    P.blocks.clear();
    if (n < 2 || nref < 1) return 0;
    static const bool oz_on = !(getenv("JXGPU_ORMTR_OZ") && atoi(getenv("JXGPU_ORMTR_OZ")) == 0);
    P.use_oz = oz_on && n >= ormtr_oz_min_n();
    // wider blocks for large n (rocBLAS form at n = 20000: 512 -> 378 ms, 1024 -> 315, 2048 -> 281, 4096 -> 327)
    int nb = (getenv("JXGPU_ORMTR_NB") && atoi(getenv("JXGPU_ORMTR_NB")) > 0) ? atoi(getenv("JXGPU_ORMTR_NB"))
                                                                              : (n >= 12000 ? 2 * OT_NB : OT_NB);
    nb = ot_round_nb(std::max(nb, OT_TB));
    while (nb > OT_TB && nb / 2 >= nref) nb /= 2;
    P.nb = nb;
    const int nblocks = (nref + nb - 1) / nb;
    size_t store_bytes = 0;
    for (int b = nblocks - 1; b >= 0; --b) {
        OrmtrBlock k;
        k.jb = b * nb;
        k.nbk = (nref - k.jb < nb) ? (nref - k.jb) : nb;
        k.rows = n - k.jb - off;
        k.o_vt = store_bytes;
        store_bytes += P.use_oz ? oz_image_bytes(k.nbk, k.rows) : ((sizeof(double) * (size_t)k.rows * k.nbk + 255) & ~(size_t)255);
        k.o_vtt = store_bytes;
        store_bytes += P.use_oz ? oz_image_bytes(k.rows, k.nbk) : ((sizeof(double) * (size_t)k.rows * k.nbk + 255) & ~(size_t)255);
        P.blocks.push_back(k);
    }
    // f64 temporaries: vc (n x nb) | vt (n x nb) | mm | tt (nb x nb each) | xw (nb x nb / 4) | gws
    const size_t nvc = (size_t)n * nb, nmm = (size_t)nb * nb;
    const size_t ngws = std::max<size_t>(4 * nmm, (size_t)1 << 20);
    if (P.work.take(3, sizeof(double) * (2 * nvc + 2 * nmm + nmm / 4 + ngws))) return 1;
    if (P.store.take(5, store_bytes + 256)) return 1;
    double *const vc = P.work.as<double>(), *const vt = vc + nvc, *const mm = vt + nvc, *const tt = mm + nmm;
    double *const xw = tt + nmm, *const gws = xw + nmm / 4;
    char *const store = P.store.as<char>();
    for (const OrmtrBlock &k : P.blocks) {
        double *vcb = P.use_oz ? vc : reinterpret_cast<double *>(store + k.o_vt);
        double *vtb = P.use_oz ? vt : reinterpret_cast<double *>(store + k.o_vtt);
        hipLaunchKernelGGL(ot_extract_v_kernel, dim3((k.rows + 255) / 256, k.nbk), dim3(256), 0, st, d_a, n, k.jb, k.nbk, d_tau,
                           vcb, k.rows, off);
        JX_LAUNCH_CHECK();
        if (P.use_oz) {
            OzImage i_vt = oz_image_at(store + k.o_vt, k.nbk, k.rows);
            if (oz_slice(st, vcb, k.rows, 1, i_vt)) return 1;                                   // element (j, r) = vc[r + j rows]
            if (oz_mm(st, i_vt, i_vt, k.nbk, k.nbk, 1.0, 0.0, mm, nb, 1)) return 1;             // upper tiles of V'V
        } else {
            if (dgemm(st, true, false, k.nbk, k.nbk, k.rows, 1.0, vcb, k.rows, vcb, k.rows, 0.0, mm, nb, 0, gws, ngws)) return 1;
        }
        hipLaunchKernelGGL(ot_fix_m_kernel, dim3((nb + 63) / 64, nb), dim3(64), 0, st, mm, k.nbk, nb, d_tau, k.jb);
        JX_LAUNCH_CHECK();
        if (ot_triinv_upper(st, mm, tt, nb, k.nbk, xw, gws, ngws)) return 1;
        // V T (rows x nbk): small next to the two products with C (k = nbk against ncols columns)
        if (dgemm(st, false, false, k.rows, k.nbk, k.nbk, 1.0, vcb, k.rows, tt, nb, 0.0, vtb, k.rows, 0, gws, ngws)) return 1;
        if (P.use_oz) {
            OzImage i_vtt = oz_image_at(store + k.o_vtt, k.rows, k.nbk);
            if (oz_slice(st, vtb, 1, k.rows, i_vtt)) return 1;                                  // element (r, j) = vt[r + j rows]
        }
    }
    return 0;
}

// C <- Q C for d_c (n, ncols), ld = n, on stream `st` (the plan must be complete in stream order).  Synchronises `st` at the end.
int ormtr_apply(hipStream_t st, const OrmtrPlan &P, double *d_c, int ncols) {
    if (P.blocks.empty() || ncols < 1) return 0;
    const int n = P.n, nb = P.nb;
    const size_t nw = (size_t)nb * ncols;
    const size_t ngws = std::max<size_t>(4 * (size_t)nb * nb, (size_t)1 << 20);
    size_t b_c = 0, b_w = 0;
    if (P.use_oz) {
        b_c = oz_image_bytes(ncols, n - P.off);      // C: rows = eigenvector columns, k = matrix rows
        b_w = oz_image_bytes(ncols, nb);             // W: rows = eigenvector columns, k = reflectors
    }
    ScratchLease ws;
    if (ws.take(6, sizeof(double) * (nw + ngws) + b_c + b_w + 512)) return 1;
    double *const w = ws.as<double>(), *const gws = w + nw;
    char *const img = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(gws + ngws) + 255) & ~(uintptr_t)255);
    const char *const store = P.store.as<char>();
    // JXGPU_EIGH_TRACE: per-part times summed over the blocks (synchronises after every part)
    const bool trace = getenv("JXGPU_EIGH_TRACE") != nullptr;
    double tsum[4] = {0, 0, 0, 0};
    auto tlast = std::chrono::steady_clock::now();
    auto mark = [&](int slot) {
        if (!trace) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        tsum[slot] += std::chrono::duration<double, std::milli>(now - tlast).count();
        tlast = now;
    };
    mark(3);
    for (const OrmtrBlock &k : P.blocks) {
        double *csub = d_c + (k.jb + P.off);          // rows jb+off .. n-1 of every column
        if (P.use_oz) {
            const OzImage i_vt = oz_image_at(const_cast<char *>(store) + k.o_vt, k.nbk, k.rows);
            const OzImage i_vtt = oz_image_at(const_cast<char *>(store) + k.o_vtt, k.rows, k.nbk);
            OzImage i_c = oz_image_at(img, ncols, k.rows), i_w = oz_image_at(img + b_c, ncols, k.nbk);
            if (oz_slice(st, csub, n, 1, i_c)) return 1;                // element (col, r) = csub[r + col n]
            mark(0);
            // V' is upper trapezoidal in (reflector, row): mode 2 skips the k steps left of a row tile's diagonal
            if (oz_mm(st, i_vt, i_c, k.nbk, ncols, 1.0, 0.0, w, nb, 2)) return 1;               // W = V' C
            mark(1);
            if (oz_slice(st, w, nb, 1, i_w)) return 1;                  // element (col, j) = w[j + col nb]
            if (oz_mm(st, i_vtt, i_w, k.rows, ncols, -1.0, 1.0, csub, n, 0)) return 1;          // C -= (V T) W
            mark(2);
        } else {
            const double *vcb = reinterpret_cast<const double *>(store + k.o_vt);
            const double *vtb = reinterpret_cast<const double *>(store + k.o_vtt);
            if (dgemm(st, true, false, k.nbk, ncols, k.rows, 1.0, vcb, k.rows, csub, n, 0.0, w, nb, 0, gws, ngws)) return 1;
            if (dgemm(st, false, false, k.rows, ncols, k.nbk, -1.0, vtb, k.rows, w, nb, 1.0, csub, n, 0, gws, ngws)) return 1;
        }
    }
    JX_HIP(hipStreamSynchronize(st));   // work buffers are released on return
    if (trace)
        fprintf(stderr, "[jxgpu ormtr n=%d nb=%d oz=%d cols=%d] slice C %.1f | W = V'C %.1f | slice W + update %.1f ms (V, T, V T prepared "
                "ahead)\n", n, nb, (int)P.use_oz, ncols, tsum[0], tsum[1], tsum[2]);
    return 0;
}

// One call: prepare + apply on the same stream.
// General form: reflector j (j = 0 .. nref-1) = [0 (j + off rows); 1; A(j+off+1 : n, j)] with factor d_tau[j];
// d_c (n, ncols), ld = n (ncols < n: a rank's share of the eigenvector columns).
int ormtr_lower_off(hipStream_t st, const double *d_a, int n, int off, int nref, const double *d_tau, double *d_c, int ncols) {
    if (n < 2 || nref < 1 || ncols < 1) return 0;
    OrmtrPlan P;
    if (ormtr_prepare(st, d_a, n, off, nref, d_tau, P)) return 1;
    return ormtr_apply(st, P, d_c, ncols);
}

// d_a: (n,n) column-major after sytrd_lower; d_tau (n-1); d_c (n,n) column-major, overwritten with Q C.
int ormtr_lower(hipStream_t st, const double *d_a, int n, const double *d_tau, double *d_c) {
    return ormtr_lower_off(st, d_a, n, 1, n - 1, d_tau, d_c, n);
}

}  // namespace jx
