// Sparse symmetric solves on the device: (K + lambda I) X = B for many right-hand sides at once, K a thresholded GRM in CSR
// (full symmetric pattern), by Jacobi-preconditioned conjugate gradients.
//
// Where it is used: the SparseLMM routes (src/stats/splmm.rs) when the relatedness graph of the sparse GRM holds a connected
// component beyond one dense eigenproblem on the GPU (janusx._SparseFactorReml).  The reference factorises K + lambda I
// sparsely on the host for any structure (src/math/cholesky.rs:733, 1018-1183) and solves one system per SNP
// (`exact_scan_blocks_core`, src/stats/splmm.rs:2567-2880); here a block of decoded SNP rows is the right-hand side of ONE
// multi-vector CG: the matrix is streamed once per iteration for all of them.  K + lambda I of a thresholded GRM is well
// conditioned at the lambdas of a REML optimum ((s_max + lambda) / (s_min + lambda), a few units), so 15 - 40 iterations reach
// 1e-11.
//
// Layout: all vectors are (n, ldr) row-major f64 -- row i holds entry i of every right-hand side, ldr = the number of
// right-hand sides rounded up to 64 -- so that the gather  y[i, :] = sum_k val[k] x[col[k], :]  reads whole contiguous rows:
// HBM-bound, 8 ldr bytes per non-zero.  Sums over i (p'Ap, r'z, r'r per right-hand side) are two-stage and in a fixed order:
// per chunk of SPS_ROWS rows, then over the chunks -- bit-reproducible.
#include <cmath>
#include <string>
#include <vector>

#include "jx_common.h"

namespace jx {

constexpr int SPS_ROWS = 128;      // rows per workgroup of the streaming kernels
constexpr int SPS_T = 256;         // threads per workgroup = right-hand sides per column slab

// d_out (n, ldr) f64 <- rows (nrhs, ld) f32 transposed; columns nrhs .. ldr-1 are zero
__global__ __launch_bounds__(256) void sps_rows_to_cols_kernel(const float *__restrict__ rows, int nrhs, int n, int64_t ld,
                                                               double *__restrict__ out, int ldr) {
    __shared__ float tile[64][65];
    const int i0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) {
        const int r = r0 + k, i = i0 + tx;
        tile[k][tx] = (r < nrhs && i < n) ? rows[(int64_t)r * ld + i] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        const int i = i0 + k, r = r0 + tx;
        if (i < n && r < ldr) out[(int64_t)i * ldr + r] = (double)tile[tx][k];
    }
}

// ap = (K + lambda I) p for a chunk of rows; part[chunk][r] = sum over the chunk's rows of p[i, r] ap[i, r]
__global__ __launch_bounds__(SPS_T) void sps_spmm_kernel(int n, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                         const double *__restrict__ val, double lambda,
                                                         const double *__restrict__ p, int ldr, double *__restrict__ ap,
                                                         double *__restrict__ part) {
    const int r = blockIdx.y * SPS_T + threadIdx.x;
    const int i0 = blockIdx.x * SPS_ROWS;
    const int i1 = min(i0 + SPS_ROWS, n);
    double acc = 0.0;
    if (r < ldr) {
        for (int i = i0; i < i1; ++i) {
            const double pi = p[(int64_t)i * ldr + r];
            double s = lambda * pi;
            const int64_t k1 = rowptr[i + 1];
            for (int64_t k = rowptr[i]; k < k1; ++k) s = fma(val[k], p[(int64_t)col[k] * ldr + r], s);
            ap[(int64_t)i * ldr + r] = s;
            acc = fma(pi, s, acc);
        }
        part[(int64_t)blockIdx.x * ldr + r] = acc;
    }
}

// out[r] = sum over the chunks (in index order) of part[chunk][r]; nq quantities interleaved as part[(q * nchunks + chunk) * ldr + r]
__global__ __launch_bounds__(SPS_T) void sps_reduce_kernel(const double *__restrict__ part, int nchunks, int ldr, int nq,
                                                           double *__restrict__ out) {
    const int r = blockIdx.x * SPS_T + threadIdx.x;
    if (r >= ldr) return;
    for (int q = 0; q < nq; ++q) {
        double s = 0.0;
        for (int c = 0; c < nchunks; ++c) s += part[((int64_t)q * nchunks + c) * ldr + r];
        out[(int64_t)q * ldr + r] = s;
    }
}

// sc (7, ldr): [0] rz  [1] pap  [2] rz_new  [3] rr  [4] alpha  [5] beta  [6] bnorm2
__global__ __launch_bounds__(SPS_T) void sps_alpha_kernel(double *__restrict__ sc, int ldr) {
    const int r = blockIdx.x * SPS_T + threadIdx.x;
    if (r >= ldr) return;
    const double rz = sc[r], pap = sc[ldr + r];
    sc[4 * ldr + r] = (pap > 0.0 && isfinite(pap) && rz != 0.0) ? rz / pap : 0.0;      // a converged / zero column stays where it is
}
__global__ __launch_bounds__(SPS_T) void sps_beta_kernel(double *__restrict__ sc, int ldr) {
    const int r = blockIdx.x * SPS_T + threadIdx.x;
    if (r >= ldr) return;
    const double rz = sc[r], rzn = sc[2 * ldr + r];
    sc[5 * ldr + r] = (rz != 0.0 && isfinite(rzn)) ? rzn / rz : 0.0;
    sc[r] = rzn;
}

// x += alpha p; r -= alpha ap; z = dinv r; partial sums of r z and r r
__global__ __launch_bounds__(SPS_T) void sps_update_kernel(int n, const double *__restrict__ dinv, const double *__restrict__ sc,
                                                           const double *__restrict__ p, const double *__restrict__ ap,
                                                           double *__restrict__ x, double *__restrict__ res,
                                                           double *__restrict__ z, int ldr, int nchunks,
                                                           double *__restrict__ part) {
    const int r = blockIdx.y * SPS_T + threadIdx.x;
    if (r >= ldr) return;
    const int i0 = blockIdx.x * SPS_ROWS;
    const int i1 = min(i0 + SPS_ROWS, n);
    const double alpha = sc[4 * ldr + r];
    double a_rz = 0.0, a_rr = 0.0;
    for (int i = i0; i < i1; ++i) {
        const int64_t o = (int64_t)i * ldr + r;
        x[o] = fma(alpha, p[o], x[o]);
        const double rv = fma(-alpha, ap[o], res[o]);
        res[o] = rv;
        const double zv = dinv[i] * rv;
        z[o] = zv;
        a_rz = fma(rv, zv, a_rz);
        a_rr = fma(rv, rv, a_rr);
    }
    part[(int64_t)blockIdx.x * ldr + r] = a_rz;
    part[((int64_t)nchunks + blockIdx.x) * ldr + r] = a_rr;
}

// start: x = 0, res = b, z = dinv b, p = z; partial sums of r z and r r (= |b|^2)
__global__ __launch_bounds__(SPS_T) void sps_start_kernel(int n, const double *__restrict__ dinv, const double *__restrict__ b,
                                                          double *__restrict__ x, double *__restrict__ res, double *__restrict__ z,
                                                          double *__restrict__ p, int ldr, int nchunks, double *__restrict__ part) {
    const int r = blockIdx.y * SPS_T + threadIdx.x;
    if (r >= ldr) return;
    const int i0 = blockIdx.x * SPS_ROWS;
    const int i1 = min(i0 + SPS_ROWS, n);
    double a_rz = 0.0, a_rr = 0.0;
    for (int i = i0; i < i1; ++i) {
        const int64_t o = (int64_t)i * ldr + r;
        const double rv = b[o];
        const double zv = dinv[i] * rv;
        x[o] = 0.0;
        res[o] = rv;
        z[o] = zv;
        p[o] = zv;
        a_rz = fma(rv, zv, a_rz);
        a_rr = fma(rv, rv, a_rr);
    }
    part[(int64_t)blockIdx.x * ldr + r] = a_rz;
    part[((int64_t)nchunks + blockIdx.x) * ldr + r] = a_rr;
}

// p = z + beta p
__global__ __launch_bounds__(SPS_T) void sps_dir_kernel(int n, const double *__restrict__ sc, const double *__restrict__ z,
                                                        double *__restrict__ p, int ldr) {
    const int r = blockIdx.y * SPS_T + threadIdx.x;
    if (r >= ldr) return;
    const int i0 = blockIdx.x * SPS_ROWS;
    const int i1 = min(i0 + SPS_ROWS, n);
    const double beta = sc[5 * ldr + r];
    for (int i = i0; i < i1; ++i) {
        const int64_t o = (int64_t)i * ldr + r;
        p[o] = fma(beta, p[o], z[o]);
    }
}

// per right-hand side r: sums[r][0] = g.z, [1] = g.py, [2 + k] = g.vx[:, k]  (partial per chunk, then reduced in chunk order)
__global__ __launch_bounds__(SPS_T) void sps_scan_sums_kernel(int n, const double *__restrict__ g, const double *__restrict__ z,
                                                              const double *__restrict__ py, const double *__restrict__ vx, int p,
                                                              int ldr, int nchunks, double *__restrict__ part) {
    const int r = blockIdx.y * SPS_T + threadIdx.x;
    if (r >= ldr) return;
    const int i0 = blockIdx.x * SPS_ROWS;
    const int i1 = min(i0 + SPS_ROWS, n);
    double acc[2 + JXG_MAX_COV];
    for (int k = 0; k < 2 + p; ++k) acc[k] = 0.0;
    for (int i = i0; i < i1; ++i) {
        const int64_t o = (int64_t)i * ldr + r;
        const double gv = g[o];
        acc[0] = fma(gv, z[o], acc[0]);
        acc[1] = fma(gv, py[i], acc[1]);
        for (int k = 0; k < p; ++k) acc[2 + k] = fma(gv, vx[(int64_t)i * p + k], acc[2 + k]);
    }
    for (int k = 0; k < 2 + p; ++k) part[((int64_t)k * nchunks + blockIdx.x) * ldr + r] = acc[k];
}

// sums (nrhs, p + 2) <- red (p + 2, ldr)
__global__ __launch_bounds__(SPS_T) void sps_sums_pack_kernel(const double *__restrict__ red, int ldr, int nrhs, int nq,
                                                              double *__restrict__ sums) {
    const int r = blockIdx.x * SPS_T + threadIdx.x;
    if (r >= nrhs) return;
    for (int q = 0; q < nq; ++q) sums[(int64_t)r * nq + q] = red[(int64_t)q * ldr + r];
}

}  // namespace jx

using namespace jx;

static inline int sps_chunks(int n) { return (n + SPS_ROWS - 1) / SPS_ROWS; }

extern "C" int jxg_sps_ldr(int nrhs) { return (nrhs + 63) / 64 * 64; }

// doubles of workspace for jxg_sps_solve_multi / jxg_sps_scan_sums: res, z, p, ap (n ldr each) + partials + scalars
extern "C" int64_t jxg_sps_work_doubles(int n, int ldr) {
    const int64_t nq = 2 + JXG_MAX_COV;
    return 4 * (int64_t)n * ldr + nq * (int64_t)sps_chunks(n) * ldr + (7 + nq) * (int64_t)ldr;
}

extern "C" int jxg_sps_rows_to_cols_f64(const float *d_rows, int nrhs, int n, int64_t ld, double *d_out, int ldr, void *stream) {
    if (nrhs <= 0 || n <= 0) return 0;
    if (ldr < nrhs) return fail("jxg_sps_rows_to_cols_f64: ldr < nrhs");
    hipLaunchKernelGGL(sps_rows_to_cols_kernel, dim3((n + 63) / 64, (ldr + 63) / 64), dim3(256), 0, (hipStream_t)stream, d_rows,
                       nrhs, n, ld, d_out, ldr);
    JX_LAUNCH_CHECK();
    return 0;
}

// (K + lambda I) X = B, nrhs right-hand sides in the (n, ldr) layout; d_dinv[i] = 1 / (K_ii + lambda).  Stops when every
// right-hand side has |r| <= tol |b| or after max_iter iterations (checked every 4); h_info: [iterations, max |r| / |b|].
extern "C" int jxg_sps_solve_multi(int n, const int64_t *d_rowptr, const int32_t *d_col, const double *d_val, double lambda,
                                   const double *d_dinv, const double *d_b, int nrhs, int ldr, double tol, int max_iter,
                                   double *d_x, double *d_work, double *h_info, void *stream) {
    if (n <= 0 || nrhs <= 0) return 0;
    if (ldr < nrhs || (ldr & 63)) return fail("jxg_sps_solve_multi: ldr must be a multiple of 64 and >= nrhs");
    if (!(std::isfinite(lambda) && lambda >= 0.0)) return fail("jxg_sps_solve_multi: lambda must be finite and >= 0");
    if (!(std::isfinite(tol) && tol > 0.0) || max_iter <= 0) return fail("jxg_sps_solve_multi: tol / max_iter out of range");
    hipStream_t st = (hipStream_t)stream;
    const int nch = sps_chunks(n);
    double *res = d_work, *z = res + (int64_t)n * ldr, *p = z + (int64_t)n * ldr, *ap = p + (int64_t)n * ldr;
    double *part = ap + (int64_t)n * ldr;
    double *sc = part + (int64_t)(2 + JXG_MAX_COV) * nch * ldr;
    const dim3 grid(nch, (ldr + SPS_T - 1) / SPS_T), blk(SPS_T);
    const dim3 g1((ldr + SPS_T - 1) / SPS_T);
    hipLaunchKernelGGL(sps_start_kernel, grid, blk, 0, st, n, d_dinv, d_b, d_x, res, z, p, ldr, nch, part);
    JX_LAUNCH_CHECK();
    // rz -> sc[0], rr -> sc[1] (scratch), keep |b|^2 in sc[6]
    hipLaunchKernelGGL(sps_reduce_kernel, g1, blk, 0, st, part, nch, ldr, 2, sc);
    JX_LAUNCH_CHECK();
    JX_HIP(hipMemcpyAsync(sc + 6 * (int64_t)ldr, sc + (int64_t)ldr, sizeof(double) * (size_t)ldr, hipMemcpyDeviceToDevice, st));
    std::vector<double> h(2 * (size_t)ldr);
    int it = 0;
    double worst = 0.0;
    for (; it < max_iter;) {
        hipLaunchKernelGGL(sps_spmm_kernel, grid, blk, 0, st, n, d_rowptr, d_col, d_val, lambda, p, ldr, ap, part);
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sps_reduce_kernel, g1, blk, 0, st, part, nch, ldr, 1, sc + (int64_t)ldr);      // pap
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sps_alpha_kernel, g1, blk, 0, st, sc, ldr);
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sps_update_kernel, grid, blk, 0, st, n, d_dinv, sc, p, ap, d_x, res, z, ldr, nch, part);
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sps_reduce_kernel, g1, blk, 0, st, part, nch, ldr, 2, sc + 2 * (int64_t)ldr);  // rz_new, rr
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sps_beta_kernel, g1, blk, 0, st, sc, ldr);
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(sps_dir_kernel, grid, blk, 0, st, n, sc, z, p, ldr);
        JX_LAUNCH_CHECK();
        ++it;
        if ((it & 3) == 0 || it == max_iter) {
            JX_HIP(hipMemcpyAsync(h.data(), sc + 3 * (int64_t)ldr, sizeof(double) * (size_t)ldr, hipMemcpyDeviceToHost, st));
            JX_HIP(hipMemcpyAsync(h.data() + ldr, sc + 6 * (int64_t)ldr, sizeof(double) * (size_t)ldr, hipMemcpyDeviceToHost, st));
            JX_HIP(hipStreamSynchronize(st));
            worst = 0.0;
            for (int r = 0; r < nrhs; ++r) {
                const double bb = h[(size_t)ldr + r], rr = h[r];
                if (!std::isfinite(rr)) return fail("jxg_sps_solve_multi: the iteration broke down (non-finite residual)");
                if (bb > 0.0) worst = fmax(worst, sqrt(rr / bb));
            }
            if (worst <= tol) break;
        }
    }
    if (h_info) {
        h_info[0] = (double)it;
        h_info[1] = worst;
    }
    if (worst > tol)
        return fail("jxg_sps_solve_multi: conjugate gradients did not reach tol = " + std::to_string(tol) + " in " +
                    std::to_string(max_iter) + " iterations (max |r| / |b| = " + std::to_string(worst) + ")");
    return 0;
}

// SparseLMM exact-scan sums of a block: d_g (decoded rows) and d_z = (K + lambda I)^-1 g in the (n, ldr) layout ->
// d_sums (nrhs, p + 2) = [g'V^-1 g, g.Py, g.(V^-1 X)[:, k]]: the layout jxg_fvlmm_finish_dev takes with ntiles = 1.
extern "C" int jxg_sps_scan_sums(int n, const double *d_g, const double *d_z, int nrhs, int ldr, const double *d_py,
                                 const double *d_vinvx, int p, double *d_sums, double *d_work, void *stream) {
    if (n <= 0 || nrhs <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_sps_scan_sums: p out of range");
    hipStream_t st = (hipStream_t)stream;
    const int nch = sps_chunks(n);
    double *part = d_work + 4 * (int64_t)n * ldr;
    double *red = part + (int64_t)(2 + JXG_MAX_COV) * nch * ldr + 7 * (int64_t)ldr;
    const dim3 grid(nch, (ldr + SPS_T - 1) / SPS_T), blk(SPS_T);
    hipLaunchKernelGGL(sps_scan_sums_kernel, grid, blk, 0, st, n, d_g, d_z, d_py, d_vinvx, p, ldr, nch, part);
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(sps_reduce_kernel, dim3((ldr + SPS_T - 1) / SPS_T), blk, 0, st, part, nch, ldr, p + 2, red);
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(sps_sums_pack_kernel, dim3((nrhs + SPS_T - 1) / SPS_T), blk, 0, st, red, ldr, nrhs, p + 2, d_sums);
    JX_LAUNCH_CHECK();
    return 0;
}
