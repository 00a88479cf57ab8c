// Exact marker-space rrBLUP over the 2-bit payload (the route of the reference for m <= 15000 markers).
//
// Reference: `rrblup_exact_snp_packed` (src/stats/rrblup.rs:3179-3490): cache `build_rrblup_exact_snp_cache_from_source`
// (:1613-1899: A* = Z Z' over the training samples by DSYRK on the f32 design values cast to f64, minus the rank-one
// centring term, LAPACK eigendecomposition, positive spectrum capped at n_train - 1) and fit
// `fit_rrblup_exact_snp_from_cache_source` (:1951-2430: z = Z y_c, projection on the eigenbasis, Brent on the REML cost of
// the spectrum :1568-1611, beta = V diag(1 / (s + lambda)) V' z, intercept, predictions by `pcg_x_mul_samples`).
//
// Here: the design values are decoded once into an f64 image (m x n_train), A* is one TN product of the library's own
// f64-MFMA GEMM (k_dgemm.hip), the eigendecomposition is the library's own (eigh.cpp: two-stage reduction from m = 10000),
// the projections are two thin GEMMs on the eigenvector matrix in place, and the predictions stream the payload
// (`jxg_packed_dot`, k_gblup.hip).  Only the spectrum (<= n_train - 1 numbers) crosses to the host for Brent.
#include <math.h>

#include <algorithm>
#include <vector>

#include "jx_common.h"

namespace jx {

int dgemm(hipStream_t st, bool ta, bool tb, int m, int n, int k, double alpha, const double *a, int64_t lda,
          const double *b, int64_t ldb, double beta, double *c, int64_t ldc, int ksplit, double *ws, size_t ws_doubles);

// out[r][i] = lut[r][code(r, i)] as f64, r < m, i < n (row-major, n contiguous); P32: 32 bytes per SNP and 128-sample tile
__global__ __launch_bounds__(256) void rrx_decode_kernel(const uint8_t *__restrict__ p32, int64_t m, int n,
                                                         const float *__restrict__ lut, double *__restrict__ out) {
    const int64_t r = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int tile = i >> 7, within = i & 127;
    const uint8_t byte = p32[((int64_t)tile * m + r) * 32 + (within >> 2)];
    const int code = (byte >> (2 * (within & 3))) & 3;
    out[r * n + i] = (double)lut[4 * r + code];
}

// a[i + j m] -= rs[i] rs[j] * scale  (`symmetrize_upper_minus_rank1_in_place`, rrblup.rs:1549-1565)
__global__ __launch_bounds__(256) void rrx_center_kernel(double *__restrict__ a, int m, const double *__restrict__ rs,
                                                         double scale) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)m * m) return;
    const int i = (int)(e % m), j = (int)(e / m);
    a[e] -= scale * rs[i] * rs[j];
}

// src/math/brent.rs:1-136 (no initial point), objective evaluated on the host
template <typename F>
static void rrx_brent(F f, double low, double high, double tol_in, int max_iter, double &xb, double &fb) {
    double a = low, c = high;
    if (!(a < c)) std::swap(a, c);
    const double eps = 2.220446049250313e-16;
    const double tol = std::max(fabs(tol_in), 1e-12);
    double x = 0.5 * (a + c), w = x, v = x;
    double fx = f(x), fw = fx, fv = fx;
    double d = 0.0, e = 0.0;
    for (int it = 0; it < max_iter; ++it) {
        const double m = 0.5 * (a + c);
        const double tol1 = tol * fabs(x) + eps, tol2 = 2.0 * tol1;
        if (fabs(x - m) <= tol2 - 0.5 * (c - a)) break;
        double u;
        bool parabolic = false;
        if (fabs(e) > tol1) {
            double p = (x - v) * ((x - w) * (fx - fv)) - (x - w) * ((x - v) * (fx - fw));
            double q = 2.0 * (((x - v) * (fx - fw)) - ((x - w) * (fx - fv)));
            if (q > 0.0) p = -p;
            else q = -q;
            bool ok = false;
            if (fabs(q) > eps) {
                const double sstep = p / q;
                u = x + sstep;
                if ((u - a) >= tol2 && (c - u) >= tol2 && fabs(sstep) < 0.5 * fabs(e)) ok = true;
            }
            if (ok) {
                d = p / q;
                u = x + d;
                if ((u - a) < tol2 || (c - u) < tol2) d = (x < m) ? tol1 : -tol1;
                parabolic = true;
            }
        }
        if (!parabolic) {
            e = (x < m) ? (c - x) : (a - x);
            d = 0.3819660 * e;
        }
        if (fabs(d) < tol1) d = (d >= 0.0) ? tol1 : -tol1;
        u = x + d;
        const double fu = f(u);
        if (fu <= fx) {
            if (u >= x) a = x;
            else c = x;
            v = w; fv = fw;
            w = x; fw = fx;
            x = u; fx = fu;
        } else {
            if (u >= x) c = u;
            else a = u;
            if (fu <= fw || w == x) {
                v = w; fv = fw;
                w = u; fw = fu;
            } else if (fu <= fv || v == x || v == w) {
                v = u; fv = fu;
            }
        }
    }
    xb = x;
    fb = fx;
}

// rrblup_exact_reml_cost_from_spectrum (rrblup.rs:1568-1611)
static double rrx_cost(double lambda, const std::vector<double> &ev, const std::vector<double> &yp, double y_resid_ss,
                       int64_t n_eff) {
    if (!(isfinite(lambda) && lambda > 0.0)) return INFINITY;
    const size_t r = ev.size();
    if (r != yp.size() || n_eff == 0 || (size_t)n_eff < r) return INFINITY;
    double quad = 0.0, log_det = 0.0, ss = 0.0;
    for (size_t k = 0; k < r; ++k) {
        const double s = ev[k], yk = yp[k];
        if (!(isfinite(s) && s >= 0.0 && isfinite(yk))) return INFINITY;
        const double vk = s + lambda;
        if (!(isfinite(vk) && vk > 0.0)) return INFINITY;
        quad += (yk * yk) / vk;
        log_det += log(vk);
        ss += yk * yk;
    }
    const int64_t null_df = n_eff - (int64_t)r;
    const double null_ss = std::max(y_resid_ss - ss, 0.0);
    if (null_df > 0) {
        quad += null_ss / lambda;
        log_det += (double)null_df * log(lambda);
    }
    if (!(isfinite(quad) && quad > 0.0 && isfinite(log_det))) return INFINITY;
    return 0.5 * ((double)n_eff * log(quad) + log_det);
}

}  // namespace jx

using namespace jx;

// value_lut (eff_m, 4) f32: standardised design values by 2-bit code (entry 1, the missing code, must be 0).
// row_indices (eff_m) int64 or NULL (all m_total rows).  out_pred_train (n_train) or NULL; out_pred_test (n_test) or NULL.
// out_scalars: [0] pve_trainvar, [1] lambda, [2] REML (= -cost), [3] var_g, [4] sigma_e2, [5] rank, [6] intercept alpha,
// [7] y mean.
extern "C" int jx_rrblup_exact_snp_packed(const uint8_t *packed, int64_t m_total, int n_samples, const int64_t *row_indices,
                                          int64_t eff_m, const float *value_lut, const int64_t *train_idx, int n_train,
                                          const double *y_train, const int64_t *test_idx, int n_test,
                                          double log10_lambda_low, double log10_lambda_high, double reml_tol,
                                          int reml_max_iter, float *out_beta, double *out_pred_train,
                                          double *out_pred_test, double *out_scalars) {
    if (n_samples <= 0) return fail("n_samples must be > 0");
    if (m_total <= 0 || eff_m <= 0) return fail("rrblup_exact_snp_packed received zero active markers.");
    if (n_train <= 1) return fail("rrblup_exact_snp_packed requires at least two training samples.");
    if (!(isfinite(log10_lambda_low) && isfinite(log10_lambda_high)))
        return fail("rrblup_exact_snp_packed requires finite log10(lambda) bounds.");
    if (!(isfinite(reml_tol) && reml_tol > 0.0)) return fail("rrblup_exact_snp_packed requires finite reml_tol > 0.");
    if (reml_max_iter <= 0) return fail("rrblup_exact_snp_packed requires reml_max_iter > 0.");
    if (eff_m > 46000) return fail("rrblup_exact_snp_packed: too many markers for the exact marker-space route");
    const int m = (int)eff_m;
    const int64_t bps = ((int64_t)n_samples + 3) / 4;
    std::vector<int32_t> tr32(n_train), te32(n_test > 0 ? n_test : 0);
    for (int i = 0; i < n_train; ++i) {
        if (train_idx[i] < 0 || train_idx[i] >= n_samples) return fail("train_sample_indices out of range");
        tr32[i] = (int32_t)train_idx[i];
    }
    for (int i = 0; i < n_test; ++i) {
        if (test_idx[i] < 0 || test_idx[i] >= n_samples) return fail("test_sample_indices out of range");
        te32[i] = (int32_t)test_idx[i];
    }
    if (row_indices)
        for (int64_t j = 0; j < eff_m; ++j)
            if (row_indices[j] < 0 || row_indices[j] >= m_total) return fail("site_keep row index out of range");
    double y_mean = 0.0;
    for (int i = 0; i < n_train; ++i) {
        if (!isfinite(y_train[i])) return fail("y_train contains non-finite values.");
        y_mean += y_train[i];
    }
    y_mean /= (double)n_train;
    double y_center_ss = 0.0;
    for (int i = 0; i < n_train; ++i) y_center_ss += (y_train[i] - y_mean) * (y_train[i] - y_mean);

    hipStream_t st = nullptr;
    DevBuf raw, didx, drow, p32, dlut, dcnt;
    if (raw.alloc((size_t)(m_total * bps))) return 1;
    JX_HIP(hipMemcpy(raw.p, packed, (size_t)(m_total * bps), hipMemcpyHostToDevice));
    if (didx.alloc(sizeof(int32_t) * (size_t)n_train)) return 1;
    JX_HIP(hipMemcpy(didx.p, tr32.data(), sizeof(int32_t) * (size_t)n_train, hipMemcpyHostToDevice));
    const int64_t *d_rowidx = nullptr;
    if (row_indices) {
        if (drow.alloc(sizeof(int64_t) * (size_t)eff_m)) return 1;
        JX_HIP(hipMemcpy(drow.p, row_indices, sizeof(int64_t) * (size_t)eff_m, hipMemcpyHostToDevice));
        d_rowidx = drow.as<int64_t>();
    }
    const int nt = num_tiles(n_train);
    if (p32.alloc((size_t)nt * (size_t)eff_m * 32)) return 1;
    if (jxg_repack_p32(raw.as<uint8_t>(), bps, n_samples, m_total, didx.as<int32_t>(), n_train, d_rowidx, eff_m,
                       p32.as<uint8_t>(), st))
        return 1;
    if (dlut.alloc(sizeof(float) * 4 * (size_t)eff_m)) return 1;
    JX_HIP(hipMemcpy(dlut.p, value_lut, sizeof(float) * 4 * (size_t)eff_m, hipMemcpyHostToDevice));
    if (dcnt.alloc(sizeof(int32_t) * 3 * (size_t)eff_m)) return 1;
    if (jxg_row_counts_p32(p32.as<uint8_t>(), eff_m, n_train, dcnt.as<int32_t>(), st)) return 1;
    std::vector<int32_t> cnt(3 * (size_t)eff_m);
    JX_HIP(hipMemcpy(cnt.data(), dcnt.p, sizeof(int32_t) * 3 * (size_t)eff_m, hipMemcpyDeviceToHost));
    // row sums of the f32 design values in f64 (`row_major_block_row_sum_and_cast_f64`): count-weighted sums of the three
    // genotype values
    std::vector<double> rs(eff_m);
    std::vector<float> mu(eff_m);
    for (int64_t j = 0; j < eff_m; ++j) {
        const double c1 = cnt[3 * j + 1], c2 = cnt[3 * j + 2];
        const double c0 = (double)n_train - (double)cnt[3 * j] - c1 - c2;
        rs[j] = c0 * (double)value_lut[4 * j] + c1 * (double)value_lut[4 * j + 2] + c2 * (double)value_lut[4 * j + 3];
        mu[j] = (float)(rs[j] / (double)n_train);
    }
    const uint8_t *P = p32.as<uint8_t>();
    const float *L = dlut.as<float>();

    // A* = Z Z' - rs rs' / n_train
    DevBuf dz, da, dw, drs, dv64m, dv64n, dcoef;
    if (dz.alloc(sizeof(double) * (size_t)eff_m * (size_t)n_train)) return 1;
    if (da.alloc(sizeof(double) * (size_t)eff_m * (size_t)eff_m)) return 1;
    if (dw.alloc(sizeof(double) * (size_t)eff_m) || drs.alloc(sizeof(double) * (size_t)eff_m) ||
        dv64m.alloc(sizeof(double) * (size_t)eff_m) ||
        dv64n.alloc(sizeof(double) * (size_t)(n_train > n_test ? n_train : n_test)) ||
        dcoef.alloc(sizeof(double) * (size_t)eff_m))
        return 1;
    hipLaunchKernelGGL(rrx_decode_kernel, dim3((unsigned)((n_train + 255) / 256), (unsigned)eff_m), dim3(256), 0, st, P,
                       eff_m, n_train, L, dz.as<double>());
    JX_LAUNCH_CHECK();
    // the image is (n_train x m) column-major with ld = n_train: A = image' image
    if (dgemm(st, true, false, m, m, n_train, 1.0, dz.as<double>(), n_train, dz.as<double>(), n_train, 0.0, da.as<double>(),
              m, 1, nullptr, 0))
        return 1;
    JX_HIP(hipMemcpyAsync(drs.p, rs.data(), sizeof(double) * (size_t)eff_m, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rrx_center_kernel, dim3((unsigned)(((int64_t)m * m + 255) / 256)), dim3(256), 0, st, da.as<double>(), m,
                       drs.as<double>(), 1.0 / (double)n_train);
    JX_LAUNCH_CHECK();
    JX_HIP(hipStreamSynchronize(st));
    dz.release();
    if (jxg_eigh_f64(da.as<double>(), m, 0.0, dw.as<double>(), st)) return 1;     // da <- U' row-major = eigenvectors as columns
    std::vector<double> evals_all(eff_m);
    JX_HIP(hipMemcpy(evals_all.data(), dw.p, sizeof(double) * (size_t)eff_m, hipMemcpyDeviceToHost));
    const double max_eval = std::max(evals_all[eff_m - 1], 0.0);
    const double tol = 2.220446049250313e-16 * std::max(max_eval, 1.0) * (double)std::max<int64_t>(eff_m, 1);
    const int64_t n_eff = (int64_t)n_train - 1;
    int64_t keep_start = -1;
    for (int64_t j = 0; j < eff_m; ++j)
        if (evals_all[j] > tol) {
            keep_start = j;
            break;
        }
    if (keep_start < 0) return fail("rrblup_exact_snp_packed found no positive spectrum after centering.");
    if (eff_m - keep_start > n_eff) keep_start = eff_m - n_eff;
    const int rank = (int)(eff_m - keep_start);
    if (n_eff == 0 || n_eff < rank) return fail("rrblup_exact_snp_packed invalid effective df");
    std::vector<double> eigvals(evals_all.begin() + keep_start, evals_all.end());
    const double *vkeep = da.as<double>() + (size_t)keep_start * (size_t)eff_m;     // columns keep_start .. of V (ld = m)

    // z = Z y_c (f32 GEMV per sample block in the reference; f64 accumulation of the f32-rounded vector here)
    {
        std::vector<double> yc(n_train);
        for (int i = 0; i < n_train; ++i) yc[i] = (double)(float)(y_train[i] - y_mean);
        JX_HIP(hipMemcpy(dv64n.p, yc.data(), sizeof(double) * (size_t)n_train, hipMemcpyHostToDevice));
        if (jxg_packed_tdot(P, eff_m, n_train, nullptr, m, L, dv64n.as<double>(), dv64m.as<double>(), st)) return 1;
    }
    // coeff = V' z
    if (dgemm(st, true, false, rank, 1, m, 1.0, vkeep, m, dv64m.as<double>(), m, 0.0, dcoef.as<double>(), rank, 1, nullptr, 0))
        return 1;
    std::vector<double> coeff(rank), y_proj(rank), w(rank);
    JX_HIP(hipMemcpy(coeff.data(), dcoef.p, sizeof(double) * (size_t)rank, hipMemcpyDeviceToHost));
    for (int k = 0; k < rank; ++k) y_proj[k] = coeff[k] / sqrt(eigvals[k]);
    const double low = std::min(log10_lambda_low, log10_lambda_high), high = std::max(log10_lambda_low, log10_lambda_high);
    double best_log10 = 0.0, best_cost = 0.0;
    rrx_brent([&](double x) { return rrx_cost(pow(10.0, x), eigvals, y_proj, y_center_ss, n_eff); }, low, high, reml_tol,
              reml_max_iter, best_log10, best_cost);
    const double lambda_opt = std::max(pow(10.0, best_log10), 1e-12);
    double quad = 0.0, y_proj_ss = 0.0, g_center_ss = 0.0;
    for (int k = 0; k < rank; ++k) {
        const double yk = y_proj[k], denom = eigvals[k] + lambda_opt, ck = coeff[k];
        quad += (yk * yk) / denom;
        y_proj_ss += yk * yk;
        g_center_ss += eigvals[k] * (ck * ck) / (denom * denom);
        w[k] = ck / denom;
    }
    const double null_ss = std::max(y_center_ss - y_proj_ss, 0.0);
    if (n_eff - rank > 0) quad += null_ss / lambda_opt;
    const double sigma_beta2 = quad / (double)n_eff;
    const double sigma_e2 = lambda_opt * sigma_beta2;
    // beta = V w
    JX_HIP(hipMemcpy(dcoef.p, w.data(), sizeof(double) * (size_t)rank, hipMemcpyHostToDevice));
    if (dgemm(st, false, false, m, 1, rank, 1.0, vkeep, m, dcoef.as<double>(), rank, 0.0, dv64m.as<double>(), m, 1, nullptr, 0))
        return 1;
    std::vector<double> beta64(eff_m);
    JX_HIP(hipMemcpy(beta64.data(), dv64m.p, sizeof(double) * (size_t)eff_m, hipMemcpyDeviceToHost));
    double mean_dot = 0.0;
    for (int64_t j = 0; j < eff_m; ++j) {
        out_beta[j] = (float)beta64[j];
        beta64[j] = (double)out_beta[j];
        mean_dot += (double)mu[j] * beta64[j];
    }
    const double alpha_use = y_mean - mean_dot;
    const double var_g = g_center_ss / (double)(n_train - 1);
    const double den = var_g + sigma_e2;
    const double pve = (isfinite(den) && den > 0.0) ? var_g / den : NAN;

    // predictions: alpha + Z_samples' beta (pcg_x_mul_samples, f32 output widened)
    JX_HIP(hipMemcpy(dv64m.p, beta64.data(), sizeof(double) * (size_t)eff_m, hipMemcpyHostToDevice));
    if (out_pred_train) {
        if (jxg_packed_dot(P, eff_m, n_train, nullptr, m, L, dv64m.as<double>(), dv64n.as<double>(), st)) return 1;
        JX_HIP(hipStreamSynchronize(st));
        JX_HIP(hipMemcpy(out_pred_train, dv64n.p, sizeof(double) * (size_t)n_train, hipMemcpyDeviceToHost));
        for (int i = 0; i < n_train; ++i) out_pred_train[i] = (double)(float)out_pred_train[i] + alpha_use;
    }
    if (n_test > 0 && out_pred_test) {
        DevBuf dte, p32t;
        if (dte.alloc(sizeof(int32_t) * (size_t)n_test)) return 1;
        JX_HIP(hipMemcpy(dte.p, te32.data(), sizeof(int32_t) * (size_t)n_test, hipMemcpyHostToDevice));
        const int ntt = num_tiles(n_test);
        if (p32t.alloc((size_t)ntt * (size_t)eff_m * 32)) return 1;
        if (jxg_repack_p32(raw.as<uint8_t>(), bps, n_samples, m_total, dte.as<int32_t>(), n_test, d_rowidx, eff_m,
                           p32t.as<uint8_t>(), st))
            return 1;
        if (jxg_packed_dot(p32t.as<uint8_t>(), eff_m, n_test, nullptr, m, L, dv64m.as<double>(), dv64n.as<double>(), st))
            return 1;
        JX_HIP(hipStreamSynchronize(st));
        JX_HIP(hipMemcpy(out_pred_test, dv64n.p, sizeof(double) * (size_t)n_test, hipMemcpyDeviceToHost));
        for (int i = 0; i < n_test; ++i) out_pred_test[i] = (double)(float)out_pred_test[i] + alpha_use;
    }
    JX_HIP(hipStreamSynchronize(st));
    out_scalars[0] = pve;
    out_scalars[1] = lambda_opt;
    out_scalars[2] = -best_cost;
    out_scalars[3] = var_g;
    out_scalars[4] = sigma_e2;
    out_scalars[5] = (double)rank;
    out_scalars[6] = alpha_use;
    out_scalars[7] = y_mean;
    return 0;
}
