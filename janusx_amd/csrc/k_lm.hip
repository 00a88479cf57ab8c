// Plain linear-model scan (the LM route `jx gwas -lmm / -fvlmm` falls back to when the null likelihood-ratio test of
// src/stats/gwas_unified.rs:121-175 finds no polygenic variance): `lm_block_assoc_packed`, src/stats/glm.rs:3550-3860.
//   r_y = M_X y (host, f64),  U = G X (f32 operands),  a = G r_y,  d = rowsum(G o G),  s = d - u' C u,  C = (X'X)^-1
//   beta = a / s,  rss = max(y'M_X y - beta a, 0),  ve = rss / df,  se = sqrt(ve / s),  t = beta / se,
//   pwald = two-sided Student t (betai / betacf, glm.rs:383-481),  plrt = chi2_1 sf of n ln(1 + t^2 / df) (glm.rs:483-500).
// G is never decoded to memory: the value of (SNP r, sample i) is lut[r][code], read from the P32 image (p32[tile][snp]
// [32 B], 128 samples per record); thread = SNP, so a wave's record reads are contiguous, the workgroup walks the sample
// tiles with the tile's columns of [X | r_y] (rounded to f32 as the reference does before its sgemm, accumulated in f64
// here) broadcast from LDS.  HBM-bound on paper (n / 4 bytes per SNP), in practice bound by the f64 FMA issue: NC + 1
// FMAs per genotype.
#include <cmath>
#include <vector>

#include "jx_common.h"

namespace jx {

constexpr int LM_THREADS = 128;
constexpr int LM_MAXC = 4;          // columns of [X | r_y] per pass

// sums[r][c0 + c] = sum_i v(r,i) xr[i][c0 + c] (c < NC) and, when dsum != nullptr, dsum[r] = sum_i v(r,i)^2.
// xr: (n, ldx) f64 row-major on the device.
template <int NC>
__global__ __launch_bounds__(LM_THREADS) void lm_dots_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                             const int32_t *__restrict__ rows, int nrows,
                                                             const float *__restrict__ lut,
                                                             const double *__restrict__ xr, int ldx, int c0, int n,
                                                             double *__restrict__ sums, int lds,
                                                             double *__restrict__ dsum) {
    __shared__ double a_sh[128 * NC];
    __shared__ double one_sh[128];
    const int tid = threadIdx.x;
    const int r = blockIdx.x * LM_THREADS + tid;
    const bool live = r < nrows;
    const int64_t rec = live ? (rows ? (int64_t)rows[r] : (int64_t)r) : 0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    if (live) {
        const float4 l = *reinterpret_cast<const float4 *>(lut + (int64_t)r * 4);
        v0 = (double)l.x, v1 = (double)l.y, v2 = (double)l.z, v3 = (double)l.w;
    }
    double acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    double dd = 0.0;
    const int ntiles = (n + 127) / 128;
    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();
        {
            const int i = tile * 128 + tid;
#pragma unroll
            for (int c = 0; c < NC; ++c) a_sh[tid * NC + c] = (i < n) ? xr[(int64_t)i * ldx + c0 + c] : 0.0;
            one_sh[tid] = (i < n) ? 1.0 : 0.0;
        }
        __syncthreads();
        if (!live) continue;
        const uint4 *p = reinterpret_cast<const uint4 *>(p32 + ((int64_t)tile * m_total + rec) * 32);
        const uint4 w0 = p[0], w1 = p[1];
        const uint32_t words[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t word = words[w];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t code = word & 3u;
                word >>= 2;
                const double lo = (code & 1u) ? v1 : v0;
                const double hi = (code & 1u) ? v3 : v2;
                const double v = (code & 2u) ? hi : lo;
                const int i = w * 16 + k;
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = fma(v, a_sh[i * NC + c], acc[c]);
                dd = fma(v * one_sh[i], v, dd);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int c = 0; c < NC; ++c) sums[(int64_t)r * lds + c0 + c] = acc[c];
    if (dsum) dsum[r] = dd;
}

// Dense rows (`lm_block_assoc_f32`, src/stats/glm.rs:4313-4497: an already decoded SNP-major f32 block): one wave per LM_DR
// rows, lanes over the samples (coalesced row reads), the columns of [X | r_y] read once per sample and lane for the LM_DR rows.
constexpr int LM_DR = 4;
template <int NC>
__global__ __launch_bounds__(256) void lm_dots_dense_kernel(const float *__restrict__ g, int64_t ld, int nrows,
                                                            const double *__restrict__ xr, int ldx, int c0, int n,
                                                            double *__restrict__ sums, int lds, double *__restrict__ dsum) {
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LM_DR;
    if (r0 >= nrows) return;
    double acc[LM_DR][NC], dd[LM_DR];
#pragma unroll
    for (int j = 0; j < LM_DR; ++j) {
        dd[j] = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[j][c] = 0.0;
    }
    for (int i = lane; i < n; i += 64) {
        double xv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) xv[c] = xr[(int64_t)i * ldx + c0 + c];
#pragma unroll
        for (int j = 0; j < LM_DR; ++j) {
            const double v = (r0 + j < nrows) ? (double)g[(int64_t)(r0 + j) * ld + i] : 0.0;
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[j][c] = fma(v, xv[c], acc[j][c]);
            dd[j] = fma(v, v, dd[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < LM_DR; ++j) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            dd[j] += __shfl_xor(dd[j], o);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[j][c] += __shfl_xor(acc[j][c], o);
        }
        if (lane == 0 && r0 + j < nrows) {
#pragma unroll
            for (int c = 0; c < NC; ++c) sums[(int64_t)(r0 + j) * lds + c0 + c] = acc[j][c];
            if (dsum) dsum[r0 + j] = dd[j];
        }
    }
}

// ---- p-values (glm.rs:383-500) ------------------------------------------------------------------------------------
__device__ double lm_betacf(double a, double b, double x) {
    const int maxit = 200;
    const double eps = 3.0e-14, fpmin = 1.0e-300;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0;
    double d = 1.0 - qab * x / qap;
    if (fabs(d) < fpmin) d = fpmin;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= maxit; ++m) {
        const double fm = (double)m, m2 = 2.0 * fm;
        double aa = fm * (b - fm) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c;
        if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + fm) * (qab + fm) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d;
        if (fabs(d) < fpmin) d = fpmin;
        c = 1.0 + aa / c;
        if (fabs(c) < fpmin) c = fpmin;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < eps) break;
    }
    return h;
}

__device__ double lm_betai(double a, double b, double x, double ln_beta) {
    if (!(x >= 0.0 && x <= 1.0)) return NAN;
    if (x == 0.0) return 0.0;
    if (x == 1.0) return 1.0;
    if (x < (a + 1.0) / (a + b + 2.0)) {
        const double front = exp(a * log(x) + b * log(1.0 - x) - ln_beta) / a;
        return front * lm_betacf(a, b, x);
    }
    const double front = exp(b * log(1.0 - x) + a * log(x) - ln_beta) / b;
    return 1.0 - front * lm_betacf(b, a, 1.0 - x);
}

constexpr double LM_MIN_POS = 2.2250738585072014e-308;

__device__ double lm_student_t_two_sided(double t, int df, double ln_beta) {
    if (df <= 0) return NAN;
    if (!isfinite(t)) return isnan(t) ? NAN : LM_MIN_POS;
    const double v = (double)df;
    double p = lm_betai(0.5 * v, 0.5, v / (v + t * t), ln_beta);
    if (!isfinite(p)) p = 1.0;
    return fmin(fmax(p, LM_MIN_POS), 1.0);
}

__device__ double lm_chi2_sf_df1(double stat) {
    if (!isfinite(stat) || stat < 0.0) return NAN;
    const double p = erfc(sqrt(0.5 * stat));
    if (!isfinite(p)) return 1.0;
    return fmin(fmax(p, LM_MIN_POS), 1.0);
}

// out[r] = (beta, se, pwald, plrt).  sums (nrows, lds): columns 0 .. q0-1 = u, column q0 = a.  ixx (q0, q0) row-major.
__global__ __launch_bounds__(256) void lm_stats_kernel(const double *__restrict__ sums, int lds,
                                                       const double *__restrict__ dsum, int nrows, int q0,
                                                       const double *__restrict__ ixx, double yy_r, int n_obs, int df,
                                                       double ln_beta, double *__restrict__ out, int dense_rules) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const double *u = sums + (int64_t)r * lds;
    double corr = 0.0;
    for (int k = 0; k < q0; ++k) {
        double acc = 0.0;
        for (int t = 0; t < q0; ++t) acc += ixx[k * q0 + t] * u[t];
        corr += u[k] * acc;
    }
    const double s = dsum[r] - corr;
    const double a = u[q0];
    double beta = NAN, se = NAN, pw = NAN, pl = NAN;
    // dense_rules: the row conditions of the dense-block entry point (glm.rs:4457-4479: s must exceed 1e-12, a non-finite
    // variance or standard error voids the row) instead of those of the packed one (glm.rs:3790-3830)
    if (dense_rules ? (isfinite(s) && s > 1e-12) : (!(s < 1e-12) && isfinite(s))) {
        beta = a / s;
        const double rss = fmax(yy_r - beta * a, 0.0);
        const double ve = rss / (double)df;
        if (dense_rules ? (isfinite(ve) && ve > 0.0) : (ve > 0.0)) {
            se = sqrt(ve / s);
            if (dense_rules && !(isfinite(beta) && isfinite(se) && se > 0.0)) {
                beta = se = NAN;
            } else {
                const double t = beta / se;
                const double t2 = t * t;
                pw = lm_student_t_two_sided(t, df, ln_beta);
                pl = (df <= 0 || !isfinite(t2) || t2 < 0.0) ? NAN
                                                             : lm_chi2_sf_df1((double)n_obs * log(1.0 + t2 / (double)df));
            }
        }
    }
    double *o = out + (int64_t)r * 4;
    o[0] = beta, o[1] = se, o[2] = pw, o[3] = pl;
}

// ---- SparseLMM approximate (GRAMMAR-gamma) scan: `grammar_scan_blocks_core`, additive model (src/stats/splmm.rs:2935-3316) ----
// Same streaming dots as the LM scan (lm_dots_kernel: [X | score] columns rounded to f32 by the caller like
// `pack_score_design_rhs_f32`, :1732-1757; the f32 sgemm outputs of the reference are reproduced by rounding the f64 sums to
// f32), then per SNP  g'Mg = max(g'g - (X'g)'(X'X)^-1(X'g), 0)  (rows whose residual sum of squares is effectively zero are
// (NaN, NaN, 1), :1710-1717),  denominator = r_hat g'Mg  and `splmm_wald_from_score_denom` (:2517-2538).
__device__ double sp_chi2_sf_df1(double stat) {        // src/math/linalg.rs:7-17
    if (!isfinite(stat) || stat <= 0.0) return 1.0;
    const double p = erfc(sqrt(0.5 * stat));
    if (!isfinite(p)) return 1.0;
    return fmin(fmax(p, LM_MIN_POS), 1.0);
}

__global__ __launch_bounds__(256) void splmm_grammar_stats_kernel(const double *__restrict__ sums, int lds,
                                                                  const double *__restrict__ dsum, int nrows, int p,
                                                                  const double *__restrict__ ixx, double score_scale,
                                                                  double denom_scale, double sigma2,
                                                                  double *__restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const double *u = sums + (int64_t)r * lds;
    double quad = 0.0;
    for (int k = 0; k < p; ++k) {
        double acc = 0.0;
        for (int t = 0; t < p; ++t) acc += ixx[k * p + t] * (double)(float)u[t];
        quad += (double)(float)u[k] * acc;
    }
    const double s_sq = dsum[r];
    const double s_m_s = fmax(s_sq - quad, 0.0);
    double beta = NAN, se = NAN, pw = 1.0;
    const bool zero = !(isfinite(s_m_s) && isfinite(s_sq)) || s_m_s <= fmax(1e-10, 1e-12 * fmax(fabs(s_sq), 1.0));
    if (!zero) {
        const double score = score_scale * (double)(float)u[p];
        const double denom = denom_scale * s_m_s;
        if (isfinite(score) && isfinite(denom) && denom > 1e-30 && isfinite(sigma2) && sigma2 > 0.0) {
            const double b = score / denom, vb = sigma2 / denom;
            if (isfinite(b) && isfinite(vb) && vb > 0.0) {
                const double s = sqrt(vb), chisq = (score * score) / (sigma2 * denom);
                if (isfinite(s) && s > 0.0 && isfinite(chisq) && chisq >= 0.0) beta = b, se = s, pw = sp_chi2_sf_df1(chisq);
            }
        }
    }
    double *o = out + (int64_t)r * 3;
    o[0] = beta, o[1] = se, o[2] = pw;
}

// ---- gamma of the approximate route: sums over a ROTATED marker row g~ = U'g (src/stats/splmm_approx.rs:921-1068) -------------
// With V^-1 = U diag(w) U', g_r = M_X g and c = (X'X)^-1 X'g every quantity of `estimate_gamma_from_markers` is a combination
// of   g~'g~,  g~'W g~,  g~'a~,  X~'g~ (p),  X~'W g~ (p)   (a~ = U'a):  g_r'g_r = g~'g~ - (X~'g~)'c,
// g_r'V^-1 g_r = g~'Wg~ - 2 c'(X~'Wg~) + c'(X~'WX~)c,  g_r'a = g~'a~ - c'(X~'a~).  One workgroup per sampled marker.
constexpr int SG_MAXP = 16;
__global__ __launch_bounds__(256) void splmm_gamma_sums_kernel(const float *__restrict__ grot, int64_t ld, int n, int p,
                                                               const double *__restrict__ w, const double *__restrict__ a,
                                                               const double *__restrict__ x, double *__restrict__ out) {
    __shared__ double red[256];
    const float *g = grot + (int64_t)blockIdx.x * ld;
    const int tid = threadIdx.x;
    double acc[3 + 2 * SG_MAXP];
    for (int k = 0; k < 3 + 2 * p; ++k) acc[k] = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double gv = (double)g[i], wi = w[i], wg = wi * gv;
        acc[0] = fma(gv, gv, acc[0]);
        acc[1] = fma(wg, gv, acc[1]);
        acc[2] = fma(gv, a[i], acc[2]);
        for (int k = 0; k < p; ++k) {
            const double xv = x[(int64_t)i * p + k];
            acc[3 + k] = fma(gv, xv, acc[3 + k]);
            acc[3 + p + k] = fma(wg, xv, acc[3 + p + k]);
        }
    }
    for (int k = 0; k < 3 + 2 * p; ++k) {
        red[tid] = acc[k];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) out[(int64_t)blockIdx.x * (3 + 2 * p) + k] = red[0];
        __syncthreads();
    }
}

}  // namespace jx

using namespace jx;

// LM scan of `nrows` SNPs of a resident P32 image.  d_lut (nrows, 4) f32 = decoded value by 2-bit code; d_xr (n, q0 + 1)
// f64 = [X | r_y] (the caller rounds to f32 first where it mirrors the reference); d_ixx (q0, q0) f64; d_work
// (nrows * (q0 + 2)) f64 scratch; d_out (nrows, 4) f64 = beta, se, pwald, plrt.
extern "C" int jxg_lm_scan_p32(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                               const float *d_lut, const double *d_xr, int q0, const double *d_ixx, double yy_r,
                               double *d_work, double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    if (q0 < 0 || n <= q0 + 1) return fail("n too small: require n > q0+1");
    hipStream_t st = (hipStream_t)stream;
    const int ncols = q0 + 1;
    double *sums = d_work, *dsum = d_work + (size_t)nrows * ncols;
    const dim3 grid((nrows + LM_THREADS - 1) / LM_THREADS), block(LM_THREADS);
    for (int c0 = 0; c0 < ncols; c0 += LM_MAXC) {
        const int nc = std::min(LM_MAXC, ncols - c0);
        double *dd = c0 == 0 ? dsum : nullptr;
        switch (nc) {
        case 1: hipLaunchKernelGGL(lm_dots_kernel<1>, grid, block, 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_xr, ncols, c0, n, sums, ncols, dd); break;
        case 2: hipLaunchKernelGGL(lm_dots_kernel<2>, grid, block, 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_xr, ncols, c0, n, sums, ncols, dd); break;
        case 3: hipLaunchKernelGGL(lm_dots_kernel<3>, grid, block, 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_xr, ncols, c0, n, sums, ncols, dd); break;
        default: hipLaunchKernelGGL(lm_dots_kernel<4>, grid, block, 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_xr, ncols, c0, n, sums, ncols, dd); break;
        }
        JX_LAUNCH_CHECK();
    }
    const int df = n - q0 - 1;
    const double ln_beta = lgamma(0.5 * df) + lgamma(0.5) - lgamma(0.5 * df + 0.5);
    hipLaunchKernelGGL(lm_stats_kernel, dim3((nrows + 255) / 256), dim3(256), 0, st, sums, ncols, dsum, nrows, q0, d_ixx,
                       yy_r, n, df, ln_beta, d_out, 0);
    JX_LAUNCH_CHECK();
    return 0;
}

// LM scan of a dense SNP-major f32 block on the device: d_g (nrows, ld >= n); the other arguments as jxg_lm_scan_p32.
// `lm_block_assoc_f32`, src/stats/glm.rs:4313-4497.
extern "C" int jxg_lm_scan_dense(const float *d_g, int nrows, int n, int64_t ld, const double *d_xr, int q0,
                                 const double *d_ixx, double yy_r, double *d_work, double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    if (q0 < 0 || n <= q0 + 1) return fail("n too small: require n > q0+1");
    if (ld < n) return fail("jxg_lm_scan_dense: ld must be >= n");
    hipStream_t st = (hipStream_t)stream;
    const int ncols = q0 + 1;
    double *sums = d_work, *dsum = d_work + (size_t)nrows * ncols;
    const dim3 grid((nrows + 4 * LM_DR - 1) / (4 * LM_DR)), block(256);
    for (int c0 = 0; c0 < ncols; c0 += LM_MAXC) {
        const int nc = std::min(LM_MAXC, ncols - c0);
        double *dd = c0 == 0 ? dsum : nullptr;
        switch (nc) {
        case 1: hipLaunchKernelGGL(lm_dots_dense_kernel<1>, grid, block, 0, st, d_g, ld, nrows, d_xr, ncols, c0, n, sums, ncols, dd); break;
        case 2: hipLaunchKernelGGL(lm_dots_dense_kernel<2>, grid, block, 0, st, d_g, ld, nrows, d_xr, ncols, c0, n, sums, ncols, dd); break;
        case 3: hipLaunchKernelGGL(lm_dots_dense_kernel<3>, grid, block, 0, st, d_g, ld, nrows, d_xr, ncols, c0, n, sums, ncols, dd); break;
        default: hipLaunchKernelGGL(lm_dots_dense_kernel<4>, grid, block, 0, st, d_g, ld, nrows, d_xr, ncols, c0, n, sums, ncols, dd); break;
        }
        JX_LAUNCH_CHECK();
    }
    const int df = n - q0 - 1;
    const double ln_beta = lgamma(0.5 * df) + lgamma(0.5) - lgamma(0.5 * df + 0.5);
    hipLaunchKernelGGL(lm_stats_kernel, dim3((nrows + 255) / 256), dim3(256), 0, st, sums, ncols, dsum, nrows, q0, d_ixx,
                       yy_r, n, df, ln_beta, d_out, 1);
    JX_LAUNCH_CHECK();
    return 0;
}

// GRAMMAR-gamma scan of `nrows` SNPs of a resident P32 image.  d_lut (nrows, 4) f32: mean-imputed additive values by 2-bit
// code ([0, 2 maf, 1, 2] or flipped, not centred); d_xr (n, p + 1) f64 = [X | score vector] (rounded to f32 by the caller);
// d_ixx (p, p) f64 = (X'X)^-1; d_work (nrows * (p + 2)) f64 scratch; d_out (nrows, 3) f64 = beta, se, p.
extern "C" int jxg_splmm_grammar_scan_p32(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                          const float *d_lut, const double *d_xr, int p, const double *d_ixx,
                                          double score_scale, double denom_scale, double sigma2, double *d_work,
                                          double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || n <= p) return fail("SparseLMM scan requires n > p >= 1");
    if (!(std::isfinite(score_scale) && score_scale > 0.0))
        return fail("SparseLMM scan requires finite positive score scale, got " + std::to_string(score_scale));
    if (!(std::isfinite(denom_scale) && denom_scale > 0.0))
        return fail("SparseLMM scan requires finite positive denominator scale, got " + std::to_string(denom_scale));
    if (!(std::isfinite(sigma2) && sigma2 > 0.0))
        return fail("SparseLMM scan requires finite positive Wald sigma2, got " + std::to_string(sigma2));
    hipStream_t st = (hipStream_t)stream;
    const int ncols = p + 1;
    double *sums = d_work, *dsum = d_work + (size_t)nrows * ncols;
    const dim3 grid((nrows + LM_THREADS - 1) / LM_THREADS), block(LM_THREADS);
    for (int c0 = 0; c0 < ncols; c0 += LM_MAXC) {
        const int nc = std::min(LM_MAXC, ncols - c0);
        double *dd = c0 == 0 ? dsum : nullptr;
        switch (nc) {
        case 1: hipLaunchKernelGGL(lm_dots_kernel<1>, grid, block, 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_xr, ncols, c0, n, sums, ncols, dd); break;
        case 2: hipLaunchKernelGGL(lm_dots_kernel<2>, grid, block, 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_xr, ncols, c0, n, sums, ncols, dd); break;
        case 3: hipLaunchKernelGGL(lm_dots_kernel<3>, grid, block, 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_xr, ncols, c0, n, sums, ncols, dd); break;
        default: hipLaunchKernelGGL(lm_dots_kernel<4>, grid, block, 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_xr, ncols, c0, n, sums, ncols, dd); break;
        }
        JX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(splmm_grammar_stats_kernel, dim3((nrows + 255) / 256), dim3(256), 0, st, sums, ncols, dsum, nrows, p,
                       d_ixx, score_scale, denom_scale, sigma2, d_out);
    JX_LAUNCH_CHECK();
    return 0;
}

// d_grot (nrows, ld) f32 rotated marker rows; d_w, d_a (n) f64; d_x (n, p) f64 row-major (X~ = U'X);
// d_out (nrows, 3 + 2 p) f64 = [g~'g~, g~'Wg~, g~'a~, X~'g~ (p), X~'Wg~ (p)].
extern "C" int jxg_splmm_gamma_sums(const float *d_grot, int nrows, int n, int64_t ld, int p, const double *d_w,
                                    const double *d_a, const double *d_x, double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > SG_MAXP) return fail("SparseLMM gamma estimation supports 1 .. 16 design columns, got " + std::to_string(p));
    hipLaunchKernelGGL(splmm_gamma_sums_kernel, dim3(nrows), dim3(256), 0, (hipStream_t)stream, d_grot, ld, n, p, d_w, d_a,
                       d_x, d_out);
    JX_LAUNCH_CHECK();
    return 0;
}
