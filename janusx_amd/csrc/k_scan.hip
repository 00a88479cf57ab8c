// Per-SNP mixed-model kernels on the rotated scale: REML/ML likelihoods, Brent, GLS beta/SE, Wald p,
// fixed-lambda quadratic forms, X/y rotation, null-model fit.
//
// Reference: src/stats/reml.rs:109-198 (rotate X,y), :255-362 (reml_loglike), :364-470 (ml_loglike),
// :472-568 (final_beta_se), :572-616 (null fit); src/math/brent.rs (Brent); src/math/linalg.rs:2-17, 341-398;
// src/stats/lmm.rs:94-199 (exact scan); src/stats/fvlmm.rs:1484-1563, 1691-1805 (fixed-lambda scan).
//
// One 256-thread workgroup per SNP.  Every objective evaluation is two passes over the n rotated samples
// (normal equations, then the explicit residual quadratic form, exactly the reference's formulas) with f64
// wave-shuffle + LDS reductions; the tiny Cholesky/Brent logic runs redundantly in every lane on identical
// reduced values, so control flow stays workgroup-uniform.
#include "scan_common.h"

namespace jx {

template <int MAXD>
struct EvalOut {
    bool ok;
    double q;        // r' V^-1 r
    double logdetv;  // sum ln(s+lbd)
    double logdetx;  // 2 * sum ln L_kk
    double beta_k;   // last coefficient
    double ainv_kk;  // [(X'V^-1X + ridge)^-1]_kk
};

// One full evaluation of the normal equations at lambda (reml.rs:286-344). `g` may be null (null model).
template <int MAXD, bool WAVE>
__device__ void eval_normal_eq(double lbd, const double *__restrict__ s, const double *__restrict__ xcov,
                               const double *__restrict__ y, const float *__restrict__ g, int n, int p_cov,
                               double *shm /* LDS scratch */, EvalOut<MAXD> &o, bool want_ainv) {
    constexpr int NA = MAXD * (MAXD + 1) / 2;
    constexpr int NV = NA + MAXD + 2;
    const int dim = p_cov + (g ? 1 : 0);
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
    // v layout: [0, NA) lower triangle row-major (r, c<=r); [NA, NA+MAXD) b; NA+MAXD logdet; NA+MAXD+1 bad count
#pragma unroll 2
    for (int i = Par<WAVE>::tid(); i < n; i += Par<WAVE>::kThreads) {
        const double vv = s[i] + lbd;
        if (vv <= 0.0) v[NA + MAXD + 1] += 1.0;
        const double vi = 1.0 / vv;
        const double yi = y[i];
        double xr[MAXD];
#pragma unroll
        for (int r = 0; r < MAXD; ++r) {
            if (r < p_cov)
                xr[r] = xcov[(int64_t)i * p_cov + r];
            else if (r == p_cov && g)
                xr[r] = (double)g[i];
            else
                xr[r] = 0.0;
        }
        int idx = 0;
#pragma unroll
        for (int r = 0; r < MAXD; ++r) {
            const double vx = vi * xr[r];
            v[NA + r] += vx * yi;
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                v[idx] += vx * xr[c];
                ++idx;
            }
        }
        v[NA + MAXD] += log(vv);
    }
    const int nv_used = NV;  // reduce everything (unused slots are zeros)
    Par<WAVE>::template sum<NV>(v, nv_used, shm);

    o.ok = true;
    o.logdetv = v[NA + MAXD];
    if (v[NA + MAXD + 1] > 0.0) o.ok = false;

    double a[MAXD * MAXD];
    double b[MAXD], beta[MAXD];
    {
        int idx = 0;
#pragma unroll
        for (int r = 0; r < MAXD; ++r) {
            b[r] = v[NA + r];
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                a[r * MAXD + c] = v[idx];
                a[c * MAXD + r] = v[idx];
                ++idx;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
        if (r < dim) a[r * MAXD + r] += 1e-6;
    if (!chol_inplace<MAXD>(a, dim)) o.ok = false;
    if (!o.ok) {
        o.q = 0.0;
        o.logdetx = 0.0;
        o.beta_k = 0.0;
        o.ainv_kk = 0.0;
        return;  // uniform across the workgroup: every thread sees the same reduced values
    }
    chol_solve<MAXD>(a, dim, b, beta);
    double ld = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
        if (r < dim) ld += log(a[r * MAXD + r]);
    o.logdetx = 2.0 * ld;
    o.beta_k = pick<MAXD>(beta, dim - 1);
    o.ainv_kk = 0.0;
    if (want_ainv) {
        double e[MAXD], xk[MAXD];
#pragma unroll
        for (int r = 0; r < MAXD; ++r) e[r] = (r == dim - 1) ? 1.0 : 0.0;
        chol_solve<MAXD>(a, dim, e, xk);
        o.ainv_kk = pick<MAXD>(xk, dim - 1);
    }
    // pass 2: explicit residual quadratic form (reml.rs:327-344)
    double qv[1] = {0.0};
#pragma unroll 2
    for (int i = Par<WAVE>::tid(); i < n; i += Par<WAVE>::kThreads) {
        const double vi = 1.0 / (s[i] + lbd);
        double xb = 0.0;
#pragma unroll
        for (int r = 0; r < MAXD; ++r) {
            if (r < p_cov)
                xb += xcov[(int64_t)i * p_cov + r] * beta[r];
            else if (r == p_cov && g)
                xb += (double)g[i] * beta[r];
        }
        const double ri = y[i] - xb;
        qv[0] += vi * ri * ri;
    }
    Par<WAVE>::template sum<1>(qv, 1, shm);
    o.q = qv[0];
}

// -reml_loglike (the Brent objective). Failure -> +1e8 (reference returns -1e8 for the log-likelihood).
template <int MAXD, bool WAVE>
__device__ double neg_reml(double x, const double *s, const double *xcov, const double *y, const float *g, int n,
                           int p_cov, double *shm) {
    const double lbd = pow(10.0, x);
    const int dim = p_cov + (g ? 1 : 0);
    if (!isfinite(lbd) || lbd <= 0.0 || n <= dim) return 1e8;
    EvalOut<MAXD> o;
    eval_normal_eq<MAXD, WAVE>(lbd, s, xcov, y, g, n, p_cov, shm, o, false);
    if (!o.ok) return 1e8;
    const double nf = (double)n, pf = (double)dim;
    const double total = (nf - pf) * log(o.q) + o.logdetv + o.logdetx;
    const double c = (nf - pf) * (log(nf - pf) - 1.0 - log(2.0 * M_PI)) / 2.0;
    const double reml = c - 0.5 * total;
    return isfinite(reml) ? -reml : 1e8;
}

// -ml_loglike (objective of the LMM2 second pass and its null fit; reml.rs:364-470). Failure -> +1e8.
template <int MAXD, bool WAVE>
__device__ double neg_ml(double x, const double *s, const double *xcov, const double *y, const float *g, int n,
                         int p_cov, double *shm) {
    const double lbd = pow(10.0, x);
    const int dim = p_cov + (g ? 1 : 0);
    if (!isfinite(lbd) || lbd <= 0.0 || n <= dim) return 1e8;
    EvalOut<MAXD> o;
    eval_normal_eq<MAXD, WAVE>(lbd, s, xcov, y, g, n, p_cov, shm, o, false);
    if (!o.ok || !isfinite(o.q) || o.q <= 0.0) return 1e8;
    const double nf = (double)n;
    const double ml = nf * (log(nf) - 1.0 - log(2.0 * M_PI)) / 2.0 - 0.5 * (nf * log(o.q) + o.logdetv);
    return isfinite(ml) ? -ml : 1e8;
}

template <int MAXD, bool WAVE, bool ML>
__device__ __forceinline__ double neg_obj(double x, const double *s, const double *xcov, const double *y,
                                          const float *g, int n, int p_cov, double *shm) {
    if (ML) return neg_ml<MAXD, WAVE>(x, s, xcov, y, g, n, p_cov, shm);
    return neg_reml<MAXD, WAVE>(x, s, xcov, y, g, n, p_cov, shm);
}

// src/math/brent.rs:1-136, verbatim control flow (including `e` not being updated on parabolic steps).
template <int MAXD, bool WAVE, bool ML = false>
__device__ void brent_reml(const double *s, const double *xcov, const double *y, const float *g, int n, int p_cov,
                           double low, double high, double tol, int max_iter, bool has_init, double init,
                           double *shm, double &xbest, double &fbest, int &evals) {
    double a = low, c = high;
    if (!(a < c)) {
        const double t = a;
        a = c;
        c = t;
    }
    const double eps = 2.220446049250313e-16;
    tol = fmax(fabs(tol), 1e-12);
    double x = (has_init && isfinite(init) && init >= a && init <= c) ? init : 0.5 * (a + c);
    double w = x, v = x;
    double fx = neg_obj<MAXD, WAVE, ML>(x, s, xcov, y, g, n, p_cov, shm);
    double fw = fx, fv = fx;
    double d = 0.0, e = 0.0;
    evals = 1;
    for (int it = 0; it < max_iter; ++it) {
        const double m = 0.5 * (a + c);
        const double tol1 = tol * fabs(x) + eps;
        const double tol2 = 2.0 * tol1;
        if (fabs(x - m) <= tol2 - 0.5 * (c - a)) break;
        double u;
        bool use_par = false;
        if (fabs(e) > tol1) {
            double pq = (x - v) * ((x - w) * (fx - fv)) - (x - w) * ((x - v) * (fx - fw));
            double q = 2.0 * (((x - v) * (fx - fw)) - ((x - w) * (fx - fv)));
            if (q > 0.0)
                pq = -pq;
            else
                q = -q;
            bool ok = false;
            if (fabs(q) > eps) {
                const double sstep = pq / q;
                u = x + sstep;
                if ((u - a) >= tol2 && (c - u) >= tol2 && fabs(sstep) < 0.5 * fabs(e)) ok = true;
            }
            if (ok) {
                d = pq / q;
                u = x + d;
                if ((u - a) < tol2 || (c - u) < tol2) d = (x < m) ? tol1 : -tol1;
                use_par = true;
            }
        }
        if (!use_par) {
            e = (x < m) ? (c - x) : (a - x);
            d = 0.3819660 * e;
        }
        if (fabs(d) < tol1) d = (d >= 0.0) ? tol1 : -tol1;
        u = x + d;
        const double fu = neg_obj<MAXD, WAVE, ML>(u, s, xcov, y, g, n, p_cov, shm);
        ++evals;
        if (fu <= fx) {
            if (u >= x)
                a = x;
            else
                c = x;
            v = w;
            fv = fw;
            w = x;
            fw = fx;
            x = u;
            fx = fu;
        } else {
            if (u >= x)
                c = u;
            else
                a = u;
            if (fu <= fw || w == x) {
                v = w;
                fv = fw;
                w = u;
                fw = fu;
            } else if (fu <= fv || v == x || v == w) {
                v = u;
                fv = fu;
            }
        }
    }
    xbest = x;
    fbest = fx;
}


// src/stats/lmm.rs:94-199: one WAVE per rotated SNP row (4 rows in flight per workgroup, no barriers, no LDS):
// every lane owns the samples lane, lane+64, ...; reductions are wave butterflies, so each wave follows its own
// Brent trajectory with wave-uniform control flow.
template <int MAXD>
__global__ __launch_bounds__(SCAN_THREADS) void lmm_scan_kernel(const float *__restrict__ grot, int nrows, int n,
                                                                const double *__restrict__ s,
                                                                const double *__restrict__ xcov,
                                                                const double *__restrict__ y, int p_cov, double low,
                                                                double high, double tol, int max_iter, int warm,
                                                                double init, int with_plrt, double nullml,
                                                                double *__restrict__ out,
                                                                int32_t *__restrict__ evals_out,
                                                                const int32_t *__restrict__ chain_off, int nchains,
                                                                double *__restrict__ carry) {
    constexpr bool WAVE = true;
    double *shm = nullptr;
    const int out_cols = with_plrt ? 4 : 3;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // chain_off: the reference's warm-start chain (carry_warm_start, lmm.rs:134-161): a wave walks the rows
    // [chain_off[c], chain_off[c + 1]) in order, every SNP's Brent starts from the optimum of the row before it; carry[c] holds
    // the state a chain starts from (NaN: none -- the interval midpoint) and receives the state it ends with, so that a chain
    // may continue in the next launch.  Without chain_off: one row per unit, `warm` / `init` as before.
    const int nunits = chain_off ? nchains : nrows;
    for (int unit = blockIdx.x * SCAN_WAVES + wave; unit < nunits; unit += gridDim.x * SCAN_WAVES) {
      const int r_beg = chain_off ? chain_off[unit] : unit, r_end = chain_off ? chain_off[unit + 1] : unit + 1;
      double last = chain_off ? carry[unit] : init;
      bool have_last = chain_off ? isfinite(last) : (warm != 0);
      for (int r = r_beg; r < r_end; ++r) {
        const float *g = grot + (int64_t)r * n;
        double *o = out + (int64_t)r * out_cols;
        // lmm.rs:63-72: ssq of the rotated row
        double ssq[1] = {0.0};
        for (int i = lane; i < n; i += 64) {
            const double v = (double)g[i];
            ssq[0] += v * v;
        }
        ssq[0] = wave_allsum(ssq[0]);
        if (!isfinite(ssq[0]) || ssq[0] <= 1e-12) {
            if (lane == 0) {
                o[0] = nan("");
                o[1] = nan("");
                o[2] = 1.0;
                if (with_plrt) o[3] = 1.0;
                if (evals_out) evals_out[r] = 0;
            }
            continue;
        }
        double xb, fb;
        int ne = 0;
        brent_reml<MAXD, WAVE>(s, xcov, y, g, n, p_cov, low, high, tol, max_iter, have_last, last, shm, xb, fb, ne);
        if (chain_off) {
            last = xb;
            have_last = true;
        }
        // final_beta_se (reml.rs:472-568)
        const double lbd = pow(10.0, xb);
        const int dim = p_cov + 1;
        double beta = nan(""), se = nan("");
        double qfin = 0.0, ldv = 0.0;
        bool have = false;
        if (isfinite(lbd) && lbd > 0.0 && n > dim) {
            EvalOut<MAXD> e;
            eval_normal_eq<MAXD, WAVE>(lbd, s, xcov, y, g, n, p_cov, shm, e, true);
            if (e.ok) {
                const double sigma2 = e.q / ((double)n - (double)dim);
                const double var = sigma2 * e.ainv_kk;
                if (var > 0.0 && isfinite(var)) {
                    beta = e.beta_k;
                    se = sqrt(var);
                }
                qfin = e.q;
                ldv = e.logdetv;
                have = true;
            }
        }
        if (lane == 0) {
            if (evals_out) evals_out[r] = ne;
            if (isfinite(beta) && isfinite(se) && se > 0.0) {
                const double z = beta / se;
                double pv = 2.0 * (0.5 * erfc(fabs(z) / 1.4142135623730951));
                if (pv < 2.2250738585072014e-308) pv = 2.2250738585072014e-308;
                if (pv > 1.0) pv = 1.0;
                o[0] = beta;
                o[1] = se;
                o[2] = isfinite(pv) ? pv : 1.0;
                if (with_plrt) {
                    // ml_loglike at the optimum (reml.rs:364-470): same normal equations, ML constant
                    double plrt = 1.0;
                    if (have && isfinite(qfin) && qfin > 0.0) {
                        const double nf = (double)n;
                        const double ml = nf * (log(nf) - 1.0 - log(2.0 * M_PI)) / 2.0 - 0.5 * (nf * log(qfin) + ldv);
                        if (isfinite(ml)) {
                            double stat = 2.0 * (ml - nullml);
                            if (!isfinite(stat) || stat < 0.0) stat = 0.0;
                            plrt = chi2_sf_df1_dev(stat);
                        }
                    }
                    o[3] = plrt;
                }
            } else {
                o[0] = nan("");
                o[1] = nan("");
                o[2] = 1.0;
                if (with_plrt) o[3] = 1.0;
            }
        }
      }
      if (chain_off && lane == 0 && have_last) carry[unit] = last;
    }
}

// LMM2 (src/stats/lmm.rs:202-330): REML Brent -> final_beta_se -> ML Brent seeded with the REML optimum -> LRT.
// One wave per rotated row; out (nrows, 6) = [beta, se, pwald, lambda_reml, ml_alt, plrt].
template <int MAXD>
__global__ __launch_bounds__(SCAN_THREADS) void lmm2_scan_kernel(const float *__restrict__ grot, int nrows, int n,
                                                                 const double *__restrict__ s,
                                                                 const double *__restrict__ xcov,
                                                                 const double *__restrict__ y, int p_cov, double low,
                                                                 double high, double tol, int max_iter, int warm,
                                                                 double init, double nullml,
                                                                 double *__restrict__ out) {
    constexpr bool WAVE = true;
    double *shm = nullptr;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * SCAN_WAVES + wave; r < nrows; r += gridDim.x * SCAN_WAVES) {
        const float *g = grot + (int64_t)r * n;
        double *o = out + (int64_t)r * 6;
        double ssq = 0.0;
        for (int i = lane; i < n; i += 64) {
            const double v = (double)g[i];
            ssq += v * v;
        }
        ssq = wave_allsum(ssq);
        double res[6] = {nan(""), nan(""), 1.0, nan(""), nan(""), 1.0};  // lmm.rs:84-91
        if (isfinite(ssq) && ssq > 1e-12) {
            double xr, fr;
            int ne = 0;
            brent_reml<MAXD, WAVE, false>(s, xcov, y, g, n, p_cov, low, high, tol, max_iter, warm != 0, init, shm, xr,
                                          fr, ne);
            const double lbd = pow(10.0, xr);
            const int dim = p_cov + 1;
            double beta = nan(""), se = nan("");
            if (isfinite(lbd) && lbd > 0.0 && n > dim) {
                EvalOut<MAXD> e;
                eval_normal_eq<MAXD, WAVE>(lbd, s, xcov, y, g, n, p_cov, shm, e, true);
                if (e.ok) {
                    const double var = e.q / ((double)n - (double)dim) * e.ainv_kk;
                    if (var > 0.0 && isfinite(var)) {
                        beta = e.beta_k;
                        se = sqrt(var);
                    }
                }
            }
            if (isfinite(beta) && isfinite(se) && se > 0.0) {  // wave-uniform: all lanes hold the same sums
                double pv = 2.0 * (0.5 * erfc(fabs(beta / se) / 1.4142135623730951));
                if (pv < 2.2250738585072014e-308) pv = 2.2250738585072014e-308;
                if (pv > 1.0) pv = 1.0;
                double xm, fm;
                brent_reml<MAXD, WAVE, true>(s, xcov, y, g, n, p_cov, low, high, tol, max_iter, true, xr, shm, xm, fm,
                                             ne);
                double ml_alt = -fm;
                if (!isfinite(ml_alt)) ml_alt = -neg_ml<MAXD, WAVE>(xm, s, xcov, y, g, n, p_cov, shm);
                double stat = isfinite(ml_alt) ? 2.0 * (ml_alt - nullml) : 0.0;
                if (!isfinite(stat) || stat < 0.0) stat = 0.0;
                const double plrt = chi2_sf_df1_dev(stat);
                res[0] = beta;
                res[1] = se;
                res[2] = isfinite(pv) ? pv : 1.0;
                res[3] = lbd;
                res[4] = ml_alt;
                res[5] = isfinite(plrt) ? plrt : 1.0;
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = res[k];
        }
    }
}

// Null ML of the LMM2 scan: Brent on -ML without a SNP column (lmm.rs:2902-2921) -> out2 = (log10 lambda, ml0).
template <int MAXD>
__global__ __launch_bounds__(SCAN_THREADS) void lmm2_null_ml_kernel(const double *__restrict__ s,
                                                                    const double *__restrict__ xcov,
                                                                    const double *__restrict__ y, int n, int p_cov,
                                                                    double low, double high, double tol, int max_iter,
                                                                    int has_init, double init,
                                                                    double *__restrict__ out2) {
    constexpr int NV = MAXD * (MAXD + 1) / 2 + MAXD + 2;
    __shared__ double shm[SCAN_WAVES * NV];
    double xb, fb;
    int ne = 0;
    brent_reml<MAXD, false, true>(s, xcov, y, nullptr, n, p_cov, low, high, tol, max_iter, has_init != 0, init, shm, xb,
                                  fb, ne);
    double ml0 = -fb;
    if (!isfinite(ml0)) ml0 = -neg_ml<MAXD, false>(xb, s, xcov, y, nullptr, n, p_cov, shm);
    if (threadIdx.x == 0) {
        out2[0] = xb;
        out2[1] = ml0;
    }
}

// Null model: Brent on -REML without a SNP column, then ML at the optimum (reml.rs:572-616).
template <int MAXD>
__global__ __launch_bounds__(SCAN_THREADS) void lmm_null_kernel(const double *__restrict__ s,
                                                                const double *__restrict__ xcov,
                                                                const double *__restrict__ y, int n, int p_cov,
                                                                double low, double high, double tol, int max_iter,
                                                                double *__restrict__ out3) {
    constexpr int NV = MAXD * (MAXD + 1) / 2 + MAXD + 2;
    __shared__ double shm[SCAN_WAVES * NV];
    double xb, fb;
    int ne = 0;
    brent_reml<MAXD, false>(s, xcov, y, nullptr, n, p_cov, low, high, tol, max_iter, false, 0.0, shm, xb, fb, ne);
    const double lbd = pow(10.0, xb);
    double ml = -1e8;
    if (isfinite(lbd) && lbd > 0.0 && n > p_cov) {
        EvalOut<MAXD> e;
        eval_normal_eq<MAXD, false>(lbd, s, xcov, y, nullptr, n, p_cov, shm, e, false);
        if (e.ok && isfinite(e.q) && e.q > 0.0) {
            const double nf = (double)n;
            const double v = nf * (log(nf) - 1.0 - log(2.0 * M_PI)) / 2.0 - 0.5 * (nf * log(e.q) + e.logdetv);
            ml = isfinite(v) ? v : -1e8;
        }
    }
    if (threadIdx.x == 0) {
        out3[0] = lbd;
        out3[1] = ml;
        out3[2] = -fb;
    }
}

// ml_loglike / reml_loglike of the null model at a given log10 lambda (reml.rs:255-470) -> out2 = (ml, reml).
template <int MAXD>
__global__ __launch_bounds__(SCAN_THREADS) void lmm_loglike_kernel(const double *__restrict__ s,
                                                                   const double *__restrict__ xcov,
                                                                   const double *__restrict__ y, int n, int p_cov,
                                                                   double log10_lbd, double *__restrict__ out2) {
    constexpr int NV = MAXD * (MAXD + 1) / 2 + MAXD + 2;
    __shared__ double shm[SCAN_WAVES * NV];
    const double lbd = pow(10.0, log10_lbd);
    double ml = -1e8;
    if (isfinite(lbd) && lbd > 0.0 && n > p_cov) {
        EvalOut<MAXD> e;
        eval_normal_eq<MAXD, false>(lbd, s, xcov, y, nullptr, n, p_cov, shm, e, false);
        if (e.ok && isfinite(e.q) && e.q > 0.0) {
            const double nf = (double)n;
            const double v = nf * (log(nf) - 1.0 - log(2.0 * M_PI)) / 2.0 - 0.5 * (nf * log(e.q) + e.logdetv);
            ml = isfinite(v) ? v : -1e8;
        }
    }
    const double nr = neg_reml<MAXD, false>(log10_lbd, s, xcov, y, nullptr, n, p_cov, shm);
    if (threadIdx.x == 0) {
        out2[0] = ml;
        out2[1] = -nr;
    }
}

// X~ = U^T [X | y]: one workgroup per output row (reml.rs:157-170: f32 U^T widened, f64 accumulation).
__global__ __launch_bounds__(SCAN_THREADS) void rotate_xy_kernel(const float *__restrict__ ut, int n,
                                                                 const double *__restrict__ xy, int q,
                                                                 double *__restrict__ out) {
    __shared__ double shm[SCAN_WAVES * 16];
    const int i = blockIdx.x;
    const float *row = ut + (int64_t)i * n;
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.0;
    for (int j = threadIdx.x; j < n; j += SCAN_THREADS) {
        const double u = (double)row[j];
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < q) v[c] += u * xy[(int64_t)j * q + c];
    }
    block_sum<16>(v, q, shm);
    if (threadIdx.x == 0)
        for (int c = 0; c < q; ++c) out[(int64_t)i * q + c] = v[c];
}

// Fixed-lambda cache, single workgroup (fvlmm.rs:1484-1563). scal = (ypy, log_det_v, status) ; a_chol p*p.
template <int MAXD>
__global__ __launch_bounds__(SCAN_THREADS) void fvlmm_prepare_kernel(const double *__restrict__ s,
                                                                     const double *__restrict__ xcov,
                                                                     const double *__restrict__ y, int n, int p,
                                                                     double lbd, float *__restrict__ w,
                                                                     float *__restrict__ py, float *__restrict__ wx,
                                                                     double *__restrict__ a_chol_out,
                                                                     double *__restrict__ scal) {
    constexpr int NA = MAXD * (MAXD + 1) / 2;
    constexpr int NV = NA + MAXD + 3;
    __shared__ double shm[SCAN_WAVES * NV];
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
    for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
        const double vv = s[i] + lbd;
        if (!(isfinite(vv) && vv > 0.0)) v[NA + MAXD + 2] += 1.0;
        const float wf = (float)(1.0 / vv);
        w[i] = wf;
        v[NA + MAXD + 1] += log(vv);
        const double wi = (double)wf;
        const double yi = y[i];
        v[NA + MAXD] += wi * yi * yi;
        int idx = 0;
#pragma unroll
        for (int r = 0; r < MAXD; ++r) {
            const double xir = (r < p) ? xcov[(int64_t)i * p + r] : 0.0;
            v[NA + r] += wi * xir * yi;
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                const double xic = (c < p) ? xcov[(int64_t)i * p + c] : 0.0;
                v[idx] += wi * xir * xic;
                ++idx;
            }
        }
    }
    block_sum<NV>(v, NV, shm);
    double a[MAXD * MAXD], b[MAXD], aib[MAXD];
    {
        int idx = 0;
#pragma unroll
        for (int r = 0; r < MAXD; ++r) {
            b[r] = v[NA + r];
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                a[r * MAXD + c] = v[idx];
                a[c * MAXD + r] = v[idx];
                ++idx;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
        if (r < p) a[r * MAXD + r] += 1e-6;
    double status = 0.0;
    if (v[NA + MAXD + 2] > 0.0) status = 1.0;                        // non-positive s[i]+lbd
    if (status == 0.0 && !chol_inplace<MAXD>(a, p)) status = 2.0;     // X'WX not SPD
    if (status != 0.0) {
        if (threadIdx.x == 0) {
            scal[0] = 0.0;
            scal[1] = 0.0;
            scal[2] = status;
        }
        return;
    }
    chol_solve<MAXD>(a, p, b, aib);
    double dotv = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
        if (r < p) dotv += b[r] * aib[r];
    const double ypy = fmax(v[NA + MAXD] - dotv, 0.0);
    for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
        const double wi = (double)w[i];
        double x_aib = 0.0;
#pragma unroll
        for (int r = 0; r < MAXD; ++r) {
            if (r < p) {
                const double xir = xcov[(int64_t)i * p + r];
                wx[(int64_t)i * p + r] = (float)(wi * xir);
                x_aib += xir * aib[r];
            }
        }
        py[i] = (float)(wi * (y[i] - x_aib));
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int r = 0; r < MAXD; ++r)
#pragma unroll
            for (int c = 0; c < MAXD; ++c)
                if (r < p && c < p) a_chol_out[r * p + c] = (c <= r) ? a[r * MAXD + c] : 0.0;
        scal[0] = ypy;
        scal[1] = v[NA + MAXD + 1];
        scal[2] = 0.0;
    }
}

// Statistics of one SNP from its three weighted sums v = (sum w g~^2, g~.Py~, g~.WX~[0..p-1]) (fvlmm.rs:1728-1805; the
// score_mode note of fvlmm_scan_kernel).  Shared by the row-streaming scan kernel and by the finish kernel behind the
// rotation's fused epilogue (k_rotate.hip).
template <int MAXD>
__device__ __forceinline__ void fvlmm_row_finish(const double *v, const double *l, int p, int n, double ypy, int df,
                                                 int with_plrt, double nullml, double log_det_v, int score_mode,
                                                 double *o) {
    const double n_f = (double)n;
    const double c_ml = n_f * (log(n_f) - 1.0 - log(2.0 * M_PI)) / 2.0;
    double c[MAXD], aic[MAXD];
    // the reference's num / c are f32 GEMM outputs (fvlmm.rs:1708-1727): round like its f32 store
#pragma unroll
    for (int k = 0; k < MAXD; ++k) c[k] = (k < p) ? (score_mode ? v[2 + k] : (double)(float)v[2 + k]) : 0.0;
    chol_solve<MAXD>(l, p, c, aic);
    double ct = 0.0;
#pragma unroll
    for (int k = 0; k < MAXD; ++k)
        if (k < p) ct += c[k] * aic[k];
    if (score_mode) {
        const double denom = fmax(v[0] - ct, 0.0);
        const double score = (double)(float)v[1];           // f32 GEMV output in the reference (:2702-2711)
        const double sigma2 = ypy / (double)df;
        bool ok = isfinite(score) && isfinite(denom) && denom > 1e-30 && isfinite(sigma2) && sigma2 > 0.0;
        double beta = 0.0, se = 0.0, chisq = 0.0;
        if (ok) {
            beta = score / denom;
            const double var_beta = sigma2 / denom;
            ok = isfinite(beta) && isfinite(var_beta) && var_beta > 0.0;
            if (ok) {
                se = sqrt(var_beta);
                chisq = (score * score) / (sigma2 * denom);
                ok = isfinite(se) && se > 0.0 && isfinite(chisq) && chisq >= 0.0;
            }
        }
        o[0] = ok ? beta : nan("");
        o[1] = ok ? se : nan("");
        o[2] = ok ? chi2_sf_df1_dev(chisq) : 1.0;
        return;
    }
    const double schur = v[0] - ct;
    if (schur <= 1e-12 || !isfinite(schur)) {
        o[0] = nan("");
        o[1] = nan("");
        o[2] = nan("");
        if (with_plrt) o[3] = 1.0;
    } else {
        const double nu = (double)(float)v[1];
        const double beta = nu / schur;
        const double rwr = fmax(ypy - (nu * nu) / schur, 0.0);
        const double sigma2 = rwr / (double)df;
        const double se = sqrt(sigma2 / schur);
        double pv = 1.0;
        if (isfinite(se) && se > 0.0 && isfinite(beta)) {
            pv = 2.0 * (0.5 * erfc(fabs(beta / se) / 1.4142135623730951));
            if (pv < 2.2250738585072014e-308) pv = 2.2250738585072014e-308;
            if (pv > 1.0) pv = 1.0;
        }
        o[0] = beta;
        o[1] = se;
        o[2] = pv;
        if (with_plrt) {  // fvlmm.rs:1785-1802
            double stat = 0.0;
            if (rwr > 0.0 && isfinite(rwr)) {
                const double ml = c_ml - 0.5 * (n_f * jx_log(rwr) + log_det_v);
                if (isfinite(ml)) stat = 2.0 * (ml - nullml);
            }
            if (!isfinite(stat) || stat < 0.0) stat = 0.0;
            o[3] = chi2_sf_df1_dev(stat);
        }
    }
}

// Fixed-lambda scan (fvlmm.rs:1691-1805): num = g.Py~, c = g.WX~, d = sum w g^2, Schur complement, Wald test.
template <int MAXD>
__global__ __launch_bounds__(SCAN_THREADS) void fvlmm_scan_kernel(const float *__restrict__ grot, int nrows, int n,
                                                                  int p, const float *__restrict__ w,
                                                                  const float *__restrict__ py,
                                                                  const float *__restrict__ wx,
                                                                  const double *__restrict__ a_chol, double ypy,
                                                                  int df, int with_plrt, double nullml,
                                                                  double log_det_v, double *__restrict__ out,
                                                                  int score_mode) {
    // score_mode != 0: SparseLMM exact scan (src/stats/splmm.rs:2567-2880): same three sums, but the test keeps the
    // NULL model's sigma2 = yPy / df (df = n - p passed in), beta = score / g'Pg, se = sqrt(sigma2 / g'Pg),
    // p = chi2_sf(score^2 / (sigma2 g'Pg)) (`splmm_wald_from_score_denom` :2517-2538); g'Pg = max(g'V^-1 g - c'A^-1 c, 0),
    // c kept in f64 there, rows with g'Pg <= 1e-30 are (NaN, NaN, 1).
    // One WAVE per rotated SNP row (4 rows in flight per workgroup, no LDS, no barriers): the row is streamed once
    // with 16-byte loads (4 samples per lane per step), the p + 2 sums are wave butterflies.
    constexpr int NV = MAXD + 2;
    const int out_cols = with_plrt ? 4 : 3;
    double l[MAXD * MAXD];
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
#pragma unroll
        for (int c = 0; c < MAXD; ++c) l[r * MAXD + c] = (r < p && c < p) ? a_chol[r * p + c] : (r == c ? 1.0 : 0.0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec_ok = (n & 3) == 0;  // rows are 16-byte aligned when n is a multiple of 4
    for (int r = blockIdx.x * SCAN_WAVES + wave; r < nrows; r += gridDim.x * SCAN_WAVES) {
        const float *g = grot + (int64_t)r * n;
        double v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = 0.0;
        if (vec_ok) {
            const float4 *g4 = reinterpret_cast<const float4 *>(g);
            const float4 *w4 = reinterpret_cast<const float4 *>(w);
            const float4 *py4 = reinterpret_cast<const float4 *>(py);
            const int n4 = n >> 2;
#pragma unroll 2
            for (int i4 = lane; i4 < n4; i4 += 64) {
                const float4 gv = g4[i4], wv = w4[i4], pv = py4[i4];
                const double g0 = gv.x, g1 = gv.y, g2 = gv.z, g3 = gv.w;
                v[0] += (double)wv.x * g0 * g0 + (double)wv.y * g1 * g1 + (double)wv.z * g2 * g2 + (double)wv.w * g3 * g3;
                v[1] += g0 * (double)pv.x + g1 * (double)pv.y + g2 * (double)pv.z + g3 * (double)pv.w;
                const float *wxr = wx + (int64_t)i4 * 4 * p;
#pragma unroll
                for (int k = 0; k < MAXD; ++k)
                    if (k < p)
                        v[2 + k] += g0 * (double)wxr[k] + g1 * (double)wxr[p + k] + g2 * (double)wxr[2 * p + k] +
                                    g3 * (double)wxr[3 * p + k];
            }
        } else {
            for (int i = lane; i < n; i += 64) {
                const double gi = (double)g[i];
                v[0] += (double)w[i] * gi * gi;
                v[1] += gi * (double)py[i];
#pragma unroll
                for (int k = 0; k < MAXD; ++k)
                    if (k < p) v[2 + k] += gi * (double)wx[(int64_t)i * p + k];
            }
        }
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (k < 2 + p) v[k] = wave_allsum(v[k]);
        if (lane == 0)
            fvlmm_row_finish<MAXD>(v, l, p, n, ypy, df, with_plrt, nullml, log_det_v, score_mode,
                                   out + (int64_t)r * out_cols);
    }
}

// Finish kernel of the fused rotation epilogue: sums (ntiles, nrows, lds) f64 = per column tile and SNP the tile's share of
// (sum w g~^2, g~.Py~, g~.WX~[k]) written by rotate_f16x2_kernel<true>, added here in tile order; one thread per SNP.
template <int MAXD>
__global__ __launch_bounds__(256) void fvlmm_finish_kernel(const double *__restrict__ sums, int ntiles, int lds, int nrows, int n, int p,
                                                           const double *__restrict__ a_chol, double ypy, int df,
                                                           int with_plrt, double nullml, double log_det_v,
                                                           int score_mode, double *__restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    double l[MAXD * MAXD];
#pragma unroll
    for (int a = 0; a < MAXD; ++a)
#pragma unroll
        for (int c = 0; c < MAXD; ++c) l[a * MAXD + c] = (a < p && c < p) ? a_chol[a * p + c] : (a == c ? 1.0 : 0.0);
    double v[MAXD + 2];
#pragma unroll
    for (int k = 0; k < MAXD + 2; ++k) v[k] = 0.0;
    for (int t = 0; t < ntiles; ++t) {
        const double *sp = sums + ((int64_t)t * nrows + r) * lds;
#pragma unroll
        for (int k = 0; k < MAXD + 2; ++k)
            if (k < 2 + p) v[k] += sp[k];
    }
    fvlmm_row_finish<MAXD>(v, l, p, n, ypy, df, with_plrt, nullml, log_det_v, score_mode,
                           out + (int64_t)r * ((with_plrt && !score_mode) ? 4 : 3));
}

}  // namespace jx

using namespace jx;

#define JX_DISPATCH_DIM(dim, EXPR)                       \
    do {                                                 \
        if ((dim) <= 2) {                                \
            constexpr int MAXD = 2;                      \
            EXPR;                                        \
        } else if ((dim) <= 4) {                         \
            constexpr int MAXD = 4;                      \
            EXPR;                                        \
        } else if ((dim) <= 8) {                         \
            constexpr int MAXD = 8;                      \
            EXPR;                                        \
        } else {                                         \
            constexpr int MAXD = 16;                     \
            EXPR;                                        \
        }                                                \
    } while (0)

extern "C" int jxg_rotate_xy(const float *d_ut, int n, const double *d_xy, int q, double *d_out, void *stream) {
    if (q < 1 || q > 16) return fail("jxg_rotate_xy: q must be in [1,16]");
    hipLaunchKernelGGL(rotate_xy_kernel, dim3(n), dim3(SCAN_THREADS), 0, (hipStream_t)stream, d_ut, n, d_xy, q, d_out);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_lmm_reml_null(const double *d_s, const double *d_xcov, const double *d_y, int n, int p,
                                 double low, double high, int max_iter, double tol, double *d_out3, void *stream) {
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_reml_null: p out of range");
    if (!(low < high)) return fail("low must be < high");
    JX_DISPATCH_DIM(p, hipLaunchKernelGGL(lmm_null_kernel<MAXD>, dim3(1), dim3(SCAN_THREADS), 0, (hipStream_t)stream,
                                          d_s, d_xcov, d_y, n, p, low, high, tol, max_iter, d_out3));
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_lmm2_scan_exact(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                             const double *d_y, int p, double low, double high, double tol, int max_iter, int warm,
                             double init_log10_lbd, double nullml, double *d_out6, void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm2_scan_exact: p out of range");
    if (!(low < high)) return fail("low must be < high");
    const int dim = p + 1;
    int grid = (nrows + SCAN_WAVES - 1) / SCAN_WAVES;
    if (grid > 65536 * 8) grid = 65536 * 8;
    JX_DISPATCH_DIM(dim, hipLaunchKernelGGL(lmm2_scan_kernel<MAXD>, dim3(grid), dim3(SCAN_THREADS), 0,
                                            (hipStream_t)stream, d_grot, nrows, n, d_s, d_xcov, d_y, p, low, high,
                                            tol, max_iter, warm, init_log10_lbd, nullml, d_out6));
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_lmm2_null_ml(const double *d_s, const double *d_xcov, const double *d_y, int n, int p, double low,
                                double high, int max_iter, double tol, int has_init, double init, double *d_out2,
                                void *stream) {
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm2_null_ml: p out of range");
    if (!(low < high)) return fail("low must be < high");
    JX_DISPATCH_DIM(p, hipLaunchKernelGGL(lmm2_null_ml_kernel<MAXD>, dim3(1), dim3(SCAN_THREADS), 0,
                                          (hipStream_t)stream, d_s, d_xcov, d_y, n, p, low, high, tol, max_iter,
                                          has_init, init, d_out2));
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_lmm_loglike_null(const double *d_s, const double *d_xcov, const double *d_y, int n, int p,
                                    double log10_lbd, double *d_out2, void *stream) {
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_loglike_null: p out of range");
    JX_DISPATCH_DIM(p, hipLaunchKernelGGL(lmm_loglike_kernel<MAXD>, dim3(1), dim3(SCAN_THREADS), 0, (hipStream_t)stream,
                                          d_s, d_xcov, d_y, n, p, log10_lbd, d_out2));
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_lmm_scan_exact(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                            const double *d_y, int p, double low, double high, double tol, int max_iter, int warm,
                            double init_log10_lbd, int with_plrt, double nullml, double *d_out, int32_t *d_evals,
                            void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_scan_exact: p out of range");
    if (!(low < high)) return fail("low must be < high");
    const int dim = p + 1;
    int grid = (nrows + SCAN_WAVES - 1) / SCAN_WAVES;
    if (grid > 65536 * 8) grid = 65536 * 8;
    JX_DISPATCH_DIM(dim, hipLaunchKernelGGL(lmm_scan_kernel<MAXD>, dim3(grid), dim3(SCAN_THREADS), 0,
                                            (hipStream_t)stream, d_grot, nrows, n, d_s, d_xcov, d_y, p, low, high,
                                            tol, max_iter, warm, init_log10_lbd, with_plrt, nullml, d_out, d_evals,
                                            (const int32_t *)nullptr, 0, (double *)nullptr));
    JX_LAUNCH_CHECK();
    return 0;
}

// The same scan along the reference's warm-start chains (src/stats/lmm.rs:134-161): d_chain_off (nchains + 1 row offsets into
// this block, ascending), d_carry (nchains doubles: start state in, end state out; NaN = no state).
extern "C" int jxg_lmm_scan_exact_chain(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                                        const double *d_y, int p, double low, double high, double tol, int max_iter,
                                        const int32_t *d_chain_off, int nchains, double *d_carry, int with_plrt, double nullml,
                                        double *d_out, int32_t *d_evals, void *stream) {
    if (nrows <= 0 || nchains <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_scan_exact_chain: p out of range");
    if (!(low < high)) return fail("low must be < high");
    if (!d_chain_off || !d_carry) return fail("jxg_lmm_scan_exact_chain: chain offsets and carry states are required");
    const int dim = p + 1;
    const int grid = (nchains + SCAN_WAVES - 1) / SCAN_WAVES;
    JX_DISPATCH_DIM(dim, hipLaunchKernelGGL(lmm_scan_kernel<MAXD>, dim3(grid), dim3(SCAN_THREADS), 0,
                                            (hipStream_t)stream, d_grot, nrows, n, d_s, d_xcov, d_y, p, low, high,
                                            tol, max_iter, 0, 0.0, with_plrt, nullml, d_out, d_evals, d_chain_off, nchains,
                                            d_carry));
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_fvlmm_prepare(const double *d_s, const double *d_xcov, const double *d_y, int n, int p,
                                 double lbd, float *d_w, float *d_py, float *d_wx, double *h_a_chol,
                                 double *h_scalars3) {
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_fvlmm_prepare: p out of range");
    DevBuf tmp;
    if (tmp.alloc(sizeof(double) * (size_t)(p * p + 3))) return 1;
    double *d_a = tmp.as<double>();
    double *d_sc = d_a + p * p;
    JX_DISPATCH_DIM(p, hipLaunchKernelGGL(fvlmm_prepare_kernel<MAXD>, dim3(1), dim3(SCAN_THREADS), 0, (hipStream_t)0,
                                          d_s, d_xcov, d_y, n, p, lbd, d_w, d_py, d_wx, d_a, d_sc));
    JX_LAUNCH_CHECK();
    double sc[3];
    JX_HIP(hipMemcpy(sc, d_sc, sizeof(sc), hipMemcpyDeviceToHost));
    if (sc[2] == 1.0) return fail("non-positive s[i]+lbd");
    if (sc[2] == 2.0) return fail("X'WX not SPD");
    JX_HIP(hipMemcpy(h_a_chol, d_a, sizeof(double) * p * p, hipMemcpyDeviceToHost));
    const int df = n - p - 1;
    if (df <= 0) return fail("df <= 0");
    h_scalars3[0] = sc[0];
    h_scalars3[1] = sc[1];
    h_scalars3[2] = (double)df;
    return 0;
}

extern "C" int jxg_fvlmm_scan_dev(const float *d_grot, int nrows, int n, int p, const float *d_w, const float *d_py,
                                  const float *d_wx, const double *d_a_chol, double ypy, int df, int with_plrt,
                                  double nullml, double log_det_v, double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_fvlmm_scan: p out of range");
    int grid = (nrows + SCAN_WAVES - 1) / SCAN_WAVES;
    if (grid > 65536 * 8) grid = 65536 * 8;
    JX_DISPATCH_DIM(p, hipLaunchKernelGGL(fvlmm_scan_kernel<MAXD>, dim3(grid), dim3(SCAN_THREADS), 0,
                                          (hipStream_t)stream, d_grot, nrows, n, p, d_w, d_py, d_wx, d_a_chol, ypy, df,
                                          with_plrt, nullml, log_det_v, d_out, 0));
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_fvlmm_scan(const float *d_grot, int nrows, int n, int p, const float *d_w, const float *d_py,
                              const float *d_wx, const double *h_a_chol, double ypy, int df, int with_plrt,
                              double nullml, double log_det_v, double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_fvlmm_scan: p out of range");
    hipStream_t st = (hipStream_t)stream;
    DevBuf a;
    if (a.alloc(sizeof(double) * p * p)) return 1;
    JX_HIP(hipMemcpyAsync(a.p, h_a_chol, sizeof(double) * p * p, hipMemcpyHostToDevice, st));
    int grid = (nrows + SCAN_WAVES - 1) / SCAN_WAVES;
    if (grid > 65536 * 8) grid = 65536 * 8;
    JX_DISPATCH_DIM(p, hipLaunchKernelGGL(fvlmm_scan_kernel<MAXD>, dim3(grid), dim3(SCAN_THREADS), 0, st, d_grot,
                                          nrows, n, p, d_w, d_py, d_wx, a.as<double>(), ypy, df, with_plrt, nullml,
                                          log_det_v, d_out, 0));
    JX_LAUNCH_CHECK();
    JX_HIP(hipStreamSynchronize(st));
    return 0;
}

// SparseLMM exact scan on rotated rows (see the score_mode note in fvlmm_scan_kernel): launch only.
extern "C" int jxg_splmm_exact_scan_dev(const float *d_grot, int nrows, int n, int p, const float *d_w,
                                        const float *d_py, const float *d_wx, const double *d_a_chol, double ypy,
                                        int df, double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_splmm_exact_scan: p out of range");
    if (!(ypy > 0.0) || !std::isfinite(ypy)) return fail("SparseLMM exact scan requires finite positive yPy on K + lambda I scale");
    if (df <= 0) return fail("SparseLMM exact scan requires finite positive df");
    int grid = (nrows + SCAN_WAVES - 1) / SCAN_WAVES;
    if (grid > 65536 * 8) grid = 65536 * 8;
    JX_DISPATCH_DIM(p, hipLaunchKernelGGL(fvlmm_scan_kernel<MAXD>, dim3(grid), dim3(SCAN_THREADS), 0,
                                          (hipStream_t)stream, d_grot, nrows, n, p, d_w, d_py, d_wx, d_a_chol, ypy, df, 0,
                                          0.0, 0.0, d_out, 1));
    JX_LAUNCH_CHECK();
    return 0;
}

// Finish of the fused rotation epilogue (jxg_rotate_packed16x_fused): d_sums (ntiles, nrows, lds) f64 with lds >= p + 2;
// score_mode != 0 = SparseLMM exact scan (df = n - p), else the fixed-lambda scan (df = n - p - 1).
extern "C" int jxg_fvlmm_finish_dev(const double *d_sums, int ntiles, int lds, int nrows, int n, int p, const double *d_a_chol,
                                    double ypy, int df, int with_plrt, double nullml, double log_det_v, int score_mode,
                                    double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV || lds < p + 2 || ntiles < 1) return fail("jxg_fvlmm_finish: p out of range");
    if (score_mode && (!(ypy > 0.0) || !std::isfinite(ypy)))
        return fail("SparseLMM exact scan requires finite positive yPy on K + lambda I scale");
    if (df <= 0) return fail("df <= 0");
    JX_DISPATCH_DIM(p, hipLaunchKernelGGL(fvlmm_finish_kernel<MAXD>, dim3((nrows + 255) / 256), dim3(256), 0,
                                          (hipStream_t)stream, d_sums, ntiles, lds, nrows, n, p, d_a_chol, ypy, df,
                                          score_mode ? 0 : with_plrt, nullml, log_det_v, score_mode, d_out));
    JX_LAUNCH_CHECK();
    return 0;
}
