/* CPU restatement (plain C) of the JanusX per-SNP mixed-model kernels -- TEST INFRASTRUCTURE ONLY.
 *
 * Oracle + timed CPU baseline ("port"): sequential f64 loops in the same order as the reference
 * Rust code (cited per function).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (janusx_amd / libjxgpu.so) never does.
 * Parity status: the table in the header of oracle/jx_oracle.py says which function is pinned by which fixture.  For this
 * file: reml / ml log-likelihoods and the null fit are pinned by values the reference's own Python produced
 * (tests/golden/panel_small.npz, reference_model.npz: LMM._NULLREML, blup.REML, mlm.BLUP._REML, the replayed model layer);
 * the Brent trajectory, final_beta_se and the fixed-lambda block formulas are pinned by reading only (the reference ships no
 * numeric test for them and its crate cannot be built here) and are held equal to the numpy twin (test_python_vs_c_oracle).
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC -o liboracle.so jx_oracle.c -lm   (oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define JXO_MAXDIM 64

/* ---- src/math/linalg.rs:341-363 ---- */
static int chol_inplace(double *a, int dim) {
    for (int i = 0; i < dim; ++i) {
        for (int j = 0; j <= i; ++j) {
            double sum = a[i * dim + j];
            for (int k = 0; k < j; ++k) sum -= a[i * dim + k] * a[j * dim + k];
            if (i == j) {
                if (sum <= 1e-18) return 0;
                a[i * dim + j] = sqrt(sum);
            } else {
                a[i * dim + j] = sum / a[j * dim + j];
            }
        }
        for (int j = i + 1; j < dim; ++j) a[i * dim + j] = 0.0;
    }
    return 1;
}

/* ---- src/stats/reml.rs:46-66 ---- */
static void chol_solve(const double *l, int dim, const double *b, double *x) {
    double y[JXO_MAXDIM];
    for (int i = 0; i < dim; ++i) {
        double sum = b[i];
        for (int k = 0; k < i; ++k) sum -= l[i * dim + k] * y[k];
        y[i] = sum / l[i * dim + i];
    }
    for (int ii = 0; ii < dim; ++ii) {
        int i = dim - 1 - ii;
        double sum = y[i];
        for (int k = i + 1; k < dim; ++k) sum -= l[k * dim + i] * x[k];
        x[i] = sum / l[i * dim + i];
    }
}

/* shared normal-equation build, src/stats/reml.rs:286-319 (identical in reml/ml/final_beta_se).
 * Returns 0 on failure. a = Cholesky factor, beta = solution, *q = r' V^-1 r (explicit residual
 * pass, reml.rs:327-344), *logdetv = sum ln v_i. */
static int normal_eq(double lbd, const double *s, const double *xcov, const double *y, const double *snp,
                     int n, int p_cov, double *a, double *beta, double *q, double *logdetv) {
    int dim = p_cov + (snp ? 1 : 0);
    double b[JXO_MAXDIM];
    memset(a, 0, sizeof(double) * dim * dim);
    memset(b, 0, sizeof(double) * dim);
    for (int i = 0; i < n; ++i) {
        double vv = s[i] + lbd;
        if (vv <= 0.0) return 0;
    }
    for (int i = 0; i < n; ++i) {
        double vi = 1.0 / (s[i] + lbd);
        double yi = y[i];
        for (int r = 0; r < dim; ++r) {
            double xir = (r < p_cov) ? xcov[(size_t)i * p_cov + r] : snp[i];
            b[r] += vi * xir * yi;
            for (int c = 0; c <= r; ++c) {
                double xic = (c < p_cov) ? xcov[(size_t)i * p_cov + c] : snp[i];
                a[r * dim + c] += vi * xir * xic;
            }
        }
    }
    for (int r = 0; r < dim; ++r) {
        a[r * dim + r] += 1e-6;
        for (int c = 0; c < r; ++c) a[c * dim + r] = a[r * dim + c];
    }
    if (!chol_inplace(a, dim)) return 0;
    chol_solve(a, dim, b, beta);
    double qq = 0.0, ld = 0.0;
    for (int i = 0; i < n; ++i) {
        double xb = 0.0;
        for (int r = 0; r < dim; ++r) {
            double xir = (r < p_cov) ? xcov[(size_t)i * p_cov + r] : snp[i];
            xb += xir * beta[r];
        }
        double ri = y[i] - xb;
        double vv = s[i] + lbd;
        qq += (1.0 / vv) * ri * ri;
    }
    for (int i = 0; i < n; ++i) ld += log(s[i] + lbd);
    *q = qq;
    *logdetv = ld;
    return 1;
}

/* ---- src/stats/reml.rs:255-362 ---- */
double jxo_reml_loglike(double log10_lbd, const double *s, const double *xcov, const double *y,
                        const double *snp, int n, int p_cov) {
    double lbd = pow(10.0, log10_lbd);
    if (!isfinite(lbd) || lbd <= 0.0) return -1e8;
    int dim = p_cov + (snp ? 1 : 0);
    if (n <= dim || dim > JXO_MAXDIM) return -1e8;
    double a[JXO_MAXDIM * JXO_MAXDIM], beta[JXO_MAXDIM], q, ldv;
    if (!normal_eq(lbd, s, xcov, y, snp, n, p_cov, a, beta, &q, &ldv)) return -1e8;
    double ldx = 0.0;
    for (int i = 0; i < dim; ++i) ldx += log(a[i * dim + i]);
    ldx *= 2.0;
    double nf = (double)n, pf = (double)dim;
    double total = (nf - pf) * log(q) + ldv + ldx;
    double c = (nf - pf) * (log(nf - pf) - 1.0 - log(2.0 * M_PI)) / 2.0;
    double reml = c - 0.5 * total;
    return isfinite(reml) ? reml : -1e8;
}

/* ---- src/stats/reml.rs:364-470 ---- */
double jxo_ml_loglike(double log10_lbd, const double *s, const double *xcov, const double *y,
                      const double *snp, int n, int p_cov) {
    double lbd = pow(10.0, log10_lbd);
    if (!isfinite(lbd) || lbd <= 0.0) return -1e8;
    int dim = p_cov + (snp ? 1 : 0);
    if (n <= dim || dim > JXO_MAXDIM) return -1e8;
    double a[JXO_MAXDIM * JXO_MAXDIM], beta[JXO_MAXDIM], q, ldv;
    if (!normal_eq(lbd, s, xcov, y, snp, n, p_cov, a, beta, &q, &ldv)) return -1e8;
    if (!isfinite(q) || q <= 0.0) return -1e8;
    double nf = (double)n;
    double total = nf * log(q) + ldv;
    double c = nf * (log(nf) - 1.0 - log(2.0 * M_PI)) / 2.0;
    double ml = c - 0.5 * total;
    return isfinite(ml) ? ml : -1e8;
}

/* ---- src/stats/reml.rs:472-568 ---- */
void jxo_final_beta_se(double log10_lbd, const double *s, const double *xcov, const double *y,
                       const double *snp, int n, int p_cov, double *out3) {
    double lbd = pow(10.0, log10_lbd);
    out3[0] = out3[1] = NAN;
    out3[2] = lbd;
    if (!isfinite(lbd) || lbd <= 0.0) { out3[2] = NAN; return; }
    int dim = p_cov + 1;
    if (n <= dim || dim > JXO_MAXDIM) return;
    double a[JXO_MAXDIM * JXO_MAXDIM], beta[JXO_MAXDIM], q, ldv;
    if (!normal_eq(lbd, s, xcov, y, snp, n, p_cov, a, beta, &q, &ldv)) return;
    double sigma2 = q / ((double)n - (double)dim);
    double e[JXO_MAXDIM], x[JXO_MAXDIM];
    memset(e, 0, sizeof(e));
    e[dim - 1] = 1.0;
    chol_solve(a, dim, e, x);
    double var = sigma2 * x[dim - 1];
    if (var <= 0.0 || !isfinite(var)) return;
    out3[0] = beta[dim - 1];
    out3[1] = sqrt(var);
}

/* ---- src/math/brent.rs:1-136 ---- */
typedef struct {
    const double *s, *xcov, *y, *snp;
    int n, p_cov;
    int use_ml; /* objective: -reml_loglike (0) or -ml_loglike (1, the LMM2 second pass) */
} reml_ctx;

static double neg_reml(double x, const reml_ctx *c) {
    if (c->use_ml) return -jxo_ml_loglike(x, c->s, c->xcov, c->y, c->snp, c->n, c->p_cov);
    return -jxo_reml_loglike(x, c->s, c->xcov, c->y, c->snp, c->n, c->p_cov);
}

static void brent_min(const reml_ctx *ctx, double low, double high, double tol, int max_iter, int has_init,
                      double init, double *xbest, double *fbest, int *n_evals) {
    double a = low, c = high;
    if (!(a < c)) { double t = a; a = c; c = t; }
    const double eps = DBL_EPSILON;
    tol = fmax(fabs(tol), 1e-12);
    double x = (has_init && isfinite(init) && init >= a && init <= c) ? init : 0.5 * (a + c);
    double w = x, v = x;
    double fx = neg_reml(x, ctx), fw = fx, fv = fx;
    double d = 0.0, e = 0.0;
    int evals = 1;
    for (int it = 0; it < max_iter; ++it) {
        double m = 0.5 * (a + c);
        double tol1 = tol * fabs(x) + eps;
        double tol2 = 2.0 * tol1;
        if (fabs(x - m) <= tol2 - 0.5 * (c - a)) break;
        double u;
        int use_par = 0;
        if (fabs(e) > tol1) {
            double p = (x - v) * ((x - w) * (fx - fv)) - (x - w) * ((x - v) * (fx - fw));
            double q = 2.0 * (((x - v) * (fx - fw)) - ((x - w) * (fx - fv)));
            if (q > 0.0) p = -p; else q = -q;
            int ok = 0;
            if (fabs(q) > eps) {
                double sstep = p / q;
                u = x + sstep;
                if ((u - a) >= tol2 && (c - u) >= tol2 && fabs(sstep) < 0.5 * fabs(e)) ok = 1;
            }
            if (ok) {
                d = p / q;
                u = x + d;
                if ((u - a) < tol2 || (c - u) < tol2) d = (x < m) ? tol1 : -tol1;
                use_par = 1;
            }
        }
        if (!use_par) {
            e = (x < m) ? (c - x) : (a - x);
            d = 0.3819660 * e;
        }
        if (fabs(d) < tol1) d = (d >= 0.0) ? tol1 : -tol1;
        u = x + d;
        double fu = neg_reml(u, ctx);
        ++evals;
        if (fu <= fx) {
            if (u >= x) a = x; else c = x;
            v = w; fv = fw;
            w = x; fw = fx;
            x = u; fx = fu;
        } else {
            if (u >= x) c = u; else a = u;
            if (fu <= fw || w == x) {
                v = w; fv = fw;
                w = u; fw = fu;
            } else if (fu <= fv || v == x || v == w) {
                v = u; fv = fu;
            }
        }
    }
    *xbest = x;
    *fbest = fx;
    if (n_evals) *n_evals = evals;
}

/* `lmm_reml_null_f32`, src/stats/reml.rs:572-616 -> out = (lbd, ml, reml) */
void jxo_lmm_reml_null(const double *s, const double *xcov, const double *y, int n, int p_cov, double low,
                       double high, int max_iter, double tol, double *out3) {
    reml_ctx ctx = {s, xcov, y, NULL, n, p_cov, 0};
    double xb, fb;
    brent_min(&ctx, low, high, tol, max_iter, 0, 0.0, &xb, &fb, NULL);
    out3[0] = pow(10.0, xb);
    out3[1] = jxo_ml_loglike(xb, s, xcov, y, NULL, n, p_cov);
    out3[2] = -fb;
}

static double chi2_sf_df1(double stat) { /* src/math/linalg.rs:7-17 */
    if (!isfinite(stat) || stat <= 0.0) return 1.0;
    double p = erfc(sqrt(0.5 * stat));
    if (!isfinite(p)) return 1.0;
    if (p < DBL_MIN) p = DBL_MIN;
    if (p > 1.0) p = 1.0;
    return p;
}

/* `run_rotated_reml_assoc_block_f32`, src/stats/lmm.rs:94-199.
 * warm: 0 = no warm start (core API contract), 1 = seed every SNP with `init` (seed_with_init_guess),
 *       2 = sequential chain through the block (carry_warm_start with one rayon split). */
void jxo_lmm_scan_rotated_block(const float *g_rot, int rows, int n, const double *s, const double *xcov,
                                const double *y, int p_cov, double low, double high, double tol, int max_iter,
                                int warm, double init, int with_plrt, double nullml, double *out,
                                int *evals_out, int threads) {
    int out_cols = with_plrt ? 4 : 3;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    if (warm == 2) threads = 1;
#pragma omp parallel if (warm != 2)
    {
        double *snp = (double *)malloc(sizeof(double) * (size_t)n);
        double last = init;
        int have_last = (warm != 0) && isfinite(init);
#pragma omp for schedule(dynamic, 16)
        for (int r = 0; r < rows; ++r) {
            const float *row = g_rot + (size_t)r * n;
            double *o = out + (size_t)r * out_cols;
            double ssq = 0.0;
            for (int i = 0; i < n; ++i) { /* lmm.rs:63-72 */
                double v = (double)row[i];
                snp[i] = v;
                ssq += v * v;
            }
            if (evals_out) evals_out[r] = 0;
            if (!isfinite(ssq) || ssq <= 1e-12) {
                o[0] = NAN; o[1] = NAN; o[2] = 1.0;
                if (with_plrt) o[3] = 1.0;
                continue;
            }
            reml_ctx ctx = {s, xcov, y, snp, n, p_cov, 0};
            double xb, fb;
            int ne = 0;
            int hi = (warm == 1) ? isfinite(init) : (warm == 2 ? have_last : 0);
            brent_min(&ctx, low, high, tol, max_iter, hi, (warm == 2) ? last : init, &xb, &fb, &ne);
            if (warm == 2) { last = xb; have_last = 1; }
            if (evals_out) evals_out[r] = ne;
            double bs[3];
            jxo_final_beta_se(xb, s, xcov, y, snp, n, p_cov, bs);
            if (isfinite(bs[0]) && isfinite(bs[1]) && bs[1] > 0.0) {
                double z = bs[0] / bs[1];
                double p = 2.0 * (0.5 * erfc(fabs(z) / M_SQRT2));
                if (p < DBL_MIN) p = DBL_MIN;
                if (p > 1.0) p = 1.0;
                o[0] = bs[0]; o[1] = bs[1]; o[2] = isfinite(p) ? p : 1.0;
                if (with_plrt) {
                    double ml = jxo_ml_loglike(xb, s, xcov, y, snp, n, p_cov);
                    if (isfinite(ml)) {
                        double stat = 2.0 * (ml - nullml);
                        if (!isfinite(stat) || stat < 0.0) stat = 0.0;
                        o[3] = chi2_sf_df1(stat);
                    } else {
                        o[3] = 1.0;
                    }
                }
            } else {
                o[0] = NAN; o[1] = NAN; o[2] = 1.0;
                if (with_plrt) o[3] = 1.0;
            }
        }
        free(snp);
    }
}

/* Null ML of the LMM2 scan: Brent on -ml_loglike without a SNP column (src/stats/lmm.rs:2902-2921) -> out2 =
 * (log10 lambda, ml0). */
void jxo_lmm2_null_ml(const double *s, const double *xcov, const double *y, int n, int p_cov, double low,
                      double high, int max_iter, double tol, int has_init, double init, double *out2) {
    reml_ctx ctx = {s, xcov, y, NULL, n, p_cov, 1};
    double xb, fb;
    brent_min(&ctx, low, high, tol, max_iter, has_init, init, &xb, &fb, NULL);
    double ml0 = -fb;
    if (!isfinite(ml0)) ml0 = jxo_ml_loglike(xb, s, xcov, y, NULL, n, p_cov);
    out2[0] = xb;
    out2[1] = ml0;
}

/* `run_rotated_lmm2_assoc_block_f32`, src/stats/lmm.rs:202-330, without the warm-start chain: REML Brent (seeded
 * with init when has_init) -> final_beta_se -> ML Brent seeded with the REML optimum -> LRT against nullml.
 * out (rows, 6) = [beta, se, pwald, lambda_reml, ml_alt, plrt]. */
void jxo_lmm2_scan_rotated_block(const float *g_rot, int rows, int n, const double *s, const double *xcov,
                                 const double *y, int p_cov, double low, double high, double tol, int max_iter,
                                 int has_init, double init, double nullml, double *out, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        double *snp = (double *)malloc(sizeof(double) * (size_t)n);
#pragma omp for schedule(dynamic, 16)
        for (int r = 0; r < rows; ++r) {
            const float *row = g_rot + (size_t)r * n;
            double *o = out + (size_t)r * 6;
            double ssq = 0.0;
            for (int i = 0; i < n; ++i) {
                double v = (double)row[i];
                snp[i] = v;
                ssq += v * v;
            }
            o[0] = NAN; o[1] = NAN; o[2] = 1.0; o[3] = NAN; o[4] = NAN; o[5] = 1.0; /* lmm.rs:84-91 */
            if (!isfinite(ssq) || ssq <= 1e-12) continue;
            reml_ctx ctx = {s, xcov, y, snp, n, p_cov, 0};
            double xr, fr;
            brent_min(&ctx, low, high, tol, max_iter, has_init && isfinite(init), init, &xr, &fr, NULL);
            double bs[3];
            jxo_final_beta_se(xr, s, xcov, y, snp, n, p_cov, bs);
            if (!(isfinite(bs[0]) && isfinite(bs[1]) && bs[1] > 0.0)) continue;
            double z = bs[0] / bs[1];
            double pw = 2.0 * (0.5 * erfc(fabs(z) / M_SQRT2));
            if (pw < DBL_MIN) pw = DBL_MIN;
            if (pw > 1.0) pw = 1.0;
            reml_ctx mctx = {s, xcov, y, snp, n, p_cov, 1};
            double xm, fm;
            brent_min(&mctx, low, high, tol, max_iter, 1, xr, &xm, &fm, NULL);
            double ml_alt = -fm;
            if (!isfinite(ml_alt)) ml_alt = jxo_ml_loglike(xm, s, xcov, y, snp, n, p_cov);
            double stat = isfinite(ml_alt) ? 2.0 * (ml_alt - nullml) : 0.0;
            if (!isfinite(stat) || stat < 0.0) stat = 0.0;
            double plrt = chi2_sf_df1(stat);
            o[0] = bs[0]; o[1] = bs[1]; o[2] = isfinite(pw) ? pw : 1.0;
            o[3] = bs[2]; o[4] = ml_alt; o[5] = isfinite(plrt) ? plrt : 1.0;
        }
        free(snp);
    }
}

/* fixed-lambda epilogue, src/stats/fvlmm.rs:1729-1803. num (rows) and cbuf (rows,p) are the f32 GEMM
 * results g_rot*Py~ and g_rot*WX~ (computed by the caller's BLAS, as the reference does). */
void jxo_fvlmm_assoc_block(const float *g_rot, int rows, int n, int p, const float *w, const float *num,
                           const float *cbuf, const double *a_chol, double ypy, int df, double *out,
                           int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float *row = g_rot + (size_t)r * n;
        double *o = out + (size_t)r * 3;
        double d = 0.0;
        for (int i = 0; i < n; ++i) {
            double gi = (double)row[i];
            d += (double)w[i] * gi * gi;
        }
        double c[JXO_MAXDIM], aic[JXO_MAXDIM];
        for (int k = 0; k < p; ++k) c[k] = (double)cbuf[(size_t)r * p + k];
        chol_solve(a_chol, p, c, aic);
        double ct = 0.0;
        for (int k = 0; k < p; ++k) ct += c[k] * aic[k];
        double schur = d - ct;
        if (schur <= 1e-12 || !isfinite(schur)) { o[0] = o[1] = o[2] = NAN; continue; }
        double nu = (double)num[r];
        double beta = nu / schur;
        double rwr = fmax(ypy - (nu * nu) / schur, 0.0);
        double sigma2 = rwr / (double)df;
        double se = sqrt(sigma2 / schur);
        double pv = 1.0;
        if (isfinite(se) && se > 0.0 && isfinite(beta)) {
            pv = 2.0 * (0.5 * erfc(fabs(beta / se) / M_SQRT2));
            if (pv < DBL_MIN) pv = DBL_MIN;
            if (pv > 1.0) pv = 1.0;
        }
        o[0] = beta; o[1] = se; o[2] = pv;
    }
}

/* 2-bit decode with a per-row 4-entry f32 LUT (src/math/bedmath.rs:508 `decode_row_centered_full_lut`).
 * lut: (rows,4) f32 indexed by code. center!=0 subtracts the actual row mean (decode.rs:181-189). */
void jxo_decode_rows_lut_f32(const uint8_t *packed, int64_t bps, int n, const int64_t *row_idx, int rows,
                             const float *lut, int center, float *out, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const uint8_t *src = packed + (size_t)(row_idx ? row_idx[r] : r) * bps;
        const float *l = lut + (size_t)r * 4;
        float *dst = out + (size_t)r * n;
        double sum = 0.0;
        for (int i = 0; i < n; ++i) {
            float v = l[(src[i >> 2] >> (2 * (i & 3))) & 3];
            dst[i] = v;
            sum += (double)v;
        }
        if (center) {
            float mean = (float)(sum / (double)n);
            for (int i = 0; i < n; ++i) dst[i] -= mean;
        }
    }
}

/* per-row (missing, het, hom_alt) counts, src/io/gfreader.rs:1378-1395 */
void jxo_row_counts(const uint8_t *packed, int64_t bps, int n, int64_t m, int64_t *missing, int64_t *het,
                    int64_t *hom) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < m; ++j) {
        const uint8_t *src = packed + (size_t)j * bps;
        int64_t c1 = 0, c2 = 0, c3 = 0;
        for (int i = 0; i < n; ++i) {
            int code = (src[i >> 2] >> (2 * (i & 3))) & 3;
            c1 += (code == 1);
            c2 += (code == 2);
            c3 += (code == 3);
        }
        missing[j] = c1; het[j] = c2; hom[j] = c3;
    }
}

int jxo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
