// GBLUP: intercept-only REML on the spectral scale + BLUP back-solve + cross-GRM predictions.
// Reference: src/stats/gblup.rs:1105-1240 (`fit_gblup_reml_from_grm_row_major_f64`), :1258-1516
// (`gblup_reml_npy_grm`), `square_matrix_subset_cross_dot_f64` for the predictions K[*,train] alpha + beta0.
// The O(n^3) part is the eigendecomposition (eigh.cpp / k_sytrd.hip); everything here is O(n^2) or O(n) per
// Brent evaluation and runs in a handful of small kernels.
#include <cmath>
#include <vector>

#include "scan_common.h"

namespace jx {

// x~[k] = sum_r U[r][k], y~[k] = sum_r U[r][k] yc[r] with ut = U^T row-major (row k = eigenvector k)
__global__ __launch_bounds__(SCAN_THREADS) void gblup_rot_kernel(const double *__restrict__ ut, int n,
                                                                 const double *__restrict__ yc,
                                                                 double *__restrict__ x_rot,
                                                                 double *__restrict__ y_rot) {
    __shared__ double shm[SCAN_WAVES * 2];
    const int k = blockIdx.x;
    const double *row = ut + (int64_t)k * n;
    double v[2] = {0.0, 0.0};
    for (int r = threadIdx.x; r < n; r += SCAN_THREADS) {
        const double u = row[r];
        v[0] += u;
        v[1] += u * yc[r];
    }
    block_sum<2>(v, 2, shm);
    if (threadIdx.x == 0) {
        x_rot[k] = v[0];
        y_rot[k] = v[1];
    }
}

struct GblupEval {
    bool ok;
    double reml, ml, beta, q;
};

__device__ void gblup_eval(double x, const double *__restrict__ s, const double *__restrict__ xr,
                           const double *__restrict__ yr, int n, double *shm, double *t_out, GblupEval &o) {
    const double v_floor = 1e-12;
    o.ok = false;
    o.reml = o.ml = o.beta = o.q = 0.0;
    const double lbd = pow(10.0, x);
    if (!(isfinite(lbd) && lbd > 0.0)) return;
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
        const double vi = fmax(s[i] + lbd, v_floor);
        const double inv = 1.0 / vi;
        v[0] += log(vi);
        v[1] += inv * xr[i] * xr[i];
        v[2] += inv * xr[i] * yr[i];
    }
    block_sum<3>(v, 3, shm);
    if (!(isfinite(v[1]) && v[1] > v_floor)) return;
    const double beta = v[2] / v[1];
    double q[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
        const double inv = 1.0 / fmax(s[i] + lbd, v_floor);
        const double ri = yr[i] - xr[i] * beta;
        q[0] += inv * ri * ri;
        if (t_out) t_out[i] = inv * ri;
    }
    block_sum<1>(q, 1, shm);
    if (!(isfinite(q[0]) && q[0] > v_floor)) return;
    const double n_eff = (double)(n - 1), nf = (double)n;
    const double c_reml = n_eff * (log(n_eff) - 1.0 - log(2.0 * M_PI)) / 2.0;
    const double c_ml = nf * (log(nf) - 1.0 - log(2.0 * M_PI)) / 2.0;
    const double reml = c_reml - 0.5 * (n_eff * log(q[0]) + v[0] + log(v[1]));
    const double ml = c_ml - 0.5 * (nf * log(q[0]) + v[0]);
    if (!(isfinite(reml) && isfinite(ml))) return;
    o.ok = true;
    o.reml = reml;
    o.ml = ml;
    o.beta = beta;
    o.q = q[0];
}

// Brent (src/math/brent.rs) on -REML (1e100 on failure), then the optimum's quantities.
// out: [0] status (0 ok), [1] lambda, [2] beta_rot, [3] rtv_invr, [4] ml, [5] reml; t (n) = v^-1 r at the optimum.
__global__ __launch_bounds__(SCAN_THREADS) void gblup_reml_kernel(const double *__restrict__ s,
                                                                  const double *__restrict__ xr,
                                                                  const double *__restrict__ yr, int n, double low,
                                                                  double high, double tol_in, int max_iter,
                                                                  double *__restrict__ t, double *__restrict__ out) {
    __shared__ double shm[SCAN_WAVES * 3];
    GblupEval ev;
    auto cost = [&](double xx) {
        gblup_eval(xx, s, xr, yr, n, shm, nullptr, ev);
        return ev.ok ? -ev.reml : 1e100;
    };
    double a = low, c = high;
    if (!(a < c)) {
        const double tt = a;
        a = c;
        c = tt;
    }
    const double eps = 2.220446049250313e-16;
    const double tol = fmax(fabs(tol_in), 1e-12);
    double x = 0.5 * (a + c);
    double w = x, v = x;
    double fx = cost(x), fw = fx, fv = fx;
    double d = 0.0, e = 0.0;
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
        const double fu = cost(u);
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
    gblup_eval(x, s, xr, yr, n, shm, t, ev);
    if (threadIdx.x == 0) {
        out[0] = ev.ok ? 0.0 : 1.0;
        out[1] = pow(10.0, x);
        out[2] = ev.beta;
        out[3] = ev.q;
        out[4] = ev.ml;
        out[5] = ev.reml;
    }
}

// alpha[row] = sum_k U[row][k] t[k] = sum_k ut[k][row] t[k]; one thread per row, coalesced over rows
__global__ __launch_bounds__(SCAN_THREADS) void gblup_alpha_kernel(const double *__restrict__ ut, int n,
                                                                   const double *__restrict__ t,
                                                                   double *__restrict__ alpha) {
    const int row = blockIdx.x * SCAN_THREADS + threadIdx.x;
    if (row >= n) return;
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < n; ++k) acc += ut[(int64_t)k * n + row] * t[k];
    alpha[row] = acc;
}

// out[i] = beta0 + sum_j K[rows[i]][cols[j]] alpha[j]  (K (n_full, n_full) f32 or f64 row-major)
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void cross_dot_kernel(const T *__restrict__ k, int64_t n_full,
                                                                 const int32_t *__restrict__ rows,
                                                                 const int32_t *__restrict__ cols, int ncols,
                                                                 const double *__restrict__ alpha, double beta0,
                                                                 double *__restrict__ out) {
    __shared__ double shm[SCAN_WAVES];
    const T *krow = k + (int64_t)rows[blockIdx.x] * n_full;
    double v[1] = {0.0};
    for (int j = threadIdx.x; j < ncols; j += SCAN_THREADS) v[0] += (double)krow[cols[j]] * alpha[j];
    block_sum<1>(v, 1, shm);
    if (threadIdx.x == 0) out[blockIdx.x] = v[0] + beta0;
}


// ---- matrix-free products with the 2-bit genotype matrix (P32 image: p32[tile][snp][32 B], 128 samples per record) --
// The decoded value of (SNP r, sample i) is lut[r][code]: any of the reference's per-SNP decodes (mean-imputed
// raw genotype, centred, standardised) is a 4-entry table, so M' alpha and M beta never materialise M.

// out[r] += sum_i lut[r][code(r,i)] * alpha[i]  (`compute_malpha_from_meta_stream`, src/stats/gblup.rs:859-925;
// also the Z'v half of the PCG operator, src/math/pcg.rs:578-640).  grid (ceil(nrows/256), ntiles); thread = SNP.
__global__ __launch_bounds__(256) void packed_tdot_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                          const int32_t *__restrict__ rows, int nrows,
                                                          const float *__restrict__ lut, const double *__restrict__ alpha,
                                                          int n, double *__restrict__ out) {
    __shared__ double a_sh[128];
    __shared__ double a_tot;
    const int tile = blockIdx.y;
    if (threadIdx.x < 128) {
        const int i = tile * 128 + threadIdx.x;
        a_sh[threadIdx.x] = (i < n) ? alpha[i] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < 64) {      // the tile's sum by the first wave (fixed tree), not by one thread
        double t = a_sh[threadIdx.x] + a_sh[threadIdx.x + 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (threadIdx.x == 0) a_tot = t;
    }
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const int64_t rec = rows ? (int64_t)rows[r] : (int64_t)r;
    const uint4 *p = reinterpret_cast<const uint4 *>(p32 + ((int64_t)tile * m_total + rec) * 32);
    const uint4 w0 = p[0], w1 = p[1];
    const uint32_t words[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    double c1 = 0.0, c2 = 0.0, c3 = 0.0;  // class sums of alpha for codes 01 (missing), 10 (het), 11 (hom alt)
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        uint32_t word = words[w];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t code = word & 3u;
            word >>= 2;
            const double a = a_sh[w * 16 + k];
            c1 += (code == 1u) ? a : 0.0;
            c2 += (code == 2u) ? a : 0.0;
            c3 += (code == 3u) ? a : 0.0;
        }
    }
    const double c0 = a_tot - c1 - c2 - c3;
    const float *l = lut + (int64_t)r * 4;
    const double v = (double)l[0] * c0 + (double)l[1] * c1 + (double)l[2] * c2 + (double)l[3] * c3;
    if (v != 0.0) unsafeAtomicAdd(&out[r], v);
}


// out[r] += sum_i lut[r][code(r,i)] * u[i] with u rounded to f32 (the PCG vectors of rrblup_pcg_bed are f32), bit-plane
// form: per record the lo / hi / both bit planes of the 2-bit codes index ONE 16-entry partial-sum table per sample
// quad (T[q][x] = sum_k bit_2k(x) u[4q+k], x a 0x55-masked byte), i.e. three LDS lookups + three f32 adds per four
// genotypes instead of a decode and an f64 select-add per genotype:
//   S_lo = sum u [b0], S_hi = sum u [b1], S_b = sum u [b0 & b1]  ->  missing = S_lo - S_b, het = S_hi - S_b, hom = S_b.
// Partial sums stay f32 inside a 128-sample tile (the reference's GEMV is f32 throughout), tiles merge in f64.
// grid (ceil(nrows / (256 * PT_RPT)), ntiles), 256 threads, thread = SNP.
constexpr int PT_RPT = 4;
constexpr int PT_STRIDE = 86;   // 0x55 + 1 table entries per quad
__global__ __launch_bounds__(256) void packed_tdot_f32_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                              const int32_t *__restrict__ rows, int nrows,
                                                              const float *__restrict__ lut,
                                                              const double *__restrict__ u, int n,
                                                              double *__restrict__ out) {
    __shared__ float tab[32 * PT_STRIDE];
    __shared__ float u_sh[128];
    __shared__ float u_tot;
    const int tile = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < 128) {
        const int i = tile * 128 + tid;
        u_sh[tid] = (i < n) ? (float)u[i] : 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < 32 * 16; e += 256) {   // the 16 valid (0x55-pattern) entries of every quad
        const int q = e >> 4, b = e & 15;
        const int x = (b & 1) | ((b & 2) << 1) | ((b & 4) << 2) | ((b & 8) << 3);
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((b >> k) & 1) t += u_sh[4 * q + k];
        tab[q * PT_STRIDE + x] = t;
    }
    if (tid < 64) {
        // sum of the tile's 128 values by the first wave (fixed tree: two per lane, then a butterfly) -- one thread adding 128
        // dependent LDS reads held the whole workgroup at the barrier below for longer than its lookups take
        double t = (double)u_sh[tid] + (double)u_sh[tid + 64];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (tid == 0) u_tot = (float)t;
    }
    __syncthreads();
    const float utot = u_tot;
#pragma unroll 1
    for (int rr = 0; rr < PT_RPT; ++rr) {
        const int r = (blockIdx.x * PT_RPT + rr) * 256 + tid;
        if (r >= nrows) break;
        const int64_t rec = rows ? (int64_t)rows[r] : (int64_t)r;
        const uint4 *pp = reinterpret_cast<const uint4 *>(p32 + ((int64_t)tile * m_total + rec) * 32);
        const uint4 w0 = pp[0], w1 = pp[1];
        const uint32_t words[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        float slo = 0.0f, shi = 0.0f, sb = 0.0f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const uint32_t xl = words[d] & 0x55555555u;
            const uint32_t xh = (words[d] >> 1) & 0x55555555u;
            const uint32_t xb = xl & xh;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float *t = tab + (4 * d + k) * PT_STRIDE;
                slo += t[(xl >> (8 * k)) & 0xFFu];
                shi += t[(xh >> (8 * k)) & 0xFFu];
                sb += t[(xb >> (8 * k)) & 0xFFu];
            }
        }
        const double c3 = (double)sb, c2 = (double)shi - c3, c1 = (double)slo - c3;
        const double c0 = (double)utot - c1 - c2 - c3;
        const float *l = lut + (int64_t)r * 4;
        const double v = (double)l[0] * c0 + (double)l[1] * c1 + (double)l[2] * c2 + (double)l[3] * c3;
        if (v != 0.0) unsafeAtomicAdd(&out[r], v);
    }
}

// ---- sample-major image of the payload for the Z'p half: t32[snp_tile][sample][32 B], 128 consecutive SNPs (of the
// row list) of one sample per record -- the transpose of the P32 image, built once per solve.
// grid (snp tiles, sample tiles), 128 threads: thread j gathers the codes of sample j from the 128 staged records.
__global__ __launch_bounds__(128) void p32_transpose_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                            const int32_t *__restrict__ rows, int nrows, int n,
                                                            uint8_t *__restrict__ t32) {
    __shared__ __attribute__((aligned(16))) uint32_t rec_sh[128][9];   // 8 payload dwords + 1 pad (bank spread)
    const int st = blockIdx.x, tile = blockIdx.y;
    const int j = threadIdx.x;
    {
        const int r = st * 128 + j;
        uint4 a = make_uint4(0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u), b = a;
        if (r < nrows) {
            const int64_t rec = rows ? (int64_t)rows[r] : (int64_t)r;
            const uint4 *pp = reinterpret_cast<const uint4 *>(p32 + ((int64_t)tile * m_total + rec) * 32);
            a = pp[0];
            b = pp[1];
        }
        rec_sh[j][0] = a.x; rec_sh[j][1] = a.y; rec_sh[j][2] = a.z; rec_sh[j][3] = a.w;
        rec_sh[j][4] = b.x; rec_sh[j][5] = b.y; rec_sh[j][6] = b.z; rec_sh[j][7] = b.w;
    }
    __syncthreads();
    const int dw = j >> 4, sh = 2 * (j & 15);
    uint32_t o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) w |= ((rec_sh[16 * d + k][dw] >> sh) & 3u) << (2 * k);
        o[d] = w;
    }
    const int i = tile * 128 + j;
    if (i < n) {
        uint4 *dst = reinterpret_cast<uint4 *>(t32 + ((int64_t)st * n + i) * 32);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

// per-SNP bit-plane weights of w_r[code] = lut[r][code] * p_r (f32 product, as the reference's f32 decode x f32 weight):
// w[code] = w0 + b0 (w1 - w0) + b1 (w2 - w0) + b0 b1 (w3 - w2 - w1 + w0);  wq[r] = (d_lo, d_hi, d_both, w0); sum w0 -> w0sum
__global__ __launch_bounds__(256) void pcg_plane_weights_kernel(const float *__restrict__ lut,
                                                                const double *__restrict__ p, int nrows,
                                                                float4 *__restrict__ wq, double *__restrict__ w0sum) {
    __shared__ double sh[4];
    const int r = blockIdx.x * 256 + threadIdx.x;
    double w0d = 0.0;
    if (r < nrows) {
        const float pr = (float)p[r];
        const float w0 = lut[(int64_t)r * 4 + 0] * pr, w1 = lut[(int64_t)r * 4 + 1] * pr;
        const float w2 = lut[(int64_t)r * 4 + 2] * pr, w3 = lut[(int64_t)r * 4 + 3] * pr;
        const double d3 = ((double)w3 - (double)w2) - (double)w1 + (double)w0;
        wq[r] = make_float4(w1 - w0, w2 - w0, (float)d3, w0);
        w0d = (double)w0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w0d += __shfl_xor(w0d, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w0d;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = sh[0] + sh[1] + sh[2] + sh[3];
        if (t != 0.0) unsafeAtomicAdd(w0sum, t);
    }
}

// out[i] += sum_r w_r[code(r,i)] from the sample-major image: per 128-SNP tile three 16-entry tables per SNP quad
// (lo / hi / both planes), three LDS lookups per four genotypes, f32 partial sums inside a tile, f64 across tiles.
// grid (ceil(n / (256 * PD_SPT)), tile slices), 256 threads, thread = sample (PD_SPT samples each).
constexpr int PD_SPT = 4;
__global__ __launch_bounds__(256) void packed_dot_t32_kernel(const uint8_t *__restrict__ t32, int nrows, int n,
                                                             const float4 *__restrict__ wq,
                                                             const double *__restrict__ w0sum, int tiles_per_slice,
                                                             double *__restrict__ out) {
    __shared__ float tab[3][32 * PT_STRIDE];
    const int tid = threadIdx.x;
    const int nst = (nrows + 127) / 128;
    const int st0 = blockIdx.y * tiles_per_slice;
    const int st1 = (st0 + tiles_per_slice < nst) ? (st0 + tiles_per_slice) : nst;
    double acc[PD_SPT];
#pragma unroll
    for (int s = 0; s < PD_SPT; ++s) acc[s] = 0.0;
    const int i0 = blockIdx.x * 256 * PD_SPT + tid;
    for (int st = st0; st < st1; ++st) {
        __syncthreads();
        for (int e = tid; e < 3 * 32 * 16; e += 256) {
            const int pl = e >> 9, q = (e >> 4) & 31, b = e & 15;
            const int x = (b & 1) | ((b & 2) << 1) | ((b & 4) << 2) | ((b & 8) << 3);
            float t = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = st * 128 + 4 * q + k;
                if (((b >> k) & 1) && r < nrows) {
                    const float4 w = wq[r];
                    t += (pl == 0) ? w.x : (pl == 1 ? w.y : w.z);
                }
            }
            tab[pl][q * PT_STRIDE + x] = t;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < PD_SPT; ++s) {
            const int i = i0 + 256 * s;
            if (i >= n) continue;
            const uint4 *pp = reinterpret_cast<const uint4 *>(t32 + ((int64_t)st * n + i) * 32);
            const uint4 w0 = pp[0], w1 = pp[1];
            const uint32_t words[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            float f = 0.0f;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const uint32_t xl = words[d] & 0x55555555u;
                const uint32_t xh = (words[d] >> 1) & 0x55555555u;
                const uint32_t xb = xl & xh;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int o = (4 * d + k) * PT_STRIDE;
                    f += tab[0][o + ((xl >> (8 * k)) & 0xFFu)];
                    f += tab[1][o + ((xh >> (8 * k)) & 0xFFu)];
                    f += tab[2][o + ((xb >> (8 * k)) & 0xFFu)];
                }
            }
            acc[s] += (double)f;
        }
    }
    const double base = (blockIdx.y == 0) ? w0sum[0] : 0.0;
#pragma unroll
    for (int s = 0; s < PD_SPT; ++s) {
        const int i = i0 + 256 * s;
        if (i < n) {
            const double v = acc[s] + base;
            if (v != 0.0) unsafeAtomicAdd(&out[i], v);
        }
    }
}

// out[i] += sum_r lut[r][code(r,i)] * beta[r]  (`predict_from_effect_stream`, src/stats/gblup.rs:1037-1103; the
// Z v half of the PCG operator).  grid (ntiles, row slices of PD_SLICE); 128 threads x 2 row halves; records and
// the per-row weight tables w[r][c] = lut[r][c] * beta[r] are staged through LDS 64 rows at a time.
constexpr int PD_SLICE = 2048;
__global__ __launch_bounds__(256) void packed_dot_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                         const int32_t *__restrict__ rows, int nrows,
                                                         const float *__restrict__ lut, const double *__restrict__ beta,
                                                         int n, double *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t rec_sh[64][32];
    __shared__ double w_sh[64][4];
    __shared__ double half_sh[128];
    const int tile = blockIdx.x;
    const int r_begin = blockIdx.y * PD_SLICE;
    const int r_end = (r_begin + PD_SLICE < nrows) ? (r_begin + PD_SLICE) : nrows;
    const int tid = threadIdx.x;
    const int j = tid & 127, half = tid >> 7;   // sample within the tile, row parity group
    double acc = 0.0;
    for (int r0 = r_begin; r0 < r_end; r0 += 64) {
        const int cnt = (r_end - r0 < 64) ? (r_end - r0) : 64;
        __syncthreads();
        {   // 64 records x 32 B = 128 x 16 B
            if (tid < 128) {
                const int lr = tid >> 1, part = tid & 1;
                uint4 v = make_uint4(0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u);
                if (lr < cnt) {
                    const int64_t rec = rows ? (int64_t)rows[r0 + lr] : (int64_t)(r0 + lr);
                    v = reinterpret_cast<const uint4 *>(p32 + ((int64_t)tile * m_total + rec) * 32)[part];
                }
                reinterpret_cast<uint4 *>(&rec_sh[lr][0])[part] = v;
            } else {
                const int k = tid - 128;           // 128 threads fill 64 x 4 weights, two each
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = k * 2 + u, lr = e >> 2, c = e & 3;
                    w_sh[lr][c] = (lr < cnt) ? (double)lut[(int64_t)(r0 + lr) * 4 + c] * beta[r0 + lr] : 0.0;
                }
            }
        }
        __syncthreads();
        for (int lr = half; lr < cnt; lr += 2) {
            const uint32_t code = (rec_sh[lr][j >> 2] >> (2 * (j & 3))) & 3u;
            acc += w_sh[lr][code];
        }
    }
    __syncthreads();
    if (half == 1) half_sh[j] = acc;
    __syncthreads();
    if (half == 0) {
        acc += half_sh[j];
        const int i = tile * 128 + j;
        if (i < n && acc != 0.0) unsafeAtomicAdd(&out[i], acc);
    }
}

}  // namespace jx

using namespace jx;

extern "C" int jxg_eigh_f64(double *d_a, int n, double ridge, double *d_w, void *stream);

// d_k (n,n) f64 training GRM (overwritten with U^T), d_yc (n) centred phenotype.
// h_out: [0] lambda, [1] beta_rot, [2] rtv_invr, [3] ml, [4] reml, [5] mean(s).  d_alpha (n).
extern "C" int jxg_gblup_fit(double *d_k, int n, double ridge, const double *d_yc, double low, double high, double tol,
                             int max_iter, double *d_alpha, double *h_out, void *stream) {
    if (n <= 1) return fail("GBLUP REML requires at least 2 training samples.");
    if (!(low < high)) return fail("low/high must be finite and low < high");
    hipStream_t st = (hipStream_t)stream;
    DevBuf s, xr, yr, t, out;
    if (s.alloc(sizeof(double) * n) || xr.alloc(sizeof(double) * n) || yr.alloc(sizeof(double) * n) ||
        t.alloc(sizeof(double) * n) || out.alloc(sizeof(double) * 8))
        return 1;
    if (jxg_eigh_f64(d_k, n, ridge, s.as<double>(), stream)) return 1;
    hipLaunchKernelGGL(gblup_rot_kernel, dim3(n), dim3(SCAN_THREADS), 0, st, d_k, n, d_yc, xr.as<double>(),
                       yr.as<double>());
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(gblup_reml_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, s.as<double>(), xr.as<double>(),
                       yr.as<double>(), n, low, high, tol, max_iter, t.as<double>(), out.as<double>());
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(gblup_alpha_kernel, dim3((n + SCAN_THREADS - 1) / SCAN_THREADS), dim3(SCAN_THREADS), 0, st, d_k,
                       n, t.as<double>(), d_alpha);
    JX_LAUNCH_CHECK();
    double ho[8];
    JX_HIP(hipMemcpyAsync(ho, out.p, sizeof(double) * 6, hipMemcpyDeviceToHost, st));
    std::vector<double> hs((size_t)n);
    JX_HIP(hipMemcpyAsync(hs.data(), s.p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    if (ho[0] != 0.0) return fail("GBLUP REML optimization failed to produce a valid optimum.");
    double sum = 0.0;
    for (int i = 0; i < n; ++i) sum += hs[i];
    for (int i = 0; i < 5; ++i) h_out[i] = ho[i + 1];
    h_out[5] = sum / (double)n;
    return 0;
}


extern "C" int jxg_packed_tdot(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                               const float *d_lut, const double *d_alpha, double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    JX_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)nrows, st));
    dim3 grid((nrows + 255) / 256, (n + 127) / 128);
    hipLaunchKernelGGL(packed_tdot_kernel, grid, dim3(256), 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_alpha, n,
                       d_out);
    JX_LAUNCH_CHECK();
    return 0;
}

namespace jx {
// k_pcg_i8.hip: the same two products on the int8 matrix pipes (exact plane sums of a four-digit image of the vector)
bool pcg_i8_enabled(int n, int nrows);
int packed_tdot_i8(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows, const float *d_lut,
                   const double *d_u, double *d_out);
int packed_dot_t32_i8(hipStream_t st, const uint8_t *d_t32, int n, int nrows, const float *d_lut, const double *d_beta, void *d_work,
                      double *d_out);
}  // namespace jx

extern "C" int jxg_packed_tdot_f32(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                   const float *d_lut, const double *d_u, double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (pcg_i8_enabled(n, nrows)) return packed_tdot_i8(st, d_p32, m_total, n, d_rows, nrows, d_lut, d_u, d_out);
    JX_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)nrows, st));
    dim3 grid((nrows + 256 * PT_RPT - 1) / (256 * PT_RPT), (n + 127) / 128);
    hipLaunchKernelGGL(packed_tdot_f32_kernel, grid, dim3(256), 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_u, n,
                       d_out);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t jxg_t32_bytes(int n, int nrows) { return (int64_t)((nrows + 127) / 128) * (int64_t)n * 32; }

extern "C" int jxg_p32_transpose(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                 uint8_t *d_t32, void *stream) {
    if (nrows <= 0 || n <= 0) return 0;
    dim3 grid((nrows + 127) / 128, (n + 127) / 128);
    hipLaunchKernelGGL(p32_transpose_kernel, grid, dim3(128), 0, (hipStream_t)stream, d_p32, m_total, d_rows, nrows, n,
                       d_t32);
    JX_LAUNCH_CHECK();
    return 0;
}

// d_work: nrows float4 + one double (16 * nrows + 16 bytes)
extern "C" int jxg_packed_dot_t32(const uint8_t *d_t32, int n, int nrows, const float *d_lut, const double *d_beta,
                                  void *d_work, double *d_out, void *stream) {
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (nrows > 0 && pcg_i8_enabled(n, nrows)) return packed_dot_t32_i8(st, d_t32, n, nrows, d_lut, d_beta, d_work, d_out);
    JX_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)n, st));
    if (nrows <= 0) return 0;
    float4 *wq = (float4 *)d_work;
    double *w0sum = (double *)((char *)d_work + sizeof(float4) * (size_t)nrows);
    JX_HIP(hipMemsetAsync(w0sum, 0, sizeof(double), st));
    hipLaunchKernelGGL(pcg_plane_weights_kernel, dim3((nrows + 255) / 256), dim3(256), 0, st, d_lut, d_beta, nrows, wq,
                       w0sum);
    JX_LAUNCH_CHECK();
    const int nst = (nrows + 127) / 128;
    const int gx = (n + 256 * PD_SPT - 1) / (256 * PD_SPT);
    int slices = (1024 + gx - 1) / gx;            // ~1024 workgroups
    if (slices > nst) slices = nst;
    if (slices < 1) slices = 1;
    const int tps = (nst + slices - 1) / slices;
    slices = (nst + tps - 1) / tps;
    hipLaunchKernelGGL(packed_dot_t32_kernel, dim3(gx, slices), dim3(256), 0, st, d_t32, nrows, n, wq, w0sum, tps,
                       d_out);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_packed_dot(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                              const float *d_lut, const double *d_beta, double *d_out, void *stream) {
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    JX_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)n, st));
    if (nrows <= 0) return 0;
    dim3 grid((n + 127) / 128, (nrows + PD_SLICE - 1) / PD_SLICE);
    hipLaunchKernelGGL(packed_dot_kernel, grid, dim3(256), 0, st, d_p32, m_total, d_rows, nrows, d_lut, d_beta, n,
                       d_out);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_cross_dot(const void *d_k, int k_is_f64, int64_t n_full, const int32_t *d_rows, int nrows,
                             const int32_t *d_cols, int ncols, const double *d_alpha, double beta0, double *d_out,
                             void *stream) {
    if (nrows <= 0) return 0;
    if (k_is_f64)
        hipLaunchKernelGGL(cross_dot_kernel<double>, dim3(nrows), dim3(SCAN_THREADS), 0, (hipStream_t)stream,
                           (const double *)d_k, n_full, d_rows, d_cols, ncols, d_alpha, beta0, d_out);
    else
        hipLaunchKernelGGL(cross_dot_kernel<float>, dim3(nrows), dim3(SCAN_THREADS), 0, (hipStream_t)stream,
                           (const float *)d_k, n_full, d_rows, d_cols, ncols, d_alpha, beta0, d_out);
    JX_LAUNCH_CHECK();
    return 0;
}

// `gblup_reml_npy_grm` on an in-memory GRM (host arrays): fit on K[train,train] + g_eps I, predict train/test.
// out_scalars: [0] pve, [1] lambda, [2] ml, [3] reml, [4] sigma_g2, [5] sigma_e2, [6] beta0.
extern "C" int jx_gblup_reml_grm(const void *k_full, int k_is_f64, int64_t n_full, const int64_t *train_idx,
                                 int n_train, const double *y_train, const int64_t *test_idx, int n_test,
                                 double g_eps, double low, double high, int max_iter, double tol, int estimate_only,
                                 double *out_pred_train, double *out_pred_test, double *out_scalars) {
    if (!(std::isfinite(g_eps) && g_eps >= 0.0)) return fail("g_eps must be finite and >= 0");
    if (!(std::isfinite(low) && std::isfinite(high) && low < high)) return fail("low/high must be finite and low < high");
    if (max_iter <= 0) return fail("max_iter must be > 0");
    if (!(std::isfinite(tol) && tol > 0.0)) return fail("tol must be finite and > 0");
    if (n_train <= 1) return fail("GBLUP REML requires at least 2 training samples.");
    const size_t esz = k_is_f64 ? 8 : 4;
    std::vector<int32_t> tr((size_t)n_train), te((size_t)(n_test > 0 ? n_test : 0));
    for (int i = 0; i < n_train; ++i) {
        if (train_idx[i] < 0 || train_idx[i] >= n_full) return fail("train sample index out of range");
        tr[i] = (int32_t)train_idx[i];
    }
    for (int i = 0; i < n_test; ++i) {
        if (test_idx[i] < 0 || test_idx[i] >= n_full) return fail("test sample index out of range");
        te[i] = (int32_t)test_idx[i];
    }
    double y_mean = 0.0;
    for (int i = 0; i < n_train; ++i) y_mean += y_train[i];
    y_mean /= (double)n_train;
    std::vector<double> yc((size_t)n_train);
    for (int i = 0; i < n_train; ++i) yc[i] = y_train[i] - y_mean;

    DevBuf dk, dtr, dte, dkt, dyc, dalpha, dpred;
    if (dk.alloc(esz * (size_t)n_full * (size_t)n_full)) return 1;
    JX_HIP(hipMemcpy(dk.p, k_full, esz * (size_t)n_full * (size_t)n_full, hipMemcpyHostToDevice));
    if (dtr.alloc(sizeof(int32_t) * tr.size()) || dte.alloc(sizeof(int32_t) * (te.size() + 1))) return 1;
    JX_HIP(hipMemcpy(dtr.p, tr.data(), sizeof(int32_t) * tr.size(), hipMemcpyHostToDevice));
    if (!te.empty()) JX_HIP(hipMemcpy(dte.p, te.data(), sizeof(int32_t) * te.size(), hipMemcpyHostToDevice));
    if (dkt.alloc(sizeof(double) * (size_t)n_train * n_train) || dyc.alloc(sizeof(double) * n_train) ||
        dalpha.alloc(sizeof(double) * n_train))
        return 1;
    JX_HIP(hipMemcpy(dyc.p, yc.data(), sizeof(double) * n_train, hipMemcpyHostToDevice));
    if (jxg_gather_sub_f64(dk.p, k_is_f64, (int)n_full, dtr.as<int32_t>(), n_train, dkt.as<double>(), nullptr)) return 1;
    double fit[6];
    if (jxg_gblup_fit(dkt.as<double>(), n_train, g_eps, dyc.as<double>(), low, high, tol, max_iter, dalpha.as<double>(),
                      fit, nullptr))
        return 1;
    const double lambda = fit[0], beta_rot = fit[1], q = fit[2], ml = fit[3], reml = fit[4], mean_s = fit[5];
    const double n_eff = (double)(n_train - 1);
    const double sg2 = q / (n_eff > 1.0 ? n_eff : 1.0);
    const double se2 = lambda * sg2;
    const double var_g = sg2 * (mean_s > 0.0 ? mean_s : 0.0);
    const double den = var_g + se2;
    out_scalars[0] = (std::isfinite(den) && den > 0.0) ? var_g / den : NAN;
    out_scalars[1] = lambda;
    out_scalars[2] = ml;
    out_scalars[3] = reml;
    out_scalars[4] = sg2;
    out_scalars[5] = se2;
    out_scalars[6] = y_mean + beta_rot;
    if (estimate_only) return 0;
    const int np = n_train > n_test ? n_train : n_test;
    if (dpred.alloc(sizeof(double) * (size_t)np)) return 1;
    if (jxg_cross_dot(dk.p, k_is_f64, n_full, dtr.as<int32_t>(), n_train, dtr.as<int32_t>(), n_train, dalpha.as<double>(),
                      out_scalars[6], dpred.as<double>(), nullptr))
        return 1;
    JX_HIP(hipMemcpy(out_pred_train, dpred.p, sizeof(double) * n_train, hipMemcpyDeviceToHost));
    if (n_test > 0) {
        if (jxg_cross_dot(dk.p, k_is_f64, n_full, dte.as<int32_t>(), n_test, dtr.as<int32_t>(), n_train,
                          dalpha.as<double>(), out_scalars[6], dpred.as<double>(), nullptr))
            return 1;
        JX_HIP(hipMemcpy(out_pred_test, dpred.p, sizeof(double) * n_test, hipMemcpyDeviceToHost));
    }
    return 0;
}
