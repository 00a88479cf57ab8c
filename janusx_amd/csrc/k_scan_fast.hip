// Exact per-SNP REML scan, fast formulation.
//
// Same objective, same Brent control flow and same outputs as `lmm_scan_kernel` (k_scan.hip, the line-by-line
// restatement of src/stats/lmm.rs:94-199 / src/stats/reml.rs:255-568), but each objective evaluation touches the
// n rotated samples ONCE and only for the SNP-specific sums:
//
//   REML(x; g) needs, at lambda = 10^x,   A = X'V^-1X,  b = X'V^-1y,  y'V^-1y,  sum ln v      (lambda-only)
//                                          c = X'V^-1g,  g'V^-1g,  g'V^-1y                       (per SNP)
//   and  r'V^-1r = y'V^-1y - 2 beta'b + beta'A0 beta  (A0 = normal matrix without the 1e-6 ridge).
//
// * The lambda-only sums are analytic in x on a strip |Im x| < pi/ln 10 (their only singularities sit at
//   10^x = -s_i), so on segments of width <= 2 a 32-term Chebyshev series reproduces them to ~3e-16 relative
//   (rho = 3.05, rho^-32 = 3e-16).  They are tabulated once per call (`cheb_nodes_kernel`, 32 nodes per segment,
//   exact f64 sums) and evaluated by Clenshaw recurrences inside the scan.
// * y is shifted by the null GLS fit at the interval midpoint (y_c = y - X beta_mid): REML, beta_g and SE are
//   invariant to it (X spans the shift) and the moment form of r'V^-1r no longer cancels a large mean.
// * 1/(s_i + lambda): v_rcp_f64 seed + two Newton steps (<= 1 ulp), instead of the IEEE division sequence.
// One wave per SNP, no LDS, no barriers.  `jxg_lmm_scan_exact` keeps the two-pass reference formulation.
#include <stdlib.h>

#include <mutex>

#include "scan_common.h"

namespace jx {

constexpr int CH_N = 32;          // Chebyshev terms per segment
constexpr int CH_MAXSEG = 8;      // (high - low) <= 16
constexpr int CH_HDR = 24;        // doubles in front of y_c in the table workspace: smin, 7 spare, beta_mid (<= 15), 1 spare
// per-SNP series of the SNP-specific sums (series_coef_kernel below)
constexpr int SR_M = 64;            // series entries per quantity: <= 2 segments x 32 terms
constexpr int SR_SC = 64;           // samples per staged chunk
constexpr int SR_WP = SR_M + 16;    // LDS pitch of the table chunk [sample][entry] (doubles): the 4 k-rows of a fragment read land
                                    // on disjoint banks
constexpr int SR_GP = SR_SC + 4;    // LDS pitch of the row chunk [SNP][sample] (floats): bank = 4 snp + sample

struct ChebHeader {
    int nseg;
    int nf;       // number of tabulated functions: 1 + NAc + p + 1 (plain form), 4 + 2 p + NAc (block form)
    double low;
    double segw;
};

// Block form (dim = p + 1 >= 5, i.e. the MAXD = 8 / 16 instantiations).  The normal matrix of an evaluation is
//   A + eps I = [ Ac  c ; c'  d + eps ],   Ac = X~'WX~ + eps I = L L' (lambda-only),   c = X~'Wg~, d = g~'Wg~ (per SNP),
// so the Cholesky factorisation of the full matrix is L, the row w' = (L^-1 c)' and the last pivot sk = d + eps - w'w: with L
// tabulated (its entries are as analytic in x as the sums they come from) an evaluation needs two triangular solves instead
// of a factorisation and no dim x dim matrix in registers:
//   w = L^-1 c,  t = L^-1 bc (tabulated; bc = X~'Wy~),  beta_k = (bk - w't) / sk,  log det = 2 sum ln L_ii + ln sk,
//   r'V^-1 r = (yy - t't) - beta_k^2 sk - eps |beta|^2,  |beta|^2 = |u|^2 - 2 beta_k u'z + beta_k^2 (|z|^2 + 1)  with
//   u = L^-T t (tabulated), z = L^-T w,   [(A + eps I)^-1]_kk = 1 / sk.
// (Tabulating M = Ac^-1 instead and forming c'Mc is NOT an option: an error of 1e-9 |M| in M is amplified by |c|^2 |M| /
// c'Mc -- two covariates collinear to 1e-3 cost 4e-5 on beta that way; a perturbed L is a small relative perturbation of Ac.)
// Tabulated per node: f0 = sum ln v, f1 = log det Ac, f2 = yy - t't, f3 = |u|^2, then t (p), u (p), the lower triangle of L.
// The plain form keeps two dim x dim matrices per lane in registers: 2.2 KB of scratch per lane at MAXD = 16, and the
// one-wave-per-SNP kernel was all that could take it (measured 2.36 s per 100 000 SNPs with ten covariates at n = 20 000).
__host__ __device__ __forceinline__ int cheb_nf(int p, bool blk) {
    return blk ? 4 + 2 * p + p * (p + 1) / 2 : 1 + p * (p + 1) / 2 + p + 1;
}
__host__ __device__ __forceinline__ bool cheb_blk(int p) { return p + 1 >= 5; }

__device__ __forceinline__ double fast_rcp(double v) {
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}

// Value of lane `src` (WAVE-UNIFORM index) in every lane: two v_readlane_b32 through SGPRs instead of a ds_bpermute round trip
// per dword -- an evaluation of the tabulated form reads a dozen such values, each on its critical path.
__device__ __forceinline__ double wave_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// y_c = y - X beta_mid, beta_mid = GLS fit of the null model at lambda_mid (single workgroup).
template <int MAXD>
__global__ __launch_bounds__(SCAN_THREADS) void yshift_kernel(const double *__restrict__ s,
                                                              const double *__restrict__ xcov,
                                                              const double *__restrict__ y, int n, int p, double lbd,
                                                              double *__restrict__ yc, double *__restrict__ smin_out) {
    constexpr int NA = MAXD * (MAXD + 1) / 2;
    constexpr int NV = NA + MAXD;
    __shared__ double shm[SCAN_WAVES * NV];
    __shared__ double smin_sh[SCAN_WAVES];
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
    double smin = 1e300;
    for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
        const double si = s[i];
        smin = fmin(smin, si);
        const double vi = 1.0 / (si + lbd);
        const double yi = y[i];
        double xr[MAXD];
#pragma unroll
        for (int r = 0; r < MAXD; ++r) xr[r] = (r < p) ? xcov[(int64_t)i * p + r] : 0.0;
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
    }
    block_sum<NV>(v, NV, shm);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) smin = fmin(smin, __shfl_xor(smin, off, 64));
    if ((threadIdx.x & 63) == 0) smin_sh[threadIdx.x >> 6] = smin;
    __syncthreads();
    double a[MAXD * MAXD], b[MAXD], beta[MAXD];
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
    const bool ok = chol_inplace<MAXD>(a, p);
    chol_solve<MAXD>(a, p, b, beta);
    for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
        double xb = 0.0;
#pragma unroll
        for (int r = 0; r < MAXD; ++r)
            if (r < p) xb += xcov[(int64_t)i * p + r] * beta[r];
        yc[i] = ok ? (y[i] - xb) : y[i];
    }
    if (threadIdx.x == 0) {
        double m = smin_sh[0];
        for (int w = 1; w < SCAN_WAVES; ++w) m = fmin(m, smin_sh[w]);
        smin_out[0] = m;
        // the shift itself: with the 1e-6 ridge the SNP coefficient is invariant to it only up to eps [N beta_mid]_k
        // (N = (A + eps I)^-1) -- invisible for a well-conditioned design, 4e-5 of a standard error with two covariates
        // collinear to 1e-3; the final evaluation adds it back (fast_eval_finish*)
#pragma unroll
        for (int r = 0; r < MAXD; ++r)
            if (r < p) smin_out[8 + r] = ok ? beta[r] : 0.0;
    }
}

// One workgroup per Chebyshev node: exact f64 sums of the lambda-only functions at x_k.
// vals[(seg * nf + f) * CH_N + k];  f = 0: sum ln v; 1..NAc: lower triangle of X'V^-1X; then X'V^-1 y_c; y_c'V^-1y_c.
template <int MAXD>
__global__ __launch_bounds__(SCAN_THREADS) void cheb_nodes_kernel(const double *__restrict__ s,
                                                                  const double *__restrict__ xcov,
                                                                  const double *__restrict__ yc, int n, int p,
                                                                  double low, double segw, int nf,
                                                                  double *__restrict__ vals) {
    constexpr int NA = MAXD * (MAXD + 1) / 2;
    constexpr int NV = NA + MAXD + 2;
    __shared__ double shm[SCAN_WAVES * NV];
    const int seg = blockIdx.x / CH_N, k = blockIdx.x % CH_N;
    const double t = cos(M_PI * ((double)k + 0.5) / (double)CH_N);
    const double x = low + segw * ((double)seg + 0.5) + 0.5 * segw * t;
    const double lbd = pow(10.0, x);
    double v[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = 0.0;
    for (int i = threadIdx.x; i < n; i += SCAN_THREADS) {
        const double vv = s[i] + lbd;
        const double vi = 1.0 / vv;
        const double yi = yc[i];
        double xr[MAXD];
#pragma unroll
        for (int r = 0; r < MAXD; ++r) xr[r] = (r < p) ? xcov[(int64_t)i * p + r] : 0.0;
        v[NA + MAXD] += log(vv);
        v[NA + MAXD + 1] += vi * yi * yi;
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
    }
    block_sum<NV>(v, NV, shm);
    if (cheb_blk(p)) {
        // block form: M = (X~'WX~ + eps I)^-1 and what hangs on it, by thread 0 on shared arrays (run-time loops: p <= 15)
        __shared__ double am[MAXD * MAXD], bm[MAXD], um[MAXD], col[MAXD];
        if (threadIdx.x == 0) {
            int idx = 0;
#pragma unroll
            for (int r = 0; r < MAXD; ++r) {
                bm[r] = v[NA + r];
#pragma unroll
                for (int c = 0; c <= r; ++c) {
                    am[r * MAXD + c] = v[idx];
                    am[c * MAXD + r] = v[idx];
                    ++idx;
                }
            }
            const double yy = v[NA + MAXD + 1];
            bool ok = true;
            for (int i = 0; i < p; ++i) am[i * MAXD + i] += 1e-6;
            for (int i = 0; i < p; ++i)
                for (int j = 0; j <= i; ++j) {
                    double sum = am[i * MAXD + j];
                    for (int q = 0; q < j; ++q) sum -= am[i * MAXD + q] * am[j * MAXD + q];
                    if (i == j) {
                        if (!(sum > 1e-18)) ok = false;
                        am[i * MAXD + i] = sqrt(sum);
                    } else {
                        am[i * MAXD + j] = sum / am[j * MAXD + j];
                    }
                }
            double lnd = 0.0;
            for (int i = 0; i < p; ++i) lnd += 2.0 * log(am[i * MAXD + i]);
            // t = L^-1 bc (into col), u = L^-T t (into um)
            for (int i = 0; i < p; ++i) {
                double sum = bm[i];
                for (int q = 0; q < i; ++q) sum -= am[i * MAXD + q] * col[q];
                col[i] = sum / am[i * MAXD + i];
            }
            for (int i = p - 1; i >= 0; --i) {
                double sum = col[i];
                for (int q = i + 1; q < p; ++q) sum -= am[q * MAXD + i] * um[q];
                um[i] = sum / am[i * MAXD + i];
            }
            double tt = 0.0, uu = 0.0;
            for (int i = 0; i < p; ++i) {
                tt += col[i] * col[i];
                uu += um[i] * um[i];
            }
            const double bad = nan("");
            double *o = vals + (int64_t)seg * nf * CH_N;
            o[0 * CH_N + k] = v[NA + MAXD];
            o[1 * CH_N + k] = ok ? lnd : bad;
            o[2 * CH_N + k] = ok ? yy - tt : bad;
            o[3 * CH_N + k] = ok ? uu : bad;
            for (int i = 0; i < p; ++i) {
                o[(4 + i) * CH_N + k] = ok ? col[i] : bad;
                o[(4 + p + i) * CH_N + k] = ok ? um[i] : bad;
            }
            for (int r = 0; r < p; ++r)
                for (int c = 0; c <= r; ++c) o[(4 + 2 * p + r * (r + 1) / 2 + c) * CH_N + k] = ok ? am[r * MAXD + c] : bad;
        }
        return;
    }
    if (threadIdx.x == 0) {
        double *o = vals + (int64_t)seg * nf * CH_N;
        o[0 * CH_N + k] = v[NA + MAXD];
        int f = 1;
        int idx = 0;
#pragma unroll
        for (int r = 0; r < MAXD; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                if (r < p) {
                    o[f * CH_N + k] = v[idx];
                    ++f;
                }
                ++idx;
            }
#pragma unroll
        for (int r = 0; r < MAXD; ++r)
            if (r < p) {
                o[f * CH_N + k] = v[NA + r];
                ++f;
            }
        o[f * CH_N + k] = v[NA + MAXD + 1];
    }
}

// node values -> Chebyshev coefficients (DCT-II), one thread per (seg, f, j)
__global__ void cheb_coef_kernel(const double *__restrict__ vals, int total_funcs, double *__restrict__ coef) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_funcs * CH_N) return;
    const int fn = gid / CH_N, j = gid % CH_N;
    const double *v = vals + (int64_t)fn * CH_N;
    double acc = 0.0;
    for (int k = 0; k < CH_N; ++k) acc += v[k] * cos(M_PI * (double)j * ((double)k + 0.5) / (double)CH_N);
    coef[gid] = acc * (2.0 / (double)CH_N);
}

__device__ __forceinline__ double clenshaw(const double *__restrict__ c, double t) {
    double b1 = 0.0, b2 = 0.0;
    const double t2 = 2.0 * t;
#pragma unroll 8
    for (int j = CH_N - 1; j >= 1; --j) {
        const double b0 = fma(t2, b1, c[j] - b2);
        b2 = b1;
        b1 = b0;
    }
    return fma(t, b1, 0.5 * c[0] - b2);
}

template <int MAXD>
struct FastEval {
    bool ok;
    double reml_neg;  // Brent objective (-REML), 1e8 on failure
    double q, logdetv, beta_k, ainv_kk;
};

// lambda of an evaluation point, or a negative value when the evaluation fails before any sum is formed
// (src/stats/reml.rs:255-362: non-finite / non-positive lambda, n <= dim, a non-positive s_i + lambda)
__device__ __forceinline__ double fast_eval_lambda(double x, double smin, int n, int dim) {
    const double lbd = jx_pow10(x);
    if (!isfinite(lbd) || lbd <= 0.0 || n <= dim) return -1.0;
    if (smin + lbd <= 0.0) return -1.0;
    return lbd;
}

// SNP-specific sums over the samples [i0, i1) of one lane's stride: acc[r] += g x_r / v, acc[MAXD-1] += g^2 / v,
// acc[MAXD] += g y_c / v; s / xcov / yc are indexed from `base` (0 for whole vectors, the tile start for an LDS tile)
template <int MAXD>
__device__ __forceinline__ void fast_eval_accumulate(double lbd, const double *__restrict__ s,
                                                     const double *__restrict__ xcov, const double *__restrict__ yc,
                                                     const float *__restrict__ g, int i0, int i1, int base, int p,
                                                     double (&acc)[MAXD + 1]) {
    const int lane = threadIdx.x & 63;
#pragma unroll 4
    for (int i = i0 + lane; i < i1; i += 64) {
        const int j = i - base;
        const double vi = fast_rcp(s[j] + lbd);
        const double gi = (double)g[i];
        const double gv = gi * vi;
#pragma unroll
        for (int r = 0; r < MAXD - 1; ++r)
            if (r < p) acc[r] = fma(gv, xcov[(int64_t)j * p + r], acc[r]);
        acc[MAXD - 1] = fma(gv, gi, acc[MAXD - 1]);
        acc[MAXD] = fma(gv, yc[j], acc[MAXD]);
    }
}

// the same sums for the tiled kernel.  The rotated row comes from HBM / L2 with nothing to cover its latency but the
// bytes in flight, so a lane takes FOUR consecutive samples per load (16 bytes; 1 KB per wave instruction) and keeps B
// loads in flight; V4 = false is the scalar form (rows not 16-byte aligned: n % 4 != 0).  Optionally the sum of squares of
// the row in the same pass.  (Per-lane sample order differs from the one-wave-per-SNP kernels: results agree to rounding.)
// PERM: the LDS images are stored so that the samples 4 L + e of the 64 lanes are consecutive words for fixed e (position
// 256 G + 64 e + L for sample 256 G + 4 L + e): with the natural order the 8-byte reads of a wave were 32 bytes apart, i.e.
// 4-way bank conflicted -- the reads, three per sample and evaluation, were what bounded the kernel.
template <int MAXD, int B, bool V4, bool PERM = false>
__device__ __forceinline__ void fast_eval_accumulate_batched(double lbd, const double *__restrict__ s,
                                                             const double *__restrict__ xcov, const double *__restrict__ yc,
                                                             const float *__restrict__ g, int i0, int i1, int base, int p,
                                                             double (&acc)[MAXD + 1], double &ssq, bool want_ssq,
                                                             int xs_j = -1, int xs_r = 1) {
    // X~ element (sample j, covariate r) at xcov[j * xs_j + r * xs_r]: row-major per sample by default (xs_j = p), covariate-
    // major in the tiled kernel's LDS image (xs_j = 1, xs_r = tile: a lane's consecutive samples are then consecutive words
    // like s and y~ -- with the per-sample layout the reads of a wave were 4-way bank conflicted from three covariates on)
    if (xs_j < 0) xs_j = p;
    const int lane = threadIdx.x & 63;
    constexpr int W = V4 ? 4 : 1;
    for (int i = i0 + W * lane; i < i1; i += 64 * W * B) {
        float gb[B][W];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int idx = i + 64 * W * u;
            if (V4) {
                const float4 q = (idx < i1) ? *reinterpret_cast<const float4 *>(g + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
                gb[u][0] = q.x;
                gb[u][W > 1 ? 1 : 0] = q.y;
                gb[u][W > 2 ? 2 : 0] = q.z;
                gb[u][W > 3 ? 3 : 0] = q.w;
            } else {
                gb[u][0] = (idx < i1) ? g[idx] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int idx = i + 64 * W * u;
            if (idx < i1) {
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    // PERM: (i - base) - 4 lane is a multiple of 256, so the position is one lane-dependent base per loop
                    // iteration plus the compile-time offset 256 u + 64 e (an immediate of the LDS instruction)
                    const int j = PERM ? ((i - base) - 3 * lane + 256 * u + 64 * e) : (idx + e - base);
                    const double gi = (double)gb[u][e];
                    if (want_ssq) ssq += gi * gi;
                    if (lbd >= 0.0) {
                        const double vi = fast_rcp(s[j] + lbd);
                        const double gv = gi * vi;
#pragma unroll
                        for (int r = 0; r < MAXD - 1; ++r)
                            if (r < p) acc[r] = fma(gv, xcov[(int64_t)j * xs_j + (int64_t)r * xs_r], acc[r]);
                        acc[MAXD - 1] = fma(gv, gi, acc[MAXD - 1]);
                        acc[MAXD] = fma(gv, yc[j], acc[MAXD]);
                    }
                }
            }
        }
    }
}

// PRESUMMED: acc already holds the wave-wide values in every lane (the series form: a broadcast, not a sum)
template <int MAXD, bool PRESUMMED = false>
__device__ __forceinline__ void fast_eval_finish(double x, const ChebHeader hd, const double *__restrict__ coef, int n, int p,
                                                 bool want_ainv, const double *__restrict__ bmid, double (&acc)[MAXD + 1],
                                                 FastEval<MAXD> &o);

// One evaluation for the SNP owned by this wave. MAXD bounds dim = p + 1.
template <int MAXD>
__device__ __forceinline__ void fast_eval(double x, const ChebHeader hd, const double *__restrict__ coef,
                                          double smin, const double *__restrict__ s,
                                          const double *__restrict__ xcov, const double *__restrict__ yc,
                                          const float *__restrict__ g, int n, int p, bool want_ainv,
                                          FastEval<MAXD> &o, const double *__restrict__ bmid = nullptr) {
    const int dim = p + 1;
    o.ok = false;
    o.reml_neg = 1e8;
    o.q = 0.0;
    o.logdetv = 0.0;
    o.beta_k = 0.0;
    o.ainv_kk = 0.0;
    const double lbd = fast_eval_lambda(x, smin, n, dim);
    if (lbd < 0.0) return;
    double acc[MAXD + 1];
#pragma unroll
    for (int k = 0; k < MAXD + 1; ++k) acc[k] = 0.0;
    fast_eval_accumulate<MAXD>(lbd, s, xcov, yc, g, 0, n, 0, p, acc);
    fast_eval_finish<MAXD>(x, hd, coef, n, p, want_ainv, bmid, acc, o);
}

// Second half of an evaluation: wave sums of the per-lane accumulators, lambda-only sums from the Chebyshev tables,
// normal equations, REML.  `o` must have been reset by the caller.
// Block form of the second half (see ChebHeader): two triangular solves with the tabulated factor, no dim x dim matrix in
// registers.  The tabulated values live one per lane (function f in lane f % 64, slot f / 64) and are read with a uniform
// source lane; every lane carries the same w / z (wave-uniform arithmetic).
template <int MAXD, bool PRESUMMED = false>
__device__ __forceinline__ void fast_eval_finish_blk(double x, const ChebHeader hd, const double *__restrict__ coef, int n,
                                                     int p, bool want_ainv, const double *__restrict__ bmid,
                                                     double (&acc)[MAXD + 1], FastEval<MAXD> &o) {
    const int dim = p + 1;
    const int lane = threadIdx.x & 63;
    if (!PRESUMMED) {
#pragma unroll
        for (int k = 0; k < MAXD + 1; ++k) acc[k] = wave_allsum(acc[k]);
    }
    int seg = (int)((x - hd.low) / hd.segw);
    if (seg < 0) seg = 0;
    if (seg >= hd.nseg) seg = hd.nseg - 1;
    const double t = (x - (hd.low + hd.segw * ((double)seg + 0.5))) / (0.5 * hd.segw);
    const double *cf = coef + (int64_t)seg * hd.nf * CH_N;
    constexpr int NS = (4 + 2 * (MAXD - 1) + (MAXD - 1) * MAXD / 2 + 63) / 64;   // table slots per lane
    double mv[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int f = lane + 64 * sl;
        mv[sl] = (64 * sl < hd.nf) ? clenshaw(cf + (int64_t)(f < hd.nf ? f : 0) * CH_N, t) : 0.0;
    }
    auto tab = [&](int f) -> double {                     // f is wave-uniform
        double r = 0.0;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            const double v = wave_bcast(mv[sl], f & 63);
            if ((f >> 6) == sl) r = v;
        }
        return r;
    };
    const double logdetv = tab(0), lnd = tab(1), q0 = tab(2), uu = tab(3);
    const int ft = 4, fu = 4 + p, fl = 4 + 2 * p;
    // w = L^-1 c (forward), then z = L^-T w (backward); c_r = acc[r]
    double w[MAXD - 1], z[MAXD - 1];
    double ww = 0.0, wt = 0.0;
#pragma unroll
    for (int i = 0; i < MAXD - 1; ++i) {
        w[i] = 0.0;
        if (i < p) {
            double sum = acc[i];
#pragma unroll
            for (int q = 0; q < i; ++q) sum = fma(-tab(fl + i * (i + 1) / 2 + q), w[q], sum);
            w[i] = sum / tab(fl + i * (i + 1) / 2 + i);
            ww = fma(w[i], w[i], ww);
            wt = fma(w[i], tab(ft + i), wt);
        }
    }
    double uz = 0.0, zz = 0.0;
#pragma unroll
    for (int ii = 0; ii < MAXD - 1; ++ii) {
        const int i = MAXD - 2 - ii;
        z[i] = 0.0;
        if (i < p) {
            double sum = w[i];
#pragma unroll
            for (int q = i + 1; q < MAXD - 1; ++q)
                if (q < p) sum = fma(-tab(fl + q * (q + 1) / 2 + i), z[q], sum);
            z[i] = sum / tab(fl + i * (i + 1) / 2 + i);
            uz = fma(tab(fu + i), z[i], uz);
            zz = fma(z[i], z[i], zz);
        }
    }
    const double sk = acc[MAXD - 1] + 1e-6 - ww;          // the last pivot of the Cholesky factorisation, squared
    if (!(sk > 1e-18) || !isfinite(lnd)) return;          // chol_inplace would have failed
    const double bk = (acc[MAXD] - wt) / sk;
    const double nb2 = uu - 2.0 * bk * uz + bk * bk * (zz + 1.0);
    const double q = q0 - bk * bk * sk - 1e-6 * nb2;
    const double nf = (double)n, pf = (double)dim;
    const double total = (nf - pf) * jx_log(q) + logdetv + (lnd + jx_log(sk));
    const double cst = (nf - pf) * (jx_log(nf - pf) - 1.0 - jx_log(2.0 * M_PI)) / 2.0;
    const double reml = cst - 0.5 * total;
    o.ok = true;
    o.reml_neg = isfinite(reml) ? -reml : 1e8;
    o.q = q;
    o.logdetv = logdetv;
    o.beta_k = bk;
    o.ainv_kk = 1.0 / sk;
    if (want_ainv && bmid) {
        // y was shifted by X beta_mid: beta_k(y) = beta_k(y_c) - eps [N beta_mid]_k, and the k-th row of N is [-z' / sk, 1 / sk]
        double zb = 0.0;
#pragma unroll
        for (int i = 0; i < MAXD - 1; ++i)
            if (i < p) zb = fma(z[i], bmid[i], zb);
        o.beta_k = bk + 1e-6 * zb / sk;
    }
}

template <int MAXD, bool PRESUMMED>
__device__ __forceinline__ void fast_eval_finish(double x, const ChebHeader hd, const double *__restrict__ coef, int n, int p,
                                                 bool want_ainv, const double *__restrict__ bmid, double (&acc)[MAXD + 1],
                                                 FastEval<MAXD> &o) {
    if constexpr (MAXD >= 8) {                            // dim >= 5: block form (the tables were built for it: cheb_blk)
        fast_eval_finish_blk<MAXD, PRESUMMED>(x, hd, coef, n, p, want_ainv, bmid, acc, o);
        return;
    }
    const int dim = p + 1;
    const int lane = threadIdx.x & 63;
    if (!PRESUMMED) {
#pragma unroll
        for (int k = 0; k < MAXD + 1; ++k) acc[k] = wave_allsum(acc[k]);
    }

    // ---- lambda-only sums from the Chebyshev tables ---------------------------------------------------
    int seg = (int)((x - hd.low) / hd.segw);
    if (seg < 0) seg = 0;
    if (seg >= hd.nseg) seg = hd.nseg - 1;
    const double t = (x - (hd.low + hd.segw * ((double)seg + 0.5))) / (0.5 * hd.segw);
    const double *cf = coef + (int64_t)seg * hd.nf * CH_N;
    // lane f evaluates tabulated function f (nf <= 64 is guaranteed by the host wrapper), results are then
    // read across lanes: one 32-term Clenshaw per evaluation instead of nf of them.
    const double myval = clenshaw(cf + (lane < hd.nf ? lane : 0) * CH_N, t);
    const double logdetv = wave_bcast(myval, 0);
    double a0[MAXD * MAXD], a[MAXD * MAXD], b[MAXD], beta[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD * MAXD; ++k) a0[k] = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD - 1; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            const double val = wave_bcast(myval, 1 + r * (r + 1) / 2 + c);
            if (r < p) {
                a0[r * MAXD + c] = val;
                a0[c * MAXD + r] = val;
            }
        }
    const int fb = 1 + p * (p + 1) / 2;
#pragma unroll
    for (int r = 0; r < MAXD; ++r) b[r] = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD - 1; ++r) {
        const double val = wave_bcast(myval, fb + r);
        if (r < p) b[r] = val;
    }
    const double yy = wave_bcast(myval, fb + p);
    // SNP row/column at index p (compile-time positions via select)
#pragma unroll
    for (int r = 0; r < MAXD; ++r) {
#pragma unroll
        for (int c = 0; c < MAXD; ++c) {
            if (r == p && c < p) a0[r * MAXD + c] = acc[c < MAXD - 1 ? c : 0];
            if (c == p && r < p) a0[r * MAXD + c] = acc[r < MAXD - 1 ? r : 0];
            if (r == p && c == p) a0[r * MAXD + c] = acc[MAXD - 1];
        }
        if (r == p) b[r] = acc[MAXD];
    }
#pragma unroll
    for (int k = 0; k < MAXD * MAXD; ++k) a[k] = a0[k];
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
        if (r < dim) a[r * MAXD + r] += 1e-6;
    if (!chol_inplace<MAXD>(a, dim)) return;
    chol_solve<MAXD>(a, dim, b, beta);
    // r'V^-1 r = yy - 2 beta'b + beta'A0 beta
    double bb = 0.0, bab = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD; ++r) {
        if (r < dim) {
            bb += beta[r] * b[r];
            double row = 0.0;
#pragma unroll
            for (int c = 0; c < MAXD; ++c)
                if (c < dim) row += a0[r * MAXD + c] * beta[c];
            bab += beta[r] * row;
        }
    }
    const double q = yy - 2.0 * bb + bab;
    double ld = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
        if (r < dim) ld += jx_log(a[r * MAXD + r]);
    const double nf = (double)n, pf = (double)dim;
    const double total = (nf - pf) * jx_log(q) + logdetv + 2.0 * ld;
    const double cst = (nf - pf) * (jx_log(nf - pf) - 1.0 - jx_log(2.0 * M_PI)) / 2.0;
    const double reml = cst - 0.5 * total;
    o.ok = true;
    o.reml_neg = isfinite(reml) ? -reml : 1e8;
    o.q = q;
    o.logdetv = logdetv;
    o.beta_k = pick<MAXD>(beta, dim - 1);
    if (want_ainv) {
        double e[MAXD], xk[MAXD];
#pragma unroll
        for (int r = 0; r < MAXD; ++r) e[r] = (r == dim - 1) ? 1.0 : 0.0;
        chol_solve<MAXD>(a, dim, e, xk);
        o.ainv_kk = pick<MAXD>(xk, dim - 1);
        if (bmid) {
            // y was shifted by X beta_mid: beta_k(y) = beta_k(y_c) - eps [N beta_mid]_k = beta_k(y_c) - eps beta_mid'(N e_k)
            double corr = 0.0;
#pragma unroll
            for (int r = 0; r < MAXD - 1; ++r)
                if (r < p) corr = fma(bmid[r], xk[r], corr);
            o.beta_k -= 1e-6 * corr;
        }
    }
}

// ---- Brent on an interpolant of the objective (round 6) -----------------------------------------------------------------
// An evaluation of the tabulated form is ~1300 dependent instructions of wave-UNIFORM arithmetic (35 f64 divisions, four log /
// pow calls, two 32-term recurrences): every lane computes the same number, and a wave issues them one behind the other -- 4 us
// per evaluation when nothing else hides it (the chain scan: measured 663 ms for 20 chains of 10 000 SNPs at configs[2]).  The
// objective -REML(x; g) of one SNP is as analytic in x as the sums it is made of, so: evaluate it ONCE at the 32 Chebyshev nodes
// of each width-2 segment with ONE LANE PER NODE (the same instructions, 64 different x), turn the 64 values into the
// Chebyshev coefficients of the objective (a 32-point cosine transform per segment through LDS) and let Brent run on that series:
// an evaluation is then one Clenshaw recurrence.  Measured on the host (CPU restatement of the likelihood, 60 SNPs, both segments): the interpolant
// reproduces the objective to 3.7e-15 relative -- below the rounding noise of a direct evaluation.  beta / SE at the optimum
// still come from ONE direct evaluation (final_beta_se).  A node that fails (no positive pivot, lambda outside the table) sends
// the SNP back to direct evaluations.
template <int MAXD>
struct InterpTv {
    static constexpr int N = 1 + (MAXD - 1) * MAXD / 2 + (MAXD - 1) + 1;     // table functions of the plain form at p = MAXD - 1
};
template <int N>
__device__ __forceinline__ double pick_n(const double (&v)[N], int idx) {
    double r = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k)
        if (k == idx) r = v[k];
    return r;
}
// -REML at x from per-lane values: tv[f] = tabulated function f at x (lambda-only sums), acc = the SNP-specific sums at x.
// The arithmetic of fast_eval_finish's plain form, on values that differ from lane to lane.
template <int MAXD>
__device__ __forceinline__ double plain_objective(int n, int p, const double (&tv)[InterpTv<MAXD>::N], const double (&acc)[MAXD + 1]) {
    const int dim = p + 1;
    const double logdetv = tv[0];
    double a0[MAXD * MAXD], a[MAXD * MAXD], b[MAXD], beta[MAXD];
#pragma unroll
    for (int k = 0; k < MAXD * MAXD; ++k) a0[k] = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD - 1; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c)
            if (r < p) {
                a0[r * MAXD + c] = tv[1 + r * (r + 1) / 2 + c];
                a0[c * MAXD + r] = tv[1 + r * (r + 1) / 2 + c];
            }
    const int fb = 1 + p * (p + 1) / 2;
#pragma unroll
    for (int r = 0; r < MAXD; ++r) b[r] = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD - 1; ++r)
        if (r < p) b[r] = pick_n(tv, fb + r);
    const double yy = pick_n(tv, fb + p);
#pragma unroll
    for (int r = 0; r < MAXD; ++r) {
#pragma unroll
        for (int c = 0; c < MAXD; ++c) {
            if (r == p && c < p) a0[r * MAXD + c] = acc[c < MAXD - 1 ? c : 0];
            if (c == p && r < p) a0[r * MAXD + c] = acc[r < MAXD - 1 ? r : 0];
            if (r == p && c == p) a0[r * MAXD + c] = acc[MAXD - 1];
        }
        if (r == p) b[r] = acc[MAXD];
    }
#pragma unroll
    for (int k = 0; k < MAXD * MAXD; ++k) a[k] = a0[k];
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
        if (r < dim) a[r * MAXD + r] += 1e-6;
    if (!chol_inplace<MAXD>(a, dim)) return 1e8;
    chol_solve<MAXD>(a, dim, b, beta);
    double bb = 0.0, bab = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD; ++r) {
        if (r < dim) {
            bb += beta[r] * b[r];
            double row = 0.0;
#pragma unroll
            for (int c = 0; c < MAXD; ++c)
                if (c < dim) row += a0[r * MAXD + c] * beta[c];
            bab += beta[r] * row;
        }
    }
    const double q = yy - 2.0 * bb + bab;
    double ld = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD; ++r)
        if (r < dim) ld += jx_log(a[r * MAXD + r]);
    const double nf = (double)n, pf = (double)dim;
    const double total = (nf - pf) * jx_log(q) + logdetv + 2.0 * ld;
    const double cst = (nf - pf) * (jx_log(nf - pf) - 1.0 - jx_log(2.0 * M_PI)) / 2.0;
    const double reml = cst - 0.5 * total;
    return isfinite(reml) ? -reml : 1e8;
}
// The same for the block form (dim 5 - 8: see ChebHeader): the tabulated Cholesky factor L of the covariate block, t = L^-1 bc,
// u = L^-T t and the scalars, each taken by its own Clenshaw recurrence at this lane's x (cf = the lane's segment of the
// coefficient tables, in LDS); the arithmetic of fast_eval_finish_blk on per-lane values.
template <int MAXD>
__device__ __forceinline__ double blk_objective(int n, int p, const double *__restrict__ cf, double t, const double (&acc)[MAXD + 1]) {
    const int dim = p + 1;
    auto tabv = [&](int f) -> double { return clenshaw(cf + (int64_t)f * CH_N, t); };
    const double logdetv = tabv(0), lnd = tabv(1), q0 = tabv(2), uu = tabv(3);
    const int ft = 4, fu = 4 + p, fl = 4 + 2 * p;
    double w[MAXD - 1], z[MAXD - 1], ld[MAXD - 1], lo[(MAXD - 1) * (MAXD - 2) / 2 + 1];
    double ww = 0.0, wt = 0.0;
#pragma unroll
    for (int i = 0; i < MAXD - 1; ++i) {
        w[i] = 0.0;
        ld[i] = 1.0;
        if (i < p) {
            double sum = acc[i];
#pragma unroll
            for (int q = 0; q < i; ++q) {
                lo[i * (i - 1) / 2 + q] = tabv(fl + i * (i + 1) / 2 + q);
                sum = fma(-lo[i * (i - 1) / 2 + q], w[q], sum);
            }
            ld[i] = tabv(fl + i * (i + 1) / 2 + i);
            w[i] = sum / ld[i];
            ww = fma(w[i], w[i], ww);
            wt = fma(w[i], tabv(ft + i), wt);
        }
    }
    double uz = 0.0, zz = 0.0;
#pragma unroll
    for (int ii = 0; ii < MAXD - 1; ++ii) {
        const int i = MAXD - 2 - ii;
        z[i] = 0.0;
        if (i < p) {
            double sum = w[i];
#pragma unroll
            for (int q = i + 1; q < MAXD - 1; ++q)
                if (q < p) sum = fma(-lo[q * (q - 1) / 2 + i], z[q], sum);
            z[i] = sum / ld[i];
            uz = fma(tabv(fu + i), z[i], uz);
            zz = fma(z[i], z[i], zz);
        }
    }
    const double sk = acc[MAXD - 1] + 1e-6 - ww;
    if (!(sk > 1e-18) || !isfinite(lnd)) return 1e8;
    const double bk = (acc[MAXD] - wt) / sk;
    const double nb2 = uu - 2.0 * bk * uz + bk * bk * (zz + 1.0);
    const double q = q0 - bk * bk * sk - 1e-6 * nb2;
    const double nf = (double)n, pf = (double)dim;
    const double total = (nf - pf) * jx_log(q) + logdetv + (lnd + jx_log(sk));
    const double cst = (nf - pf) * (jx_log(nf - pf) - 1.0 - jx_log(2.0 * M_PI)) / 2.0;
    const double reml = cst - 0.5 * total;
    return isfinite(reml) ? -reml : 1e8;
}
// memory operations of one wave on its own LDS region: writes by some lanes, reads by others, no other wave involved
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
constexpr int IP_CP = CH_N + 1;     // pitch of the cosine table rows (doubles): column reads of 32 lanes on distinct banks

// ---- the chain scan in three launches (round 6) -----------------------------------------------------------------------------
// Along a warm-start chain only Brent's START depends on the row before; the interpolant of a row's objective (node evaluations
// + cosine transform) and the final evaluation at its optimum (beta / SE / p) do not.  So a chain scan on stored series is
//   phase 1  every row in parallel: the 64 Chebyshev coefficients of its objective -> `icoef`, `rflag` (lmm_scan_fast_kernel, one
//            wave per row, split.phase = 1),
//   chains   chain_interp_brent_kernel: one wave per chain, per row ONLY Brent on the stored interpolant (an evaluation = one
//            fully unrolled Clenshaw recurrence from LDS) -> `xopt`, evaluations, carry,
//   phase 2  every row in parallel: final evaluation at xopt, outputs (lmm_scan_fast_kernel, split.phase = 2).
// A chain that holds a row whose interpolant could not be formed (a node failed: that row needs direct evaluations) is flagged
// in `cflag` and walked by the one-kernel form (split.phase = 0 with split.cflag: only flagged chains).  Same arithmetic in the
// same order as the one-kernel form: the outputs are bit-identical (JXGPU_SCAN_CHAIN_SPLIT=0 runs the one-kernel form).
struct ChainSplit {
    double *icoef = nullptr;          // (rows, SR_M)
    double *xopt = nullptr;           // (rows): NaN = not a row of the split form (invalid row, or a flagged chain's)
    int32_t *rflag = nullptr;         // (rows): 0 invalid row (its outputs are written by phase 1), 1 interpolant stored, 2 direct
    const int32_t *cflag = nullptr;   // (chains): with phase 0 and chains, walk only the chains whose flag is set
    int phase = 0;
    int force_row = -1;               // test hook (JXGPU_SCAN_CHAIN_FORCE_DIRECT): this row is flagged 2 whatever its nodes say
};

// Brent's minimiser (src/math/brent.rs:1-136, verbatim control flow) of `objective` on [low, high], started from `last` when
// `have_last` (seed_with_init_guess / carry_warm_start) -> optimum, number of evaluations
template <class F>
__device__ __forceinline__ void brent_minimize(F &&objective, double low, double high, double tol_in, int max_iter, bool have_last,
                                               double last, double &x_out, int &evals_out) {
    double a = low, c = high;
    if (!(a < c)) {
        const double tt = a;
        a = c;
        c = tt;
    }
    const double eps = 2.220446049250313e-16;
    const double tol = fmax(fabs(tol_in), 1e-12);
    double x = (have_last && isfinite(last) && last >= a && last <= c) ? last : 0.5 * (a + c);
    double w = x, v = x;
    double fx = objective(x), fw = fx, fv = fx;
    double d = 0.0, e = 0.0;
    int evals = 1;
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
        const double fu = objective(u);
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
    x_out = x;
    evals_out = evals;
}

// clenshaw() with every coefficient read in flight before the recurrence starts (same operations in the same order)
__device__ __forceinline__ double clenshaw_unrolled(const double *__restrict__ c, double t) {
    double cv[CH_N];
#pragma unroll
    for (int j = 0; j < CH_N; ++j) cv[j] = c[j];
    double b1 = 0.0, b2 = 0.0;
    const double t2 = 2.0 * t;
#pragma unroll
    for (int j = CH_N - 1; j >= 1; --j) {
        const double b0 = fma(t2, b1, cv[j] - b2);
        b2 = b1;
        b1 = b0;
    }
    return fma(t, b1, 0.5 * cv[0] - b2);
}

__global__ __launch_bounds__(64) void chain_interp_brent_kernel(const double *__restrict__ icoef, const int32_t *__restrict__ rflag,
                                                                const int32_t *__restrict__ chain_off, int nchains,
                                                                double *__restrict__ carry, int32_t *__restrict__ cflag,
                                                                double *__restrict__ xopt, int32_t *__restrict__ evals_out,
                                                                const ChebHeader shd, double low, double high, double tol_in,
                                                                int max_iter) {
    __shared__ __attribute__((aligned(16))) double l_c[2][SR_M];
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    if (unit >= nchains) return;
    const int r_beg = chain_off[unit], r_end = chain_off[unit + 1];
    {
        int bad = 0;
        for (int r = r_beg + lane; r < r_end; r += 64) bad |= (rflag[r] == 2) ? 1 : 0;
        const bool any_bad = __builtin_amdgcn_ballot_w64(bad != 0) != 0ull;
        if (lane == 0) cflag[unit] = any_bad ? 1 : 0;
        if (any_bad) return;                            // the one-kernel form walks this chain
    }
    double last = carry[unit];
    bool have_last = isfinite(last);
    double pre = 0.0;
    int pflag = 0;
    if (r_beg < r_end) {
        pre = icoef[(int64_t)r_beg * SR_M + lane];
        pflag = rflag[r_beg];
    }
    const double inv_w = 1.0 / shd.segw;
    int flip = 0;
    for (int r = r_beg; r < r_end; ++r) {
        double *lc = l_c[flip];
        flip ^= 1;
        wave_lds_sync();
        lc[lane] = pre;
        wave_lds_sync();
        const int flag = pflag;
        if (r + 1 < r_end) {
            pre = icoef[(int64_t)(r + 1) * SR_M + lane];
            pflag = rflag[r + 1];
        }
        if (flag != 1) continue;                        // invalid row: phase 1 wrote its outputs, the chain's state stays
        auto objective = [&](double xx) -> double {
            int seg = (int)((xx - shd.low) * inv_w);
            if (seg < 0) seg = 0;
            if (seg >= shd.nseg) seg = shd.nseg - 1;
            const double t = (xx - (shd.low + shd.segw * ((double)seg + 0.5))) * (2.0 * inv_w);
            return clenshaw_unrolled(lc + seg * CH_N, t);
        };
        double x;
        int evals;
        brent_minimize(objective, low, high, tol_in, max_iter, have_last, last, x, evals);
        last = x;
        have_last = true;
        if (lane == 0) {
            xopt[r] = x;
            if (evals_out) evals_out[r] = evals;
        }
    }
    if (lane == 0 && have_last) carry[unit] = last;
}

// NW waves per workgroup.  LDS = true: the three vectors every evaluation of every SNP streams -- s, X~ (n x p) and the
// shifted y~, 8 n (2 + p) bytes -- are copied into LDS once per workgroup and shared by its NW waves.  Without it each
// evaluation re-reads them through L2 (they do not fit the 32 KB L1): measured 7.7 TB/s of L1<-L2 traffic and waves
// parked on s_waitcnt for 87 % of their cycles at n = 5000 (profiles/r01d_pmc_scan.json).
template <int MAXD>
__device__ __forceinline__ void series_eval_sums(double x, const ChebHeader hd, const double *__restrict__ sc, int p,
                                                 double (&acc)[MAXD + 1]);

// SERIES: the SNP-specific sums come from the SNP's own Chebyshev series (series_coef_kernel: `snp_coef`, `snp_ssq`) instead of
// a pass over the rotated row -- `grot` is not read.
// SPLIT (with INTERP): the phases of the three-launch chain scan (ChainSplit above) are compiled in
template <int MAXD, int NW, bool LDS, bool SERIES = false, bool INTERP = false, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64, (NW >= 16 ? 4 : (MAXD <= 2 ? (SPLIT ? 2 : 4) : 2))) void lmm_scan_fast_kernel(
    const float *__restrict__ grot, int nrows, int n, const double *__restrict__ s_g, const double *__restrict__ xcov_g,
    const double *__restrict__ yc_g, int p, const ChebHeader hd, const double *__restrict__ coef,
    const double *__restrict__ smin_ptr, double low, double high, double tol_in, int max_iter, int warm, double init,
    int with_plrt, double nullml, double *__restrict__ out, int32_t *__restrict__ evals_out,
    const double *__restrict__ snp_coef, const double *__restrict__ snp_ssq, const ChebHeader shd,
    const int32_t *__restrict__ chain_off = nullptr, int nchains = 0, double *__restrict__ carry = nullptr,
    const ChainSplit split = ChainSplit()) {
    extern __shared__ __attribute__((aligned(16))) double scan_lds[];
    if (MAXD == 2) p = 1;     // dim = p + 1 <= 2 and p >= 1: a compile-time p (see lmm_scan_tiled_kernel)
    // one-kernel form behind the split chain scan: only the flagged chains (one workgroup per chain there)
    if (INTERP && SPLIT && NW == 1 && chain_off && split.cflag && (int)gridDim.x >= nchains && split.cflag[blockIdx.x] == 0) return;
    const double *s = s_g, *xcov = xcov_g, *yc = yc_g;
    if (LDS) {
        double *ls = scan_lds, *lx = scan_lds + n, *ly = scan_lds + n + (int64_t)n * p;
        for (int i = threadIdx.x; i < n; i += NW * 64) {
            ls[i] = s_g[i];
            ly[i] = yc_g[i];
        }
        for (int i = threadIdx.x; i < n * p; i += NW * 64) lx[i] = xcov_g[i];
        __syncthreads();
        s = ls;
        xcov = lx;
        yc = ly;
    }
    const int out_cols = with_plrt ? 4 : 3;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const double smin = smin_ptr[0];
    // CHAINLDS (the one-wave-per-chain series instantiation): a chain is a SEQUENTIAL job, so what counts is the latency of one
    // evaluation, not the throughput of many -- with the coefficients read from global memory an evaluation took 4.5 us (two
    // dependent 32-term recurrences, each waiting for its loads four times; measured round 6: 742 ms for 20 chains of 10 000
    // SNPs at BASELINE configs[2]).  Here the lambda-only tables are staged in LDS once per workgroup and the series of the
    // NEXT SNP is fetched into registers while the current one is searched (two LDS buffers), so an evaluation reads LDS only.
    // INTERP (with SERIES, plain form): Brent on an interpolant of the objective, see plain_objective above.
    constexpr bool CHAINLDS = SERIES && (NW == 1 || INTERP);
    constexpr int NPRE = MAXD + 1;                     // (p + 2) <= MAXD + 1 series of SR_M = 64 entries: one entry per lane each
    [[maybe_unused]] double *l_coef = scan_lds, *l_sc = scan_lds, *l_cos = scan_lds, *l_fv = scan_lds;
    [[maybe_unused]] double pre[NPRE];
    if (CHAINLDS) {
        // LDS: [cosine table CH_N x IP_CP (INTERP)] [lambda-only coefficient tables] then per wave [two series buffers] [64 node
        // values / 64 coefficients (INTERP)]
        const int ncoef = hd.nseg * hd.nf * CH_N;
        double *base = scan_lds;
        if (INTERP) {
            l_cos = base;
            for (int e = threadIdx.x; e < CH_N * CH_N; e += NW * 64) {
                const int j = e / CH_N, k = e - j * CH_N;
                l_cos[j * IP_CP + k] = cos(M_PI * (double)j * ((double)k + 0.5) / (double)CH_N);
            }
            base += CH_N * IP_CP + 1;
        }
        l_coef = base;
        for (int i = threadIdx.x; i < ncoef; i += NW * 64) l_coef[i] = coef[i];
        base += (ncoef + 1) & ~1;
        l_sc = base + (size_t)wave * (2 * NPRE * SR_M + (INTERP ? 2 * SR_M : 0));
        l_fv = l_sc + 2 * NPRE * SR_M;
        __syncthreads();
        coef = l_coef;
    }
    // chain_off: the reference's warm-start chain (carry_warm_start, src/stats/lmm.rs:134-161): this wave walks the rows
    // [chain_off[c], chain_off[c + 1]) in order and every SNP's Brent starts from the optimum of the valid row before it; carry[c]
    // is the state the chain starts from (NaN: none) and receives the state it ends with (a chain may continue in a later launch).
    // Without chain_off a unit is one row and `warm` / `init` seed every row alike (seed_with_init_guess).
    const int nunits = chain_off ? nchains : nrows;
    for (int unit = blockIdx.x * NW + wave; unit < nunits; unit += gridDim.x * NW) {
      if (INTERP && SPLIT && chain_off && split.cflag && split.cflag[unit] == 0) continue;
      const int r_beg = chain_off ? chain_off[unit] : unit, r_end = chain_off ? chain_off[unit + 1] : unit + 1;
      double last = chain_off ? carry[unit] : init;
      bool have_last = chain_off ? isfinite(last) : (warm != 0);
      if (CHAINLDS && r_beg < r_end) {
#pragma unroll
          for (int k = 0; k < NPRE; ++k)
              if (k < p + 2) pre[k] = snp_coef[(int64_t)r_beg * (p + 2) * SR_M + k * SR_M + lane];
      }
      [[maybe_unused]] int lds_flip = 0;
      for (int r = r_beg; r < r_end; ++r) {
        const float *g = SERIES ? nullptr : grot + (int64_t)r * n;
        double *o = out + (int64_t)r * out_cols;
        [[maybe_unused]] double *sc_lds = nullptr;
        if (CHAINLDS) {
            // this row's series: registers -> LDS buffer (r & 1); then the next row's series on its way into the registers
            sc_lds = l_sc + (size_t)(lds_flip & 1) * NPRE * SR_M;
            lds_flip ^= 1;
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < NPRE; ++k)
                if (k < p + 2) sc_lds[k * SR_M + lane] = pre[k];
            wave_lds_sync();
            if (r + 1 < r_end) {
#pragma unroll
                for (int k = 0; k < NPRE; ++k)
                    if (k < p + 2) pre[k] = snp_coef[(int64_t)(r + 1) * (p + 2) * SR_M + k * SR_M + lane];
            }
        }
        double ssq = 0.0;
        if (SERIES) {
            ssq = snp_ssq[r];
        } else {
            for (int i = lane; i < n; i += 64) {
                const double v = (double)g[i];
                ssq += v * v;
            }
            ssq = wave_allsum(ssq);
        }
        // one evaluation of the Brent objective (and of final_beta_se) for this wave's SNP
        const double *sc = CHAINLDS ? sc_lds : (SERIES ? snp_coef + (int64_t)r * (p + 2) * SR_M : nullptr);
        auto eval_at = [&](double xx, bool want_ainv, FastEval<MAXD> &res, const double *bmid) {
            if (!SERIES) {
                fast_eval<MAXD>(xx, hd, coef, smin, s, xcov, yc, g, n, p, want_ainv, res, bmid);
                return;
            }
            res.ok = false;
            res.reml_neg = 1e8;
            res.q = 0.0;
            res.logdetv = 0.0;
            res.beta_k = 0.0;
            res.ainv_kk = 0.0;
            if (fast_eval_lambda(xx, smin, n, p + 1) < 0.0) return;
            double acc[MAXD + 1];
            series_eval_sums<MAXD>(xx, shd, sc, p, acc);
            fast_eval_finish<MAXD, true>(xx, hd, coef, n, p, want_ainv, bmid, acc, res);
        };
        if (!isfinite(ssq) || ssq <= 1e-12) {
            if (lane == 0) {
                o[0] = nan("");
                o[1] = nan("");
                o[2] = 1.0;
                if (with_plrt) o[3] = 1.0;
                if (evals_out) evals_out[r] = 0;
                if constexpr (INTERP && SPLIT)
                    if (split.phase == 1) split.rflag[r] = 0;
            }
            continue;
        }
        FastEval<MAXD> ev;
        [[maybe_unused]] bool use_interp = false;
        if constexpr (INTERP) if (!SPLIT || split.phase != 2) {
            // the objective at the 64 Chebyshev nodes (lane = node), then its Chebyshev coefficients per segment
            const int sg = min(lane >> 5, shd.nseg - 1);
            const int kk = lane & 31;
            const double tk = l_cos[IP_CP + kk];                         // T_1 at node kk: the node itself
            const double xn = shd.low + shd.segw * ((double)sg + 0.5) + tk * (0.5 * shd.segw);
            double fvn = 1e8;
            if (fast_eval_lambda(xn, smin, n, p + 1) >= 0.0) {
                double accn[MAXD + 1];
#pragma unroll
                for (int k = 0; k < MAXD + 1; ++k) accn[k] = 0.0;
#pragma unroll
                for (int rr = 0; rr < MAXD - 1; ++rr)
                    if (rr < p) accn[rr] = clenshaw(sc + rr * SR_M + sg * CH_N, tk);
                accn[MAXD - 1] = clenshaw(sc + p * SR_M + sg * CH_N, tk);
                accn[MAXD] = clenshaw(sc + (p + 1) * SR_M + sg * CH_N, tk);
                if constexpr (MAXD >= 8) {
                    // block form: the tables have their own (narrower) segments
                    int tsg = (int)((xn - hd.low) / hd.segw);
                    if (tsg < 0) tsg = 0;
                    if (tsg >= hd.nseg) tsg = hd.nseg - 1;
                    const double tt = (xn - (hd.low + hd.segw * ((double)tsg + 0.5))) / (0.5 * hd.segw);
                    fvn = blk_objective<MAXD>(n, p, coef + (int64_t)tsg * hd.nf * CH_N, tt, accn);
                } else {
                    double tv[InterpTv<MAXD>::N];
#pragma unroll
                    for (int f = 0; f < InterpTv<MAXD>::N; ++f)
                        tv[f] = (f < hd.nf) ? clenshaw(coef + ((int64_t)sg * hd.nf + f) * CH_N, tk) : 0.0;
                    fvn = plain_objective<MAXD>(n, p, tv, accn);
                }
            }
            use_interp = __builtin_amdgcn_ballot_w64(!(fvn < 1e8)) == 0ull;      // wave-uniform: every node evaluated
            if (use_interp) {
                l_fv[lane] = fvn;
                wave_lds_sync();
                const int jj = lane & 31, s2 = lane >> 5;
                double cj = 0.0;
#pragma unroll 8
                for (int k = 0; k < CH_N; ++k) cj = fma(l_fv[s2 * CH_N + k], l_cos[jj * IP_CP + k], cj);
                l_fv[SR_M + lane] = cj * (2.0 / (double)CH_N);
                wave_lds_sync();
            }
            if (SPLIT && split.phase == 1) {
                if (use_interp) split.icoef[(int64_t)r * SR_M + lane] = l_fv[SR_M + lane];
                if (lane == 0) split.rflag[r] = (use_interp && r != split.force_row) ? 1 : 2;
                continue;
            }
        }
        auto objective = [&](double xx) -> double {
            if constexpr (INTERP) {
                if (use_interp) {
                    // segment and argument by multiplication with the (exact power-of-two-free) reciprocals: an IEEE division is
                    // ~40 dependent instructions, and this is the whole evaluation now
                    const double inv_w = 1.0 / shd.segw;
                    int seg = (int)((xx - shd.low) * inv_w);
                    if (seg < 0) seg = 0;
                    if (seg >= shd.nseg) seg = shd.nseg - 1;
                    const double t = (xx - (shd.low + shd.segw * ((double)seg + 0.5))) * (2.0 * inv_w);
                    return clenshaw(l_fv + SR_M + seg * CH_N, t);
                }
            }
            eval_at(xx, false, ev, nullptr);
            return ev.reml_neg;
        };
        // ---- Brent (src/math/brent.rs:1-136, verbatim control flow: brent_minimize) ----------------------
        double x = 0.0;
        int evals = -1;
        bool searched = true;
        if constexpr (INTERP && SPLIT) {
            if (split.phase == 2) {
                x = split.xopt[r];
                if (!(x == x)) continue;                  // not a row of the split form (the one-kernel form wrote it)
                searched = false;
            }
        }
        if (searched) brent_minimize(objective, low, high, tol_in, max_iter, have_last, last, x, evals);
        if (chain_off) {
            last = x;
            have_last = true;
        }
        // ---- final_beta_se (src/stats/reml.rs:472-568) at the optimum ------------------------------------
        eval_at(x, true, ev, smin_ptr + 8);
        double beta = nan(""), se = nan("");
        const int dim = p + 1;
        if (ev.ok) {
            const double sigma2 = ev.q / ((double)n - (double)dim);
            const double var = sigma2 * ev.ainv_kk;
            if (var > 0.0 && isfinite(var)) {
                beta = ev.beta_k;
                se = sqrt(var);
            }
        }
        if (lane == 0) {
            if (evals_out && evals >= 0) evals_out[r] = evals;
            if (isfinite(beta) && isfinite(se) && se > 0.0) {
                const double z = beta / se;
                double pv = 2.0 * (0.5 * jx_erfc(fabs(z) / 1.4142135623730951));
                if (pv < 2.2250738585072014e-308) pv = 2.2250738585072014e-308;
                if (pv > 1.0) pv = 1.0;
                o[0] = beta;
                o[1] = se;
                o[2] = isfinite(pv) ? pv : 1.0;
                if (with_plrt) {
                    double plrt = 1.0;
                    if (ev.ok && isfinite(ev.q) && ev.q > 0.0) {
                        const double nf = (double)n;
                        const double ml =
                            nf * (jx_log(nf) - 1.0 - jx_log(2.0 * M_PI)) / 2.0 - 0.5 * (nf * jx_log(ev.q) + ev.logdetv);
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


// ---- per-SNP Chebyshev series of the SNP-specific sums: ONE pass over a rotated row -----------------------------------
// The sums an evaluation needs from the rotated row g~ -- c_r(x) = sum g x~_r / v, d(x) = sum g^2 / v, b(x) = sum g y_c / v with
// v_i = s_i + 10^x -- are as analytic in x = log10 lambda as the lambda-only sums tabulated above (same poles), so they have the
// same 32-term Chebyshev series per width-2 segment, and the series are LINEAR in the row: with
//   What[i][seg 32 + j] = (2 / 32) sum_k cos(pi j (k + 1/2) / 32) / (s_i + lambda_{seg,k})        (lambda-only, built once per model)
// the coefficients of a SNP are  coef_q[m] = sum_i (g_i q_i) What[i][m],  q_i in {x~_ir, g_i, y_ci}:  a dense f64 product over the
// samples -- (p + 2) rows per SNP against the n x 64 table -- on v_mfma_f64_16x16x4_f64, one streaming pass over g~ (4 n bytes per
// SNP: the algorithmic traffic; the tiled kernel re-reads the row once per Brent evaluation, 16 x).  Brent then runs on the series
// alone (lmm_scan_fast_kernel<..., SERIES>): an evaluation is one Clenshaw recurrence per lane and the unchanged evaluation
// tail -- no sample loop.  src/stats/lmm.rs:94-199 (per-SNP Brent over reml_loglike), src/stats/reml.rs:255-362.

// What[i][m] for i < npad (zero rows beyond n), m = seg 32 + j (zero beyond nseg 32)
__global__ __launch_bounds__(64) void series_what_kernel(const double *__restrict__ s, int n, int npad, ChebHeader hd,
                                                         double *__restrict__ what) {
    __shared__ double cs[CH_N][CH_N + 1];
    for (int e = threadIdx.x; e < CH_N * CH_N; e += 64) {
        const int j = e / CH_N, k = e % CH_N;
        cs[j][k] = cos(M_PI * (double)j * ((double)k + 0.5) / (double)CH_N);
    }
    __syncthreads();
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= npad) return;
    double *o = what + (int64_t)i * SR_M;
    for (int seg = 0; seg < SR_M / CH_N; ++seg) {
        double r[CH_N];
        const bool live = i < n && seg < hd.nseg;
#pragma unroll
        for (int k = 0; k < CH_N; ++k) {
            const double t = cos(M_PI * ((double)k + 0.5) / (double)CH_N);
            const double x = hd.low + hd.segw * ((double)seg + 0.5) + 0.5 * hd.segw * t;
            r[k] = live ? 1.0 / (s[i] + pow(10.0, x)) : 0.0;
        }
        for (int j = 0; j < CH_N; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < CH_N; ++k) acc += r[k] * cs[j][k];
            o[seg * CH_N + j] = acc * (2.0 / (double)CH_N);
        }
    }
}

// coef[(r nq + q) 64 + m] = sum_i g[r][i] mult_q[i] What[i][m] for the quantities q0 <= q < q0 + NQ of the nq = p + 2;
// ssq[r] = sum_i g[r][i]^2 (written by the launch with q0 == 0).  mult: q < p: x~ column q; q == p: the row itself; q == p + 1: y_c.
// 512 threads = 8 waves x 16 SNPs; more than four quantities per SNP (two or more covariates beside the intercept) take several
// launches -- a pass over the rows each -- because 16 SNPs x NQ quantities x 64 entries of f64 accumulators are 32 NQ registers.
template <int NQ>
__global__ __launch_bounds__(512, 2) void series_coef_kernel(const float *__restrict__ grot, int nrows, int n, int npad, int p,
                                                             int q0, const double *__restrict__ xcov,
                                                             const double *__restrict__ yc, const double *__restrict__ what,
                                                             double *__restrict__ coef, double *__restrict__ ssq) {
    typedef double d4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) double sr_smem[];
    double *wt = sr_smem;                                         // [SR_SC][SR_WP]
    double *mq = wt + SR_SC * SR_WP;                              // [p + 1][SR_SC]: x~ columns, then y_c
    float *gt = reinterpret_cast<float *>(mq + (p + 1) * SR_SC);  // [128][SR_GP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * 128;
    const int nq = p + 2;
    // staging map: table chunk = 64 samples x 64 entries = 32 KB contiguous (4 x 16 B per thread); row chunk: thread = (row
    // tid >> 2, 16 floats); multipliers: (p + 1) x 64 doubles, up to two per thread
    const int grow = tid >> 2, gpart = tid & 3;
    const bool grow_ok = r0 + grow < nrows;
    const bool v4 = (n & 3) == 0;
    double2 wr[4];
    float4 gr[4];
    double mr[2] = {0.0, 0.0};
    auto load_chunk = [&](int i0) {
        const double2 *wsrc = reinterpret_cast<const double2 *>(what + (int64_t)i0 * SR_M);
#pragma unroll
        for (int u = 0; u < 4; ++u) wr[u] = wsrc[tid + 512 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // branch-free: clamped (always valid) addresses, out-of-range samples and rows zeroed by selects
            const int i = i0 + 16 * gpart + 4 * u;
            float4 v;
            if (v4) {
                v = *reinterpret_cast<const float4 *>(grot + (int64_t)(grow_ok ? r0 + grow : 0) * n + min(i, n - 4));
                if (i >= n || !grow_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const float *rowp = grot + (int64_t)(grow_ok ? r0 + grow : 0) * n;
                v.x = rowp[min(i, n - 1)];
                v.y = rowp[min(i + 1, n - 1)];
                v.z = rowp[min(i + 2, n - 1)];
                v.w = rowp[min(i + 3, n - 1)];
                if (i >= n || !grow_ok) v.x = 0.f;
                if (i + 1 >= n || !grow_ok) v.y = 0.f;
                if (i + 2 >= n || !grow_ok) v.z = 0.f;
                if (i + 3 >= n || !grow_ok) v.w = 0.f;
            }
            gr[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 512 * u;
            if (e < (p + 1) * SR_SC) {
                const int q = e / SR_SC, i = i0 + e % SR_SC;
                mr[u] = (i < n) ? (q < p ? xcov[(int64_t)i * p + q] : yc[i]) : 0.0;
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = 2 * (tid + 512 * u);                    // double index inside the 64 x 64 chunk
            const int smp = e / SR_M, m = e % SR_M;
            *reinterpret_cast<double2 *>(wt + smp * SR_WP + m) = wr[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<float4 *>(gt + grow * SR_GP + 16 * gpart + 4 * u) = gr[u];
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tid + 512 * u < (p + 1) * SR_SC) mq[tid + 512 * u] = mr[u];
    };
    d4v acc[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[q][b] = d4v{0.0, 0.0, 0.0, 0.0};
    double sq = 0.0;
    const int fi = lane & 15, fk = lane >> 4;                     // fragment row / column index and k index of this lane
    const float *grow_l = gt + (wave * 16 + fi) * SR_GP + fk;
    // multiplier row of quantity q0 + q (the row itself for quantity p: flagged by -1)
    int mrow[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int gq = q0 + q;
        mrow[q] = (gq == p) ? -1 : (gq < p ? gq : p);
    }
    load_chunk(0);
    for (int i0 = 0; i0 < npad; i0 += SR_SC) {
        __syncthreads();                                          // the previous chunk has been consumed
        store_chunk();
        __syncthreads();
        if (i0 + SR_SC < npad) load_chunk(i0 + SR_SC);           // in flight while this chunk is multiplied
#pragma unroll 4
        for (int ks = 0; ks < SR_SC / 4; ++ks) {
            const double gv = (double)grow_l[4 * ks];
            double a[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) a[b] = wt[(4 * ks + fk) * SR_WP + 16 * b + fi];
            sq = fma(gv, gv, sq);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const double mult = (mrow[q] < 0) ? gv : mq[mrow[q] * SR_SC + 4 * ks + fk];
                const double bv = gv * mult;
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[q][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[b], bv, acc[q][b], 0, 0, 0);
            }
        }
    }
    // D[entry = 16 b + (lane >> 4) + 4 r][SNP = lane & 15]
    const int snp = r0 + wave * 16 + fi;
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (snp < nrows) {
        if (fk == 0 && q0 == 0) ssq[snp] = sq;
        double *o = coef + ((int64_t)snp * nq + q0) * SR_M;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (q0 + q < nq) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[q * SR_M + 16 * b + fk + 4 * r] = acc[q][b][r];
            }
    }
}

// SNP-specific sums of an evaluation from the SNP's own series (lane q evaluates quantity q; the wave sum of fast_eval_finish
// then hands every lane the value): acc[r < p] = c_r, acc[MAXD - 1] = d, acc[MAXD] = b
template <int MAXD>
__device__ __forceinline__ void series_eval_sums(double x, const ChebHeader hd, const double *__restrict__ sc, int p,
                                                 double (&acc)[MAXD + 1]) {
    // hd: the SERIES' own segmentation (series_header)
    const int lane = threadIdx.x & 63;
    int seg = (int)((x - hd.low) / hd.segw);
    if (seg < 0) seg = 0;
    if (seg >= hd.nseg) seg = hd.nseg - 1;
    const double t = (x - (hd.low + hd.segw * ((double)seg + 0.5))) / (0.5 * hd.segw);
    const int q = lane < p + 2 ? lane : 0;
    const double val = clenshaw(sc + q * SR_M + seg * CH_N, t);
    // every lane gets all p + 2 values (lane q holds quantity q): readlane broadcasts, no butterfly sum behind them
#pragma unroll
    for (int k = 0; k < MAXD + 1; ++k) acc[k] = 0.0;
#pragma unroll
    for (int r = 0; r < MAXD - 1; ++r)
        if (r < p) acc[r] = wave_bcast(val, r);
    acc[MAXD - 1] = wave_bcast(val, p);
    acc[MAXD] = wave_bcast(val, p + 1);
}

// ---- tiled form for n beyond the LDS-resident limit --------------------------------------------------------------------
// The vectors every evaluation streams (s, X~, shifted y~: 8 n (2 + p) bytes, 480 KB at n = 20 000) do not fit LDS, and the
// one-wave-per-SNP form above re-reads them through L2 for every evaluation of every SNP (measured 13.5x the algorithmic
// bytes, association 291 ms at BASELINE configs[2]).  Here a workgroup of NW waves walks the samples in tiles held in LDS:
// all waves evaluate in lock step over the tiles (one evaluation of one SNP per wave per round, whatever Brent iteration
// the wave is in), a wave that finishes a SNP takes the next one from the workgroup's queue in the following round.  Brent
// is the same state machine unrolled in time: identical evaluation points and sums (per-lane order of the samples is
// unchanged: tile lengths are multiples of 64), identical results.
struct BrentState {
    double a, c, x, w, v, fx, fw, fv, d, e;
    int it, evals;
};

// the head of one iteration of src/math/brent.rs:1-136: false = converged / out of iterations, true = evaluate at u next
__device__ __forceinline__ bool brent_propose(BrentState &b, double tol, int max_iter, double &u) {
    const double eps = 2.220446049250313e-16;
    if (b.it >= max_iter) return false;
    const double m = 0.5 * (b.a + b.c);
    const double tol1 = tol * fabs(b.x) + eps;
    const double tol2 = 2.0 * tol1;
    if (fabs(b.x - m) <= tol2 - 0.5 * (b.c - b.a)) return false;
    bool use_par = false;
    u = b.x;
    if (fabs(b.e) > tol1) {
        double pq = (b.x - b.v) * ((b.x - b.w) * (b.fx - b.fv)) - (b.x - b.w) * ((b.x - b.v) * (b.fx - b.fw));
        double q = 2.0 * (((b.x - b.v) * (b.fx - b.fw)) - ((b.x - b.w) * (b.fx - b.fv)));
        if (q > 0.0)
            pq = -pq;
        else
            q = -q;
        bool ok = false;
        if (fabs(q) > eps) {
            const double sstep = pq / q;
            u = b.x + sstep;
            if ((u - b.a) >= tol2 && (b.c - u) >= tol2 && fabs(sstep) < 0.5 * fabs(b.e)) ok = true;
        }
        if (ok) {
            b.d = pq / q;
            u = b.x + b.d;
            if ((u - b.a) < tol2 || (b.c - u) < tol2) b.d = (b.x < m) ? tol1 : -tol1;
            use_par = true;
        }
    }
    if (!use_par) {
        b.e = (b.x < m) ? (b.c - b.x) : (b.a - b.x);
        b.d = 0.3819660 * b.e;
    }
    if (fabs(b.d) < tol1) b.d = (b.d >= 0.0) ? tol1 : -tol1;
    u = b.x + b.d;
    return true;
}

// the tail of the same iteration, after f(u) is known
__device__ __forceinline__ void brent_update(BrentState &b, double u, double fu) {
    ++b.evals;
    if (fu <= b.fx) {
        if (u >= b.x)
            b.a = b.x;
        else
            b.c = b.x;
        b.v = b.w;
        b.fv = b.fw;
        b.w = b.x;
        b.fw = b.fx;
        b.x = u;
        b.fx = fu;
    } else {
        if (u >= b.x)
            b.c = u;
        else
            b.a = u;
        if (fu <= b.fw || b.w == b.x) {
            b.v = b.w;
            b.fv = b.fw;
            b.w = u;
            b.fw = fu;
        } else if (fu <= b.fv || b.v == b.x || b.v == b.w) {
            b.v = u;
            b.fv = fu;
        }
    }
    ++b.it;
}

// LMM2 = true (src/stats/lmm.rs:202-330): behind the REML search and its Wald statistics a second Brent search on -ML, seeded
// with the REML optimum, and the likelihood-ratio test against `nullml`; out (nrows, 6) = [beta, se, pwald, lambda_reml, ml_alt,
// plrt].  The evaluations are the same passes (an evaluation yields q and sum ln v, i.e. both objectives); the Wald results
// wait in a per-wave LDS slot while the second search runs (phases 4 / 5).
template <int MAXD, int NW, bool LMM2 = false>
__global__ __launch_bounds__(NW * 64) void lmm_scan_tiled_kernel(
    const float *__restrict__ grot, int nrows, int n, const double *__restrict__ s_g, const double *__restrict__ xcov_g,
    const double *__restrict__ yc_g, int p, const ChebHeader hd, const double *__restrict__ coef,
    const double *__restrict__ smin_ptr, double low, double high, double tol_in, int max_iter, int warm, double init,
    int with_plrt, double nullml, double *__restrict__ out, int32_t *__restrict__ evals_out, int tile, int rows_per_wg) {
    // MAXD = 2 is dispatched for dim = p + 1 <= 2 only, i.e. p = 1 (p >= 1 is checked by the host wrapper): a compile-time p
    // turns the X~ addresses of the batched sample loop into immediate offsets (with a run-time p the compiler kept one
    // LDS address register per sample slot of the unrolled loop alive for the whole kernel)
    if (MAXD == 2) p = 1;
    extern __shared__ __attribute__((aligned(16))) double scan_lds[];
    double *ls = scan_lds, *lx = scan_lds + tile, *ly = scan_lds + tile + (int64_t)tile * p;
    int &next_row = *reinterpret_cast<int *>(scan_lds + (int64_t)tile * (2 + p));   // queue head, behind the tile (all LDS is
                                                                                     // dynamic: the base stays 16-byte aligned)
    // The Brent state of the wave's SNP (wave-uniform: 10 doubles + 2 counters) lives in LDS between two rounds: held in
    // registers it was live across the tile loop in every lane and pushed the 1024-thread workgroup (128 VGPRs per lane)
    // into scratch (58 spilled registers, 1.4 GB of scratch writes per launch at BASELINE configs[2]).  Lane 0 stores it,
    // every lane reads it back (one broadcast read per value) where the round's evaluation is folded in.
    BrentState *const bslot = reinterpret_cast<BrentState *>(scan_lds + (int64_t)tile * (2 + p) + 2) + (threadIdx.x >> 6);
    double *const kslot = reinterpret_cast<double *>(reinterpret_cast<BrentState *>(scan_lds + (int64_t)tile * (2 + p) + 2) + 16) +
                          4 * (threadIdx.x >> 6);       // LMM2: beta, se, pwald, lambda of the REML stage
    const int out_cols = LMM2 ? 6 : (with_plrt ? 4 : 3);
    const int lane = threadIdx.x & 63;
    const double smin = smin_ptr[0];
    const int dim = p + 1;
    const int row_begin = blockIdx.x * rows_per_wg;
    const int row_end = min(nrows, row_begin + rows_per_wg);
    if (threadIdx.x == 0) next_row = row_begin;
    __syncthreads();
    const bool vec4 = (n % 4 == 0) && (tile % 256 == 0) && ((reinterpret_cast<uintptr_t>(grot) & 15) == 0);
    double lo_b = low, hi_b = high;
    if (!(lo_b < hi_b)) {
        const double tt = lo_b;
        lo_b = hi_b;
        hi_b = tt;
    }
    // wave state: phase 0 = needs a SNP, 1 = first evaluation (also forms sum g^2), 2 = Brent evaluation at u, 3 = final
    int phase = 0, r = -1;
    bool exhausted = false;
    double x_eval = 0.0;                                  // the point the round evaluates (= u of the Brent iteration in phase 2)
    for (;;) {
        // Re-derive the round-invariant quantities (sample count as a double, staging addresses, the tolerance) in every round:
        // hoisted out of the round loop they are live across the tile loop and, at 128 VGPRs per lane, end up in scratch.
        // The empty asm makes the scalar inputs opaque per round, so the few instructions stay where they are used.
        int n_r = n;
        double tol_r = tol_in;
        const double *s_r = s_g, *y_r = yc_g, *x_r = xcov_g, *coef_r = coef;
        ChebHeader hd_r = hd;
        asm volatile("" : "+s"(n_r), "+s"(tol_r), "+s"(s_r), "+s"(y_r), "+s"(x_r), "+s"(coef_r), "+s"(hd_r.segw),
                     "+s"(hd_r.low), "+s"(hd_r.nf), "+s"(hd_r.nseg));
        const double tol = fmax(fabs(tol_r), 1e-12);
        if (phase == 0 && !exhausted) {
            int nr = 0;
            if (lane == 0) nr = atomicAdd(&next_row, 1);
            nr = __shfl(nr, 0, 64);
            if (nr >= row_end) {
                exhausted = true;
            } else {
                r = nr;
                phase = 1;
                x_eval = (warm && isfinite(init) && init >= lo_b && init <= hi_b) ? init : 0.5 * (lo_b + hi_b);
            }
        }
        if (__syncthreads_and((phase == 0 && exhausted) ? 1 : 0)) break;
        const bool active = phase != 0;
        const float *g = grot + (int64_t)(active ? r : 0) * n;
        const double lbd = active ? fast_eval_lambda(x_eval, smin, n, dim) : -1.0;
        double acc[MAXD + 1];
#pragma unroll
        for (int k = 0; k < MAXD + 1; ++k) acc[k] = 0.0;
        double ssq = 0.0;
        for (int t0 = 0; t0 < n; t0 += tile) {
            const int t1 = min(n, t0 + tile);
            __syncthreads();                                   // the previous tile has been consumed
            // position of sample i of the tile in the LDS images: permuted in the 16-byte-load form (see PERM above)
            auto lpos = [&](int i) { return vec4 ? ((i & ~255) + ((i & 3) << 6) + ((i & 255) >> 2)) : i; };
            for (int i = threadIdx.x; i < t1 - t0; i += NW * 64) {
                const int q = lpos(i);
                ls[q] = s_r[t0 + i];
                ly[q] = y_r[t0 + i];
            }
            if (MAXD == 2) {
                for (int i = threadIdx.x; i < t1 - t0; i += NW * 64) lx[lpos(i)] = x_r[t0 + i];
            } else {                                           // covariate-major image: lx[r * tile + position]
                for (int i = threadIdx.x; i < (t1 - t0) * p; i += NW * 64) {
                    const int j = i / p, rr = i - j * p;
                    lx[rr * tile + lpos(j)] = x_r[(int64_t)t0 * p + i];
                }
            }
            __syncthreads();
            if (active && (lbd >= 0.0 || phase == 1)) {
                if (vec4)
                    fast_eval_accumulate_batched<MAXD, 4, true, true>(lbd, ls, lx, ly, g, t0, t1, t0, p, acc, ssq, phase == 1, 1,
                                                                      tile);
                else
                    fast_eval_accumulate_batched<MAXD, 8, false>(lbd, ls, lx, ly, g, t0, t1, t0, p, acc, ssq, phase == 1, 1, tile);
            }
        }
        if (!active) continue;
        FastEval<MAXD> ev;
        ev.ok = false;
        ev.reml_neg = 1e8;
        ev.q = 0.0;
        ev.logdetv = 0.0;
        ev.beta_k = 0.0;
        ev.ainv_kk = 0.0;
        if (lbd >= 0.0) fast_eval_finish<MAXD>(x_eval, hd_r, coef_r, n_r, p, phase == 3, smin_ptr + 8, acc, ev);
        double *o = out + (int64_t)r * out_cols;
        if (LMM2 && phase >= 4) {
            // ---- second search: -ML (neg_ml of k_scan.hip: 1e8 on any failure), Brent seeded with the REML optimum
            double mlneg = 1e8;
            if (ev.ok && isfinite(ev.q) && ev.q > 0.0) {
                const double nf = (double)n_r;
                const double ml = nf * (jx_log(nf) - 1.0 - jx_log(2.0 * M_PI)) / 2.0 - 0.5 * (nf * jx_log(ev.q) + ev.logdetv);
                if (isfinite(ml)) mlneg = -ml;
            }
            BrentState b;
            if (phase == 4) {
                b.a = lo_b;
                b.c = hi_b;
                b.x = x_eval;
                b.w = b.x;
                b.v = b.x;
                b.d = 0.0;
                b.e = 0.0;
                b.it = 0;
                b.fx = mlneg;
                b.fw = b.fx;
                b.fv = b.fx;
                b.evals = 1;
            } else {
                b = *bslot;
                brent_update(b, x_eval, mlneg);
            }
            double u;
            if (brent_propose(b, tol, max_iter, u)) {
                phase = 5;
                x_eval = u;
                if (lane == 0) *bslot = b;
                continue;
            }
            if (lane == 0) {
                const double ml_alt = -b.fx;
                double stat = isfinite(ml_alt) ? 2.0 * (ml_alt - nullml) : 0.0;
                if (!isfinite(stat) || stat < 0.0) stat = 0.0;
                const double plrt = chi2_sf_df1_dev(stat);
                o[0] = kslot[0];
                o[1] = kslot[1];
                o[2] = kslot[2];
                o[3] = kslot[3];
                o[4] = ml_alt;
                o[5] = isfinite(plrt) ? plrt : 1.0;
            }
            phase = 0;
            continue;
        }
        if (phase == 1) {
            ssq = wave_allsum(ssq);
            if (!isfinite(ssq) || ssq <= 1e-12) {
                if (lane == 0) {
                    o[0] = nan("");
                    o[1] = nan("");
                    o[2] = 1.0;
                    if (LMM2) {
                        o[3] = nan("");
                        o[4] = nan("");
                        o[5] = 1.0;
                    } else if (with_plrt) {
                        o[3] = 1.0;
                    }
                    if (evals_out) evals_out[r] = 0;
                }
                phase = 0;
                continue;
            }
        }
        if (phase == 1 || phase == 2) {
            BrentState b;
            if (phase == 1) {
                b.a = lo_b;
                b.c = hi_b;
                b.x = x_eval;
                b.w = b.x;
                b.v = b.x;
                b.d = 0.0;
                b.e = 0.0;
                b.it = 0;
                b.fx = ev.reml_neg;
                b.fw = b.fx;
                b.fv = b.fx;
                b.evals = 1;
            } else {
                b = *bslot;                                // written by this wave's lane 0 at the end of its previous round
                brent_update(b, x_eval, ev.reml_neg);
            }
            double u;
            if (brent_propose(b, tol, max_iter, u)) {
                phase = 2;
                x_eval = u;
            } else {
                phase = 3;
                x_eval = b.x;
            }
            if (lane == 0) *bslot = b;
            continue;
        }
        // phase 3: final_beta_se (src/stats/reml.rs:472-568) at the optimum
        double beta = nan(""), se = nan("");
        if (ev.ok) {
            const double sigma2 = ev.q / ((double)n_r - (double)dim);
            const double var = sigma2 * ev.ainv_kk;
            if (var > 0.0 && isfinite(var)) {
                beta = ev.beta_k;
                se = sqrt(var);
            }
        }
        if (LMM2) {
            if (isfinite(beta) && isfinite(se) && se > 0.0) {
                if (lane == 0) {
                    double pv = 2.0 * (0.5 * jx_erfc(fabs(beta / se) / 1.4142135623730951));
                    if (pv < 2.2250738585072014e-308) pv = 2.2250738585072014e-308;
                    if (pv > 1.0) pv = 1.0;
                    kslot[0] = beta;
                    kslot[1] = se;
                    kslot[2] = isfinite(pv) ? pv : 1.0;
                    kslot[3] = jx_pow10(x_eval);
                }
                phase = 4;                                 // x_eval stays: the ML search starts at the REML optimum
            } else {
                if (lane == 0) {
                    o[0] = nan("");
                    o[1] = nan("");
                    o[2] = 1.0;
                    o[3] = nan("");
                    o[4] = nan("");
                    o[5] = 1.0;
                }
                phase = 0;
            }
            continue;
        }
        if (lane == 0) {
            if (evals_out) evals_out[r] = bslot->evals;
            if (isfinite(beta) && isfinite(se) && se > 0.0) {
                const double z = beta / se;
                double pv = 2.0 * (0.5 * jx_erfc(fabs(z) / 1.4142135623730951));
                if (pv < 2.2250738585072014e-308) pv = 2.2250738585072014e-308;
                if (pv > 1.0) pv = 1.0;
                o[0] = beta;
                o[1] = se;
                o[2] = isfinite(pv) ? pv : 1.0;
                if (with_plrt) {
                    double plrt = 1.0;
                    if (ev.ok && isfinite(ev.q) && ev.q > 0.0) {
                        const double nf = (double)n_r;
                        const double ml =
                            nf * (jx_log(nf) - 1.0 - jx_log(2.0 * M_PI)) / 2.0 - 0.5 * (nf * jx_log(ev.q) + ev.logdetv);
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
        phase = 0;
    }
}

}  // namespace jx

using namespace jx;

#define JX_DISPATCH_DIM_F(dim, EXPR)                     \
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

extern "C" int jxg_lmm_scan_exact(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                                  const double *d_y, int p, double low, double high, double tol, int max_iter,
                                  int warm, double init_log10_lbd, int with_plrt, double nullml, double *d_out,
                                  int32_t *d_evals, void *stream);

// Workspace layout (doubles): [0] smin | [8..8+p) beta_mid (the shift of y) | [CH_HDR..CH_HDR+n) y_c | coef (total_funcs*CH_N) |
// vals (total_funcs*CH_N)
static bool fast_path_ok(int p, double low, double high) {
    // plain form: one table value per lane (nf <= 64, always true for dim <= 4); block form: up to three per lane
    return (high - low) <= 2.0 * CH_MAXSEG && cheb_nf(p, cheb_blk(p)) <= (cheb_blk(p) ? 192 : 64) && !getenv("JXGPU_SCAN_EXACT");
}

static ChebHeader make_header(int p, double low, double high) {
    ChebHeader hd;
    // block form: segments of width <= 1 (rho = 5.8: the entries of M = Ac^-1 are as analytic as the sums but, for an
    // ill-conditioned Ac, much larger on the Bernstein ellipse than on the interval -- with width 2 two covariates collinear
    // to 1e-3 cost 3e-6 on beta; the narrower segment buys seven more digits)
    hd.nseg = (int)ceil((high - low) / (cheb_blk(p) ? 1.0 : 2.0));
    if (hd.nseg < 1) hd.nseg = 1;
    hd.segw = (high - low) / hd.nseg;
    hd.low = low;
    hd.nf = cheb_nf(p, cheb_blk(p));
    return hd;
}

// per-SNP series form (series_coef_kernel): plain evaluation tail (up to three covariates beside the SNP) and bounds of at most
// two width-2 segments (the workflow's [log10 lambda0 - 2, + 2]); JXGPU_SCAN_SERIES=0 switches it off
// The series have their OWN segmentation -- width-2 segments whatever the evaluation tail: the block form's narrower segments
// are about the conditioning of the tabulated Cholesky factor, the SNP-specific sums are plain sums.
static ChebHeader series_header(double low, double high) {
    ChebHeader hd;
    hd.nseg = (int)ceil((high - low) / 2.0);
    if (hd.nseg < 1) hd.nseg = 1;
    hd.segw = (high - low) / hd.nseg;
    hd.low = low;
    hd.nf = 0;
    return hd;
}
static bool series_ok(int p, double low, double high) {
    static const bool on = !(getenv("JXGPU_SCAN_SERIES") && atoi(getenv("JXGPU_SCAN_SERIES")) == 0);
    return on && p >= 1 && p <= 14 && series_header(low, high).nseg * CH_N <= SR_M;
}
static int64_t series_npad(int n) { return ((int64_t)n + SR_SC - 1) / SR_SC * SR_SC; }
// the y_c segment of the table workspace, rounded up to an even number of doubles: everything behind it (the coefficient
// tables, What) stays 16-byte aligned for odd n too -- series_coef_kernel reads What with dwordx4 loads
static int64_t yc_doubles(int n) { return ((int64_t)n + 1) & ~(int64_t)1; }

extern "C" int64_t jxg_lmm_tables_bytes(int n, int p, double low, double high) {
    if (!fast_path_ok(p, low, high)) return 0;
    const ChebHeader hd = make_header(p, low, high);
    const int64_t what = series_ok(p, low, high) ? series_npad(n) * SR_M : 0;     // What[i][m] behind the lambda-only tables
    return (int64_t)sizeof(double) * (CH_HDR + yc_doubles(n) + 2 * (int64_t)hd.nseg * hd.nf * CH_N + what);
}

extern "C" int jxg_lmm_tables_build(const double *d_s, const double *d_xcov, const double *d_y, int n, int p,
                                    double low, double high, void *d_work, void *stream) {
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_tables_build: p out of range");
    if (!(low < high)) return fail("low must be < high");
    if (!fast_path_ok(p, low, high)) return fail("jxg_lmm_tables_build: configuration needs the exact scan path");
    hipStream_t st = (hipStream_t)stream;
    const ChebHeader hd = make_header(p, low, high);
    const int total_funcs = hd.nseg * hd.nf;
    double *w = (double *)d_work;
    double *smin = w, *yc = w + CH_HDR, *coef = yc + yc_doubles(n), *vals = coef + (int64_t)total_funcs * CH_N;
    const double lbd_mid = pow(10.0, 0.5 * (low + high));
    JX_DISPATCH_DIM_F(p, hipLaunchKernelGGL(yshift_kernel<MAXD>, dim3(1), dim3(SCAN_THREADS), 0, st, d_s, d_xcov, d_y,
                                            n, p, lbd_mid, yc, smin));
    JX_LAUNCH_CHECK();
    JX_DISPATCH_DIM_F(p, hipLaunchKernelGGL(cheb_nodes_kernel<MAXD>, dim3(hd.nseg * CH_N), dim3(SCAN_THREADS), 0, st,
                                            d_s, d_xcov, yc, n, p, low, hd.segw, hd.nf, vals));
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(cheb_coef_kernel, dim3((total_funcs * CH_N + 255) / 256), dim3(256), 0, st, vals, total_funcs,
                       coef);
    JX_LAUNCH_CHECK();
    if (series_ok(p, low, high)) {
        double *what = vals + (int64_t)total_funcs * CH_N;
        const int npad = (int)series_npad(n);
        hipLaunchKernelGGL(series_what_kernel, dim3((npad + 63) / 64), dim3(64), 0, st, d_s, n, npad, series_header(low, high), what);
        JX_LAUNCH_CHECK();
    }
    return 0;
}

namespace jx { extern float g_last_ms[24]; }   // [11]: form the last exact scan launch took (0 LDS-resident, 1 tiled, 2 plain)

// per-SNP series of a block of rotated rows: scoef (nrows, p + 2, SR_M), sssq (nrows) -- series_coef_kernel per group of quantities
static int series_coef_launch(const float *d_grot, int nrows, int n, int p, const double *d_xcov, const void *d_work, double low,
                              double high, double *scoef, double *sssq, hipStream_t st) {
    const ChebHeader hd = make_header(p, low, high);
    const double *w = (const double *)d_work;
    const double *yc = w + CH_HDR, *coef = yc + yc_doubles(n);
    const double *what = coef + 2 * (int64_t)hd.nseg * hd.nf * CH_N;
    const int nq = p + 2;
    const int npad = (int)series_npad(n);
    const size_t lds = sizeof(double) * ((size_t)SR_SC * SR_WP + (size_t)(p + 1) * SR_SC) + sizeof(float) * 128 * SR_GP;
#define JX_SERIES_COEF(NQV, Q0)                                                                                            \
    do {                                                                                                                  \
        auto kfn = series_coef_kernel<NQV>;                                                                               \
        static bool attr_s = false;                                                                                       \
        if (!attr_s) {                                                                                                    \
            JX_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));        \
            attr_s = true;                                                                                                \
        }                                                                                                                 \
        hipLaunchKernelGGL(kfn, dim3((nrows + 127) / 128), dim3(512), lds, st, d_grot, nrows, n, npad, p, Q0, d_xcov, yc,  \
                           what, scoef, sssq);                                                                            \
        JX_LAUNCH_CHECK();                                                                                                \
    } while (0)
    for (int q0 = 0; q0 < nq;) {
        const int left = nq - q0;
        if (left == 3) { JX_SERIES_COEF(3, q0); q0 += 3; }
        else if (left >= 4) { JX_SERIES_COEF(4, q0); q0 += 4; }
        else if (left == 2) { JX_SERIES_COEF(2, q0); q0 += 2; }
        else { JX_SERIES_COEF(1, q0); q0 += 1; }
    }
#undef JX_SERIES_COEF
    return 0;
}

// Brent on stored series: one wave per SNP, or -- chain_off given -- one wave per warm-start chain
static int series_brent_launch(int nrows, int n, const double *d_s, const double *d_xcov, int p, double low, double high,
                               const void *d_work, double tol, int max_iter, int warm, double init_log10_lbd, int with_plrt,
                               double nullml, double *d_out, int32_t *d_evals, const double *scoef, const double *sssq,
                               hipStream_t st, const int32_t *chain_off, int nchains, double *carry) {
    const ChebHeader hd = make_header(p, low, high);
    const ChebHeader shd = series_header(low, high);
    const double *w = (const double *)d_work;
    const double *smin = w, *yc = w + CH_HDR, *coef = yc + yc_doubles(n);
    const int dim = p + 1;
    const bool chain = chain_off != nullptr;
    const int nunits = chain ? nchains : nrows;
    if (nunits <= 0) return 0;
    const float *d_grot = nullptr;
    constexpr int NW = 8;
    // chains: ONE wave per workgroup -- a chain is a long sequential job and there are few of them (m / 512): spread over the
    // CUs they do not share a SIMD's issue slots
    const int grid = chain ? nunits : (nunits + NW - 1) / NW;
    const int maxd_use = dim <= 2 ? 2 : (dim <= 4 ? 4 : (dim <= 8 ? 8 : 16));
    const size_t ncoef_pad = ((size_t)hd.nseg * hd.nf * CH_N + 1) & ~(size_t)1;
    // Brent on an interpolant of the objective (plain form, dim <= 4; JXGPU_SCAN_INTERP=0: direct evaluations): see plain_objective
    static const bool interp_env = !(getenv("JXGPU_SCAN_INTERP") && atoi(getenv("JXGPU_SCAN_INTERP")) == 0);
    // plain form (dim <= 4): tables and series share their segments; block form (dim 5 - 8): the per-lane factor of a 7 x 7 block
    // is 56 registers -- beyond that (MAXD = 16) the direct evaluations stay
    const bool interp = interp_env && ((dim <= 4 && hd.nseg == shd.nseg && hd.segw == shd.segw && hd.low == shd.low) ||
                                       (dim >= 5 && dim <= 8));
    if (interp) {
        constexpr int NWI = 4;
        const int nw = chain ? 1 : NWI;
        const size_t lds = sizeof(double) * ((size_t)CH_N * IP_CP + 1 + ncoef_pad +
                                             (size_t)nw * (2 * (size_t)(maxd_use + 1) * SR_M + 2 * SR_M));
        if (lds > 160 * 1024) return fail("jxg_lmm_series_brent_tab: the interpolant form's tables do not fit LDS");
        const int gridi = chain ? nunits : (nunits + NWI - 1) / NWI;
        if (lds > 64 * 1024) {
            static bool attr_i = false;
            if (!attr_i) {
                JX_HIP(hipFuncSetAttribute((const void *)lmm_scan_fast_kernel<8, 1, false, true, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                JX_HIP(hipFuncSetAttribute((const void *)lmm_scan_fast_kernel<8, NWI, false, true, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                JX_HIP(hipFuncSetAttribute((const void *)lmm_scan_fast_kernel<8, 1, false, true, true, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                JX_HIP(hipFuncSetAttribute((const void *)lmm_scan_fast_kernel<8, NWI, false, true, true, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_i = true;
            }
        }
        // chains: three launches (ChainSplit) unless JXGPU_SCAN_CHAIN_SPLIT=0
        const bool split_on = chain && !(getenv("JXGPU_SCAN_CHAIN_SPLIT") && atoi(getenv("JXGPU_SCAN_CHAIN_SPLIT")) == 0);
        if (split_on) {
            async_pool_keep();
            ChainSplit sp;
            if (getenv("JXGPU_SCAN_CHAIN_FORCE_DIRECT")) sp.force_row = atoi(getenv("JXGPU_SCAN_CHAIN_FORCE_DIRECT"));
            int32_t *cflag = nullptr;
            const size_t b_ic = sizeof(double) * (size_t)nrows * SR_M, b_x = sizeof(double) * (size_t)nrows;
            const size_t b_rf = (sizeof(int32_t) * (size_t)nrows + 255) & ~(size_t)255;
            AsyncBlock ab;
            if (ab.alloc(b_ic + b_x + b_rf + sizeof(int32_t) * (size_t)nchains, st)) return 1;
            char *blk = (char *)ab.p;
            sp.icoef = (double *)blk;
            sp.xopt = (double *)(blk + b_ic);
            sp.rflag = (int32_t *)(blk + b_ic + b_x);
            cflag = (int32_t *)(blk + b_ic + b_x + b_rf);
            JX_HIP(hipMemsetAsync(sp.xopt, 0xff, b_x, st));                       // NaN: not a row of the split form
            const size_t lds_row = sizeof(double) * ((size_t)CH_N * IP_CP + 1 + ncoef_pad +
                                                     (size_t)NWI * (2 * (size_t)(maxd_use + 1) * SR_M + 2 * SR_M));
            const size_t lds_chain = lds;
            if (lds_row > 160 * 1024) return fail("jxg_lmm_series_brent_tab: the interpolant form's tables do not fit LDS");
            const int grid_row = (nrows + NWI - 1) / NWI;
            const int32_t *no_chain = nullptr;
            double *no_carry = nullptr;
#define JX_SPLIT_ROWS(MAXDV, PHASE)                                                                                        \
    do {                                                                                                                  \
        ChainSplit spp = sp;                                                                                              \
        spp.phase = PHASE;                                                                                                \
        hipLaunchKernelGGL((lmm_scan_fast_kernel<MAXDV, NWI, false, true, true, true>), dim3(grid_row), dim3(NWI * 64), lds_row, st, d_grot, \
                           nrows, n, d_s, d_xcov, yc, p, hd, coef, smin, low, high, tol, max_iter, 0, 0.0, with_plrt, nullml,    \
                           d_out, d_evals, scoef, sssq, shd, no_chain, 0, no_carry, spp);                                       \
        JX_LAUNCH_CHECK();                                                                                                \
    } while (0)
#define JX_SPLIT_REST(MAXDV)                                                                                               \
    do {                                                                                                                  \
        ChainSplit spp = sp;                                                                                              \
        spp.phase = 0;                                                                                                    \
        spp.cflag = cflag;                                                                                                \
        hipLaunchKernelGGL((lmm_scan_fast_kernel<MAXDV, 1, false, true, true, true>), dim3(gridi), dim3(64), lds_chain, st, d_grot, nrows, n, \
                           d_s, d_xcov, yc, p, hd, coef, smin, low, high, tol, max_iter, warm, init_log10_lbd, with_plrt, nullml, \
                           d_out, d_evals, scoef, sssq, shd, chain_off, nchains, carry, spp);                                   \
        JX_LAUNCH_CHECK();                                                                                                \
    } while (0)
#define JX_SPLIT_ALL(MAXDV)                                                                                                \
    do {                                                                                                                  \
        JX_SPLIT_ROWS(MAXDV, 1);                                                                                          \
        hipLaunchKernelGGL(chain_interp_brent_kernel, dim3(nchains), dim3(64), 0, st, sp.icoef, sp.rflag, chain_off, nchains, carry, \
                           cflag, sp.xopt, d_evals, shd, low, high, tol, max_iter);                                             \
        JX_LAUNCH_CHECK();                                                                                                \
        JX_SPLIT_REST(MAXDV);                                                                                             \
        JX_SPLIT_ROWS(MAXDV, 2);                                                                                          \
    } while (0)
            if (dim <= 2) JX_SPLIT_ALL(2);
            else if (dim <= 4) JX_SPLIT_ALL(4);
            else JX_SPLIT_ALL(8);
#undef JX_SPLIT_ALL
#undef JX_SPLIT_REST
#undef JX_SPLIT_ROWS
            return 0;
        }
#define JX_SERIES_INTERP(MAXDV)                                                                                            \
    do {                                                                                                                  \
        if (chain)                                                                                                        \
            hipLaunchKernelGGL((lmm_scan_fast_kernel<MAXDV, 1, false, true, true>), dim3(gridi), dim3(64), lds, st, d_grot, nrows, n,     \
                               d_s, d_xcov, yc, p, hd, coef, smin, low, high, tol, max_iter, warm, init_log10_lbd, with_plrt,     \
                               nullml, d_out, d_evals, scoef, sssq, shd, chain_off, nchains, carry);                              \
        else                                                                                                              \
            hipLaunchKernelGGL((lmm_scan_fast_kernel<MAXDV, NWI, false, true, true>), dim3(gridi), dim3(NWI * 64), lds, st, d_grot, nrows, \
                               n, d_s, d_xcov, yc, p, hd, coef, smin, low, high, tol, max_iter, warm, init_log10_lbd, with_plrt,  \
                               nullml, d_out, d_evals, scoef, sssq, shd);                                                         \
    } while (0)
        if (dim <= 2) JX_SERIES_INTERP(2);
        else if (dim <= 4) JX_SERIES_INTERP(4);
        else JX_SERIES_INTERP(8);
#undef JX_SERIES_INTERP
        JX_LAUNCH_CHECK();
        return 0;
    }
    // LDS of the chain instantiation: the lambda-only coefficient tables + two buffers of one SNP's series (MAXD + 1 rows of 64)
    const size_t chain_lds = sizeof(double) * (ncoef_pad + 2 * (size_t)(maxd_use + 1) * SR_M);
    if (chain && chain_lds > 160 * 1024) return fail("jxg_lmm_series_brent_tab: the chain form's tables do not fit LDS");
    if (chain && chain_lds > 64 * 1024) {
#define JX_CHAIN_ATTR(MAXDV)                                                                                              \
    do {                                                                                                                  \
        static bool attr_c = false;                                                                                       \
        if (!attr_c) {                                                                                                    \
            JX_HIP(hipFuncSetAttribute((const void *)lmm_scan_fast_kernel<MAXDV, 1, false, true>,                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                          \
            attr_c = true;                                                                                                \
        }                                                                                                                 \
    } while (0)
        if (dim <= 2) JX_CHAIN_ATTR(2);
        else if (dim <= 4) JX_CHAIN_ATTR(4);
        else if (dim <= 8) JX_CHAIN_ATTR(8);
        else JX_CHAIN_ATTR(16);
#undef JX_CHAIN_ATTR
    }
#define JX_SERIES_BRENT(MAXDV)                                                                                             \
    do {                                                                                                                  \
        if (chain)                                                                                                        \
            hipLaunchKernelGGL((lmm_scan_fast_kernel<MAXDV, 1, false, true>), dim3(grid), dim3(64), chain_lds, st, d_grot, nrows, n, d_s, \
                               d_xcov, yc, p, hd, coef, smin, low, high, tol, max_iter, warm, init_log10_lbd, with_plrt, nullml,  \
                               d_out, d_evals, scoef, sssq, shd, chain_off, nchains, carry);                                      \
        else                                                                                                              \
            hipLaunchKernelGGL((lmm_scan_fast_kernel<MAXDV, NW, false, true>), dim3(grid), dim3(NW * 64), 0, st, d_grot, nrows, n, \
                               d_s, d_xcov, yc, p, hd, coef, smin, low, high, tol, max_iter, warm, init_log10_lbd, with_plrt,     \
                               nullml, d_out, d_evals, scoef, sssq, shd);                                                         \
    } while (0)
    if (dim <= 2) JX_SERIES_BRENT(2);
    else if (dim <= 4) JX_SERIES_BRENT(4);
    else if (dim <= 8) JX_SERIES_BRENT(8);
    else JX_SERIES_BRENT(16);
#undef JX_SERIES_BRENT
    JX_LAUNCH_CHECK();
    return 0;
}

// chain_off / nchains / carry: the warm-start chains of the reference (device pointers; see lmm_scan_fast_kernel), or nullptr
static int lmm_scan_tab_impl(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov, int p,
                             double low, double high, const void *d_work, double tol, int max_iter, int warm,
                             double init_log10_lbd, int with_plrt, double nullml, double *d_out, int32_t *d_evals,
                             void *stream, const int32_t *chain_off, int nchains, double *carry) {
    if (nrows <= 0) return 0;
    const bool chain = chain_off != nullptr;
    if (chain && nchains <= 0) return 0;
    const int nunits = chain ? nchains : nrows;      // waves of work: one per chain, or one per row
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_scan_tab: p out of range");
    if (!fast_path_ok(p, low, high)) return fail("jxg_lmm_scan_tab: configuration needs the exact scan path");
    const ChebHeader hd = make_header(p, low, high);
    const double *w = (const double *)d_work;
    const double *smin = w, *yc = w + CH_HDR, *coef = yc + yc_doubles(n);
    const int dim = p + 1;
    // LDS-resident s / X~ / y~ when they fit one workgroup's share (one workgroup per CU: 160 KB less a margin)
    const size_t lds_bytes = sizeof(double) * (size_t)n * (size_t)(2 + p);
    const bool lmm2 = with_plrt == 2;               // the two-search scan (out has six columns): tiled kernel at every n
    if (chain && lmm2) return fail("jxg_lmm_scan_chain: the two-search scan has no chain form");
    const bool lds_fits0 = !lmm2 && dim <= 4 && lds_bytes <= (size_t)156 * 1024 && !getenv("JXGPU_SCAN_NOLDS");
    // per-SNP series form (one MFMA pass over the rows + Brent on the series) wherever s / X~ / y~ do not stay in LDS for the
    // one-wave-per-SNP form (measured at n = 5000, intercept only: LDS-resident form 6.8 ms, series form 10.5 ms per 50 000 SNPs).
    // Along chains the series form is taken wherever it exists: the pass over the rows stays parallel over all SNPs, only the
    // Brent searches on the series run in chain order (one wave per chain), where an evaluation is a Clenshaw recurrence
    const bool series = !lmm2 && (chain || !lds_fits0) && series_ok(p, low, high) && !getenv("JXGPU_SCAN_NOTILE");
    const bool lds_fits = lds_fits0 && !series;
    const bool use_lds = lds_fits;
    if (use_lds) {
        g_last_ms[11] = 0.f;
        if (dim <= 2) {
            constexpr int NW = 16;
            auto kfn = lmm_scan_fast_kernel<2, NW, true>;
            static bool attr_done = false;
            if (!attr_done) {
                JX_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_done = true;
            }
            int grid = (nunits + NW - 1) / NW;
            if (grid > 65536) grid = 65536;
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), lds_bytes, (hipStream_t)stream, d_grot, nrows, n, d_s,
                               d_xcov, yc, p, hd, coef, smin, low, high, tol, max_iter, warm, init_log10_lbd, with_plrt,
                               nullml, d_out, d_evals, (const double *)nullptr, (const double *)nullptr, hd, chain_off, nchains,
                               carry, ChainSplit());
        } else {
            constexpr int NW = 8;
            auto kfn = lmm_scan_fast_kernel<4, NW, true>;
            static bool attr_done4 = false;
            if (!attr_done4) {
                JX_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_done4 = true;
            }
            int grid = (nunits + NW - 1) / NW;
            if (grid > 65536) grid = 65536;
            hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), lds_bytes, (hipStream_t)stream, d_grot, nrows, n, d_s,
                               d_xcov, yc, p, hd, coef, smin, low, high, tol, max_iter, warm, init_log10_lbd, with_plrt,
                               nullml, d_out, d_evals, (const double *)nullptr, (const double *)nullptr, hd, chain_off, nchains,
                               carry, ChainSplit());
        }
        JX_LAUNCH_CHECK();
        return 0;
    }
    if (series) {
        // n beyond the LDS-resident limit (and every chain scan): ONE streaming pass over the rotated rows per group of four
        // quantities builds every SNP's Chebyshev series of its SNP-specific sums on the f64 matrix pipes, Brent then runs on
        // the series
        g_last_ms[11] = 3.f;
        hipStream_t st = (hipStream_t)stream;
        const int nq = p + 2;
        // series + sums of squares of THIS call: stream-ordered allocation, released behind the Brent kernel on the same stream
        // (a process-wide buffer would be shared by concurrent calls on other streams while their kernels are in flight: ADVICE r4)
        const size_t need = sizeof(double) * ((size_t)nrows * nq * SR_M + (size_t)nrows);
        void *sraw = nullptr;
        async_pool_keep();
        JX_HIP(hipMallocAsync(&sraw, need, st));
        struct SeriesFree {
            void *p;
            hipStream_t s;
            ~SeriesFree() { (void)hipFreeAsync(p, s); }
        } sfree{sraw, st};
        double *scoef = (double *)sraw, *sssq = scoef + (size_t)nrows * nq * SR_M;
        if (series_coef_launch(d_grot, nrows, n, p, d_xcov, d_work, low, high, scoef, sssq, st)) return 1;
        return series_brent_launch(nrows, n, d_s, d_xcov, p, low, high, d_work, tol, max_iter, warm, init_log10_lbd, with_plrt,
                                   nullml, d_out, d_evals, scoef, sssq, st, chain_off, nchains, carry);
    }
    if (!chain && (lmm2 || !getenv("JXGPU_SCAN_NOTILE"))) {
        // n beyond the LDS-resident limit: tiles of the shared vectors in LDS, NW SNPs per workgroup in lock step
        g_last_ms[11] = 1.f;
        const int per_sample = 8 * (2 + p);
        int tile = (150 * 1024 / per_sample) / 256 * 256;
        const int ntiles = (n + tile - 1) / tile;
        tile = ((n + ntiles - 1) / ntiles + 255) / 256 * 256;
        const size_t lds_tile = (size_t)per_sample * tile + 16 + 16 * sizeof(BrentState) + 16 * 4 * sizeof(double);   // + queue head + one Brent state per wave + the LMM2 slots
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t prop;
            JX_HIP(hipGetDevice(&dev));
            JX_HIP(hipGetDeviceProperties(&prop, dev));
            cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        // dim 2 - 4: 16 SNPs per workgroup (<= 128 VGPRs), plain evaluation tail; dim 5 - 8: 12 SNPs, dim 9 - 16: 8, block form
        // (fast_eval_finish_blk)
#define JX_TILED_LAUNCH(MAXDV, NWV, L2)                                                                                   \
    do {                                                                                                                  \
        constexpr int NW = NWV;                                                                                           \
        auto kfn = lmm_scan_tiled_kernel<MAXDV, NW, L2>;                                                                  \
        static bool attr_t = false;                                                                                       \
        if (!attr_t) {                                                                                                    \
            JX_HIP(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 157 * 1024));       \
            attr_t = true;                                                                                                \
        }                                                                                                                 \
        int grid = cus;                                                                                                   \
        if (grid * NW > nrows) grid = (nrows + NW - 1) / NW;                                                              \
        const int rows_per_wg = (nrows + grid - 1) / grid;                                                                \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), lds_tile, (hipStream_t)stream, d_grot, nrows, n, d_s, d_xcov,  \
                           yc, p, hd, coef, smin, low, high, tol, max_iter, warm, init_log10_lbd, with_plrt, nullml,      \
                           d_out, d_evals, tile, rows_per_wg);                                                            \
    } while (0)
#define JX_TILED_BY_DIM(L2)                                                                                               \
    do {                                                                                                                  \
        if (dim <= 2) JX_TILED_LAUNCH(2, 16, L2);                                                                         \
        else if (dim <= 4) JX_TILED_LAUNCH(4, 16, L2);                                                                    \
        else if (dim <= 8) JX_TILED_LAUNCH(8, 12, L2);                                                                    \
        else JX_TILED_LAUNCH(16, 8, L2);                                                                                  \
    } while (0)
        if (lmm2) JX_TILED_BY_DIM(true);
        else JX_TILED_BY_DIM(false);
#undef JX_TILED_BY_DIM
#undef JX_TILED_LAUNCH
        JX_LAUNCH_CHECK();
        return 0;
    }
    g_last_ms[11] = 2.f;
    int grid = (nunits + SCAN_WAVES - 1) / SCAN_WAVES;
    if (grid > 65536 * 8) grid = 65536 * 8;
    JX_DISPATCH_DIM_F(dim, hipLaunchKernelGGL((lmm_scan_fast_kernel<MAXD, SCAN_WAVES, false>), dim3(grid),
                                              dim3(SCAN_THREADS), 0, (hipStream_t)stream, d_grot, nrows, n, d_s, d_xcov,
                                              yc, p, hd, coef, smin, low, high, tol, max_iter, warm, init_log10_lbd,
                                              with_plrt, nullml, d_out, d_evals, (const double *)nullptr,
                                              (const double *)nullptr, hd, chain_off, nchains, carry));
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_lmm_scan_tab(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov, int p,
                                double low, double high, const void *d_work, double tol, int max_iter, int warm,
                                double init_log10_lbd, int with_plrt, double nullml, double *d_out, int32_t *d_evals,
                                void *stream) {
    return lmm_scan_tab_impl(d_grot, nrows, n, d_s, d_xcov, p, low, high, d_work, tol, max_iter, warm, init_log10_lbd, with_plrt,
                             nullml, d_out, d_evals, stream, nullptr, 0, nullptr);
}

// The exact scan along the reference's warm-start chains (`carry_warm_start`, src/stats/lmm.rs:134-161; the default of
// `lmm_reml_assoc_packed_f32` :3244-3245 and of the BED route :2627).  d_chain_off: nchains + 1 ascending row offsets into this
// block of rotated rows (device, int32); d_carry: nchains doubles (device) -- the log10 lambda a chain starts from (NaN: none, the
// first valid SNP starts from the interval midpoint) on entry, the optimum of the chain's last valid SNP on return, so that a
// chain cut by the caller's blocking continues in the next call.  Tables as for jxg_lmm_scan_tab.
extern "C" int jxg_lmm_scan_chain_tab(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov, int p,
                                      double low, double high, const void *d_work, double tol, int max_iter,
                                      const int32_t *d_chain_off, int nchains, double *d_carry, int with_plrt, double nullml,
                                      double *d_out, int32_t *d_evals, void *stream) {
    if (!d_chain_off || !d_carry) return fail("jxg_lmm_scan_chain_tab: chain offsets and carry states are required");
    if (with_plrt != 0 && with_plrt != 1) return fail("jxg_lmm_scan_chain_tab: with_plrt must be 0 or 1");
    return lmm_scan_tab_impl(d_grot, nrows, n, d_s, d_xcov, p, low, high, d_work, tol, max_iter, 0, 0.0, with_plrt, nullml,
                             d_out, d_evals, stream, d_chain_off, nchains, d_carry);
}

extern "C" int jxg_lmm_scan_exact_chain(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                                        const double *d_y, int p, double low, double high, double tol, int max_iter,
                                        const int32_t *d_chain_off, int nchains, double *d_carry, int with_plrt, double nullml,
                                        double *d_out, int32_t *d_evals, void *stream);

// Chain scan of one block with the tables built here (host entry points; pipeline.scan_rows builds them once per call).
extern "C" int jxg_lmm_scan_chain(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                                  const double *d_y, int p, double low, double high, double tol, int max_iter,
                                  const int32_t *d_chain_off, int nchains, double *d_carry, int with_plrt, double nullml,
                                  double *d_out, int32_t *d_evals, void *stream) {
    if (nrows <= 0 || nchains <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_scan_chain: p out of range");
    if (!(low < high)) return fail("low must be < high");
    if (!fast_path_ok(p, low, high))
        return jxg_lmm_scan_exact_chain(d_grot, nrows, n, d_s, d_xcov, d_y, p, low, high, tol, max_iter, d_chain_off, nchains,
                                        d_carry, with_plrt, nullml, d_out, d_evals, stream);
    hipStream_t st = (hipStream_t)stream;
    void *work = nullptr;
    async_pool_keep();
    JX_HIP(hipMallocAsync(&work, (size_t)jxg_lmm_tables_bytes(n, p, low, high), st));
    int rc = jxg_lmm_tables_build(d_s, d_xcov, d_y, n, p, low, high, work, stream);
    if (!rc)
        rc = jxg_lmm_scan_chain_tab(d_grot, nrows, n, d_s, d_xcov, p, low, high, work, tol, max_iter, d_chain_off, nchains, d_carry,
                                    with_plrt, nullml, d_out, d_evals, stream);
    (void)hipFreeAsync(work, st);
    return rc;
}

// ---- the series form in two steps (chain scans of a whole payload): the series of every block of rotated rows are kept, ONE Brent
// launch then walks all chains -- m / 512 sequential jobs are too few per block to fill the device block by block.
// jxg_lmm_series_doubles: doubles per row of the series storage ((p + 2) SR_M), 0 when this (p, low, high) has no series form.
extern "C" int64_t jxg_lmm_series_doubles(int p, double low, double high) {
    return (fast_path_ok(p, low, high) && series_ok(p, low, high)) ? (int64_t)(p + 2) * SR_M : 0;
}
extern "C" int jxg_lmm_series_coef_tab(const float *d_grot, int nrows, int n, const double *d_xcov, int p, double low, double high,
                                       const void *d_work, double *d_scoef, double *d_ssq, void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_series_coef_tab: p out of range");
    if (!jxg_lmm_series_doubles(p, low, high)) return fail("jxg_lmm_series_coef_tab: this configuration has no series form");
    return series_coef_launch(d_grot, nrows, n, p, d_xcov, d_work, low, high, d_scoef, d_ssq, (hipStream_t)stream);
}
// Brent on the stored series of `nrows` rows.  d_chain_off == NULL: one search per row (warm / init as jxg_lmm_scan_tab);
// otherwise along the chains (nchains + 1 offsets, d_carry as jxg_lmm_scan_chain_tab).
extern "C" int jxg_lmm_series_brent_tab(int nrows, int n, const double *d_s, const double *d_xcov, int p, double low, double high,
                                        const void *d_work, double tol, int max_iter, int warm, double init_log10_lbd,
                                        const double *d_scoef, const double *d_ssq, const int32_t *d_chain_off, int nchains,
                                        double *d_carry, int with_plrt, double nullml, double *d_out, int32_t *d_evals,
                                        void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_series_brent_tab: p out of range");
    if (!jxg_lmm_series_doubles(p, low, high)) return fail("jxg_lmm_series_brent_tab: this configuration has no series form");
    if (d_chain_off && !d_carry) return fail("jxg_lmm_series_brent_tab: chains need their carry states");
    if (with_plrt != 0 && with_plrt != 1) return fail("jxg_lmm_series_brent_tab: with_plrt must be 0 or 1");
    g_last_ms[11] = 3.f;
    return series_brent_launch(nrows, n, d_s, d_xcov, p, low, high, d_work, tol, max_iter, warm, init_log10_lbd, with_plrt, nullml,
                               d_out, d_evals, d_scoef, d_ssq, (hipStream_t)stream, d_chain_off, nchains, d_carry);
}

extern "C" int jxg_lmm_scan(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                            const double *d_y, int p, double low, double high, double tol, int max_iter, int warm,
                            double init_log10_lbd, int with_plrt, double nullml, double *d_out, int32_t *d_evals,
                            void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm_scan: p out of range");
    if (!(low < high)) return fail("low must be < high");
    if (!fast_path_ok(p, low, high))
        return jxg_lmm_scan_exact(d_grot, nrows, n, d_s, d_xcov, d_y, p, low, high, tol, max_iter, warm,
                                  init_log10_lbd, with_plrt, nullml, d_out, d_evals, stream);
    hipStream_t st = (hipStream_t)stream;
    void *work = nullptr;
    async_pool_keep();
    JX_HIP(hipMallocAsync(&work, (size_t)jxg_lmm_tables_bytes(n, p, low, high), st));
    int rc = jxg_lmm_tables_build(d_s, d_xcov, d_y, n, p, low, high, work, stream);
    if (!rc)
        rc = jxg_lmm_scan_tab(d_grot, nrows, n, d_s, d_xcov, p, low, high, work, tol, max_iter, warm, init_log10_lbd,
                              with_plrt, nullml, d_out, d_evals, stream);
    (void)hipFreeAsync(work, st);
    return rc;
}

extern "C" int jxg_lmm2_scan_exact(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                                   const double *d_y, int p, double low, double high, double tol, int max_iter, int warm,
                                   double init_log10_lbd, double nullml, double *d_out, void *stream);

// The LMM2 scan (src/stats/lmm.rs:202-330) through the tabulated evaluations: two Brent searches per SNP (REML, then ML
// seeded with the REML optimum) inside the tiled kernel; configurations outside the tables go to the reference-formulation
// kernel (jxg_lmm2_scan_exact).
extern "C" int jxg_lmm2_scan(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                             const double *d_y, int p, double low, double high, double tol, int max_iter, int warm,
                             double init_log10_lbd, double nullml, double *d_out, void *stream) {
    if (nrows <= 0) return 0;
    if (p < 1 || p > JXG_MAX_COV) return fail("jxg_lmm2_scan: p out of range");
    if (!(low < high)) return fail("low must be < high");
    if (!fast_path_ok(p, low, high) || getenv("JXGPU_LMM2_EXACT"))
        return jxg_lmm2_scan_exact(d_grot, nrows, n, d_s, d_xcov, d_y, p, low, high, tol, max_iter, warm, init_log10_lbd,
                                   nullml, d_out, stream);
    hipStream_t st = (hipStream_t)stream;
    void *work = nullptr;
    async_pool_keep();
    JX_HIP(hipMallocAsync(&work, (size_t)jxg_lmm_tables_bytes(n, p, low, high), st));
    int rc = jxg_lmm_tables_build(d_s, d_xcov, d_y, n, p, low, high, work, stream);
    if (!rc)
        rc = jxg_lmm_scan_tab(d_grot, nrows, n, d_s, d_xcov, p, low, high, work, tol, max_iter, warm, init_log10_lbd, 2,
                              nullml, d_out, nullptr, stream);
    (void)hipFreeAsync(work, st);
    return rc;
}
