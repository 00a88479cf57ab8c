// Shared device helpers of the scan kernels (reductions, tiny dense Cholesky with compile-time indexing).
#pragma once
#include "jx_common.h"

namespace jx {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_WAVES = SCAN_THREADS / 64;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum NV per-thread values over the workgroup; result broadcast to all threads (in place).
template <int NV>
__device__ __forceinline__ void block_sum(double *v, int nv, double *red /* [SCAN_WAVES][NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (k < nv) {
            const double s = wave_sum(v[k]);
            if (lane == 0) red[wave * NV + k] = s;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (k < nv) {
            double s = red[k];
#pragma unroll
            for (int w = 1; w < SCAN_WAVES; ++w) s += red[w * NV + k];
            v[k] = s;
        }
    }
    __syncthreads();
}

// all-lanes butterfly sum within a wave (every lane ends with the total; no LDS, no barrier)
__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Cooperation policy of one objective evaluation: a whole 256-thread workgroup (null model: one problem, large n)
// or a single wave (scan: one SNP per wave, four SNPs in flight per workgroup, no barriers at all).
template <bool WAVE>
struct Par {
    static constexpr int kThreads = WAVE ? 64 : SCAN_THREADS;
    static __device__ __forceinline__ int tid() { return WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x; }
    template <int NV>
    static __device__ __forceinline__ void sum(double *v, int nv, double *shm) {
        if constexpr (WAVE) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (k < nv) v[k] = wave_allsum(v[k]);
        } else {
            block_sum<NV>(v, nv, shm);
        }
    }
};

// src/math/linalg.rs:341-363. Fully unrolled with compile-time indices (runtime-indexed register arrays
// would be demoted to scratch); on a failed pivot the factorisation continues on garbage and reports false.
template <int MAXD>
__device__ __forceinline__ bool chol_inplace(double *a, int dim) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {
        if (i < dim) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double sum = a[i * MAXD + j];
#pragma unroll
                for (int k = 0; k < j; ++k) sum -= a[i * MAXD + k] * a[j * MAXD + k];
                if (i == j) {
                    if (!(sum > 1e-18)) ok = false;
                    a[i * MAXD + j] = sqrt(sum);
                } else {
                    a[i * MAXD + j] = sum / a[j * MAXD + j];
                }
            }
        }
    }
    return ok;
}

// src/stats/reml.rs:46-66 (forward then backward substitution with the lower factor)
template <int MAXD>
__device__ __forceinline__ void chol_solve(const double *l, int dim, const double *b, double *x) {
    double y[MAXD];
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {
        y[i] = 0.0;
        if (i < dim) {
            double sum = b[i];
#pragma unroll
            for (int k = 0; k < i; ++k) sum -= l[i * MAXD + k] * y[k];
            y[i] = sum / l[i * MAXD + i];
        }
    }
#pragma unroll
    for (int i = 0; i < MAXD; ++i) x[i] = 0.0;
#pragma unroll
    for (int ii = 0; ii < MAXD; ++ii) {
        const int i = MAXD - 1 - ii;
        if (i < dim) {
            double sum = y[i];
#pragma unroll
            for (int k = i + 1; k < MAXD; ++k)
                if (k < dim) sum -= l[k * MAXD + i] * x[k];
            x[i] = sum / l[i * MAXD + i];
        }
    }
}

template <int MAXD>
__device__ __forceinline__ double pick(const double *v, int idx) {  // v[idx] with compile-time indexing
    double r = 0.0;
#pragma unroll
    for (int k = 0; k < MAXD; ++k)
        if (k == idx) r = v[k];
    return r;
}


// Out-of-line transcendental wrappers: inlined ocml pow/log/erfc bodies make the compiler hoist dozens of f64
// literals into VGPR pairs that stay live across the whole Brent loop (measured: 233 VGPRs, 2 waves/SIMD).
// A real call keeps those constants inside the callee; they are invoked a handful of times per evaluation.
__device__ __attribute__((noinline)) double jx_pow10(double x) { return pow(10.0, x); }
__device__ __attribute__((noinline)) double jx_log(double x) { return log(x); }
__device__ __attribute__((noinline)) double jx_erfc(double x) { return erfc(x); }

__device__ __forceinline__ double chi2_sf_df1_dev(double stat) {  // src/math/linalg.rs:7-17
    if (!isfinite(stat) || stat <= 0.0) return 1.0;
    double p = jx_erfc(sqrt(0.5 * stat));
    if (!isfinite(p)) return 1.0;
    if (p < 2.2250738585072014e-308) p = 2.2250738585072014e-308;
    if (p > 1.0) p = 1.0;
    return p;
}

}  // namespace jx
