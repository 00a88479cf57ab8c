// Back-transformation of the bulge-chasing stage (k_sb2st.hip): C <- Q2 C, Q2 = product of the reflectors H(s,k) in
// generation order, C (n x n, column-major) = eigenvectors of the tridiagonal matrix.  Last stage but one behind
// src/math/eigh.rs:1422-1528 (the reference's LAPACK dsyevd does the equivalent inside dormtr).
//
// Reflectors are applied in blocks (group of G2 consecutive sweeps) x (step k): the G2 reflectors of one step form a
// parallelogram V (SB + G2 - 1 rows, column i shifted down by i) and one compact-WY factor I - V T V'.  Order (proved
// in scripts/proto_twostage.py): groups descending, inside a group the steps k = 0, 1, ... ascending.  A workgroup owns a
// slab of NW*16 columns of C and walks the whole block sequence; the rows of its slab live in an LDS ring while the
// window slides down by SB rows per step (every row of the slab is read and written once per group).  Per block:
//   W1 = V' Cwin,  W2 = T W1,  Cwin -= V W2      (all on v_mfma_f64_16x16x4_f64; V and T fragments straight from L2)
// T comes from sbback_tfactor_kernel: T^-1 = striu(V'V) + diag(1 / tau).
#include <stdlib.h>

#include "jx_common.h"

namespace jx {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int QB_SB = 64;                  // band width (= SB of k_sy2sb.hip / BC_SB of k_sb2st.hip)
constexpr int QB_G = 32;                   // sweeps per group
constexpr int QB_WIN = QB_SB + QB_G;       // window rows (SB + G2 - 1 rounded up to a multiple of 16)
constexpr int QB_RING = 128;               // LDS ring rows (>= QB_WIN, power of two)
constexpr int QB_RP = QB_RING + 2;         // ring pitch (doubles): 16 columns spread over all banks
constexpr int QB_WP = QB_G + 2;            // pitch of the W buffers

struct QbParams {
    const double *v2;       // (n, n): column s = reflectors of sweep s by matrix row
    const double *tau2;     // (n, ks)
    double *tq;             // (ngroups * ks) blocks of QB_G x QB_G, column-major: T of block (group, k)
    double *c;              // (n, ncols) column-major, ld = n
    int n, ks, ngroups, ncols;
};

// support of the reflector of sweep s, step k: rows [r, r + len)
__device__ __forceinline__ void qb_support(int n, int s, int k, int &r, int &len) {
    r = s + 1 + k * QB_SB;
    len = (s < n - 2 && r < n) ? min(QB_SB, n - r) : 0;
}

// one workgroup (64 threads) per (group, k): T = (striu(V'V) + diag(1/tau))^-1, zero rows / columns for tau = 0
__global__ __launch_bounds__(64) void sbback_tfactor_kernel(QbParams P) {
    __shared__ double vs[QB_WIN][QB_G + 1];
    __shared__ double m[QB_G][QB_G + 1];     // T^-1, then T
    __shared__ double tau_s[QB_G];
    const int k = blockIdx.x, grp = blockIdx.y;
    const int t = threadIdx.x;
    const int s0 = grp * QB_G;
    const int n = P.n;
    double *out = P.tq + ((int64_t)grp * P.ks + k) * (QB_G * QB_G);
    const int rlo = s0 + 1 + k * QB_SB;
    if (rlo >= n) {                     // no reflector in this block
        for (int e = t; e < QB_G * QB_G; e += 64) out[e] = 0.0;
        return;
    }
    if (t < QB_G) {
        int r, len;
        qb_support(n, s0 + t, k, r, len);
        tau_s[t] = (len > 0) ? P.tau2[(int64_t)(s0 + t) * P.ks + k] : 0.0;
    }
    __syncthreads();
    for (int e = t; e < QB_WIN * QB_G; e += 64) {
        const int i = e / QB_WIN, q = e % QB_WIN;      // column (sweep), window row
        int r, len;
        qb_support(n, s0 + i, k, r, len);
        const int row = rlo + q;
        double v = 0.0;
        if (len > 0 && tau_s[i] != 0.0 && row >= r && row < r + len) v = P.v2[(int64_t)(s0 + i) * n + row];
        vs[q][i] = v;
    }
    __syncthreads();
    for (int e = t; e < QB_G * QB_G; e += 64) {
        const int i = e / QB_G, j = e % QB_G;          // m[i][j], upper: i < j
        double acc = 0.0;
        if (i < j) {
            for (int q = 0; q < QB_WIN; ++q) acc += vs[q][i] * vs[q][j];
        } else if (i == j) {
            acc = (tau_s[i] != 0.0) ? 1.0 / tau_s[i] : 1.0;
        }
        m[i][j] = acc;
    }
    __syncthreads();
    // in-place inverse of the upper triangular m: column j by back substitution (thread = column)
    if (t < QB_G) {
        const int j = t;
        double x[QB_G];
#pragma unroll
        for (int i = QB_G - 1; i >= 0; --i) {
            double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int q = i + 1; q < QB_G; ++q)
                if (q <= j) acc -= m[i][q] * x[q];
            x[i] = (i <= j) ? acc / m[i][i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < QB_G; ++i) out[i + j * QB_G] = (tau_s[i] != 0.0 && tau_s[j] != 0.0) ? x[i] : 0.0;
    }
}

template <int NW>
__global__ __launch_bounds__(256) void sbback_apply_kernel(QbParams P) {
    constexpr int W = NW * 16;                    // slab width
    extern __shared__ __attribute__((aligned(16))) double qb_smem[];
    double *ring = qb_smem;                       // [W][QB_RP]: ring[c][row & (RING-1)]
    double *w1 = ring + W * QB_RP;                // [W][QB_WP]
    double *w2 = w1 + W * QB_WP;                  // [W][QB_WP]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lx = lane & 15, lk = lane >> 4;
    const int n = P.n;
    const int c0 = blockIdx.x * W;
    const int ncol = min(W, P.ncols - c0);
    double *cg = P.c + (int64_t)c0 * n;

    auto load_rows = [&](int ra, int rb) {        // C rows [ra, rb) of the slab -> ring
        const int nr = rb - ra;
        if (nr <= 0) return;
        for (int e = t; e < nr * W; e += 256) {
            const int c = e / nr, q = ra + e % nr;
            ring[c * QB_RP + (q & (QB_RING - 1))] = (c < ncol) ? cg[(int64_t)c * n + q] : 0.0;
        }
    };
    auto store_rows = [&](int ra, int rb) {
        const int nr = rb - ra;
        if (nr <= 0) return;
        for (int e = t; e < nr * W; e += 256) {
            const int c = e / nr, q = ra + e % nr;
            if (c < ncol) cg[(int64_t)c * n + q] = ring[c * QB_RP + (q & (QB_RING - 1))];
        }
    };

    for (int grp = P.ngroups - 1; grp >= 0; --grp) {
        const int s0 = grp * QB_G;
        int have_lo = 0, have_hi = 0;             // rows of C currently in the ring: [have_lo, have_hi)
        for (int k = 0;; ++k) {
            const int rlo = s0 + 1 + k * QB_SB;
            if (rlo >= n) break;
            const int rhi = min(n, rlo + QB_WIN);
            // ring maintenance: write back the rows above the window, read the rows the window gained
            if (k == 0) {
                have_lo = rlo;
                have_hi = rlo;
            }
            __syncthreads();                       // previous block's updates are complete
            store_rows(have_lo, min(have_hi, rlo));
            if (have_lo < rlo) have_lo = min(have_hi, rlo);
            if (have_hi < rlo) {
                have_lo = rlo;
                have_hi = rlo;
            }
            __syncthreads();
            load_rows(have_hi, rhi);
            have_hi = rhi;
            __syncthreads();
            const double *tq = P.tq + ((int64_t)grp * P.ks + k) * (QB_G * QB_G);
            // ---- W1[c][m] = sum_row Cwin[row][c] V[row][m]: (NW x G/16) blocks of 16 x 16, K = window rows
            for (int blk = wave; blk < NW * (QB_G / 16); blk += 4) {
                const int cb = blk / (QB_G / 16), mb = blk % (QB_G / 16);
                const int mcol = mb * 16 + lx;                  // reflector (B operand column)
                int r, len;
                qb_support(n, s0 + mcol, k, r, len);
                const double tauv = (len > 0) ? P.tau2[(int64_t)(s0 + mcol) * P.ks + k] : 0.0;
                if (tauv == 0.0) len = 0;
                const double *vcol = P.v2 + (int64_t)(s0 + mcol) * n;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                double bv[QB_WIN / 4];
#pragma unroll
                for (int ks = 0; ks < QB_WIN / 4; ++ks) {
                    const int row = rlo + 4 * ks + lk;
                    bv[ks] = (row >= r && row < r + len) ? vcol[row] : 0.0;
                }
#pragma unroll
                for (int ks = 0; ks < QB_WIN / 4; ++ks) {
                    const int row = rlo + 4 * ks + lk;
                    const double av = (row < rhi) ? ring[(cb * 16 + lx) * QB_RP + (row & (QB_RING - 1))] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[ks], acc, 0, 0, 0);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) w1[(cb * 16 + lk + 4 * rr) * QB_WP + mb * 16 + lx] = acc[rr];
            }
            __syncthreads();
            // ---- W2[c][m'] = sum_m W1[c][m] T[m'][m]
            for (int blk = wave; blk < NW * (QB_G / 16); blk += 4) {
                const int cb = blk / (QB_G / 16), mb = blk % (QB_G / 16);
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < QB_G / 4; ++ks) {
                    const int mm = 4 * ks + lk;
                    const double av = w1[(cb * 16 + lx) * QB_WP + mm];
                    const double bvv = tq[(mb * 16 + lx) + mm * QB_G];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bvv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) w2[(cb * 16 + lk + 4 * rr) * QB_WP + mb * 16 + lx] = acc[rr];
            }
            __syncthreads();
            // ---- Cwin[row][c] -= sum_m V[row][m] W2[c][m]: (NW x WIN/16) blocks, K = G
            for (int blk = wave; blk < NW * (QB_WIN / 16); blk += 4) {
                const int cb = blk / (QB_WIN / 16), rbk = blk % (QB_WIN / 16);
                const int row = rlo + rbk * 16 + lx;            // B operand column = window row
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                double bv[QB_G / 4];
#pragma unroll
                for (int ks = 0; ks < QB_G / 4; ++ks) {
                    const int mm = 4 * ks + lk;
                    int r, len;
                    qb_support(n, s0 + mm, k, r, len);
                    bv[ks] = (row >= r && row < r + len) ? P.v2[(int64_t)(s0 + mm) * n + row] : 0.0;
                }
#pragma unroll
                for (int ks = 0; ks < QB_G / 4; ++ks) {
                    const int mm = 4 * ks + lk;
                    const double av = w2[(cb * 16 + lx) * QB_WP + mm];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[ks], acc, 0, 0, 0);
                }
                if (row < rhi) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        ring[(cb * 16 + lk + 4 * rr) * QB_RP + (row & (QB_RING - 1))] -= acc[rr];
                }
            }
        }
        __syncthreads();
        store_rows(have_lo, have_hi);
    }
}

size_t sbback_tq_doubles(int n, int ks) { return (size_t)((n - 2 + QB_G - 1) / QB_G + 1) * ks * QB_G * QB_G; }

// C (n x ncols, ld = n) <- Q2 C.  d_tq: sbback_tq_doubles(n, ks) doubles of workspace.
int sbback_apply_q2(hipStream_t st, const double *d_v2, const double *d_tau2, int n, int ks, double *d_c, int ncols,
                    double *d_tq) {
    if (n <= 2 || ncols <= 0) return 0;
    const int ngroups = (n - 2 + QB_G - 1) / QB_G;
    QbParams P{d_v2, d_tau2, d_tq, d_c, n, ks, ngroups, ncols};
    hipLaunchKernelGGL(sbback_tfactor_kernel, dim3(ks, ngroups), dim3(64), 0, st, P);
    JX_LAUNCH_CHECK();
    // slab width: 32 columns, 16 when that leaves CUs idle
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        JX_HIP(hipGetDevice(&dev));
        JX_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    int nw = (ceil_div(ncols, 32) >= cus) ? 2 : 1;
    if (getenv("JXGPU_SBBACK_NW") && atoi(getenv("JXGPU_SBBACK_NW")) > 0) nw = atoi(getenv("JXGPU_SBBACK_NW")) >= 2 ? 2 : 1;
    const int w = nw * 16;
    const size_t lds = sizeof(double) * ((size_t)w * QB_RP + 2 * (size_t)w * QB_WP);
    if (nw == 2) hipLaunchKernelGGL(sbback_apply_kernel<2>, dim3(ceil_div(ncols, w)), dim3(256), lds, st, P);
    else hipLaunchKernelGGL(sbback_apply_kernel<1>, dim3(ceil_div(ncols, w)), dim3(256), lds, st, P);
    JX_LAUNCH_CHECK();
    return 0;
}

}  // namespace jx
