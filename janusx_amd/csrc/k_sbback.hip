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
#include <hip/hip_ext.h>
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
    int skip;      // diagnostic bit mask (JXGPU_QB_SKIP): 1 no MFMA phases, 2 no row traffic, 4 no V loads
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

constexpr int QB_VP = QB_WIN + 2;          // pitch of the V image [m][q]
constexpr int QB_T = 1024;                 // threads per workgroup: two waves per SIMD hide each other's LDS latency
constexpr int QB_NWAVE = QB_T / 64;

// NW = slab width / 16.  LDS: ring [W][QB_RP] | w1, w2 [W][QB_WP] | vs [QB_G][QB_VP]
template <int NW>
__global__ __launch_bounds__(QB_T) void sbback_apply_kernel(QbParams P) {
    constexpr int W = NW * 16;                    // slab width
    constexpr int NROW = W * QB_SB / QB_T;         // prefetch registers: one 64-row chunk of the slab
    constexpr int NV = QB_G * QB_WIN / QB_T;       // prefetch registers: one V block
    extern __shared__ __attribute__((aligned(16))) double qb_smem[];
    double *ring = qb_smem;                       // [W][QB_RP]: ring[c][row & (RING-1)]
    double *w1 = ring + W * QB_RP;                // [W][QB_WP]
    double *w2 = w1 + W * QB_WP;                  // [W][QB_WP]
    double *vs = w2 + W * QB_WP;                  // [QB_G][QB_VP]: vs[m][q] = V[rlo + q][sweep s0 + m], zero off its support
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lx = lane & 15, lk = lane >> 4;
    const int n = P.n;
    const int c0 = blockIdx.x * W;
    const int ncol = min(W, P.ncols - c0);
    double *cg = P.c + (int64_t)c0 * n;

    // 64-row chunk [ra, ra + 64) of the slab, rows >= rb masked: thread element (c = e / 64, q = e % 64)
    auto chunk_load = [&](int ra, int rb, double (&reg)[NROW]) {
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            const int e = i * QB_T + t;
            const int c = e >> 6, q = ra + (e & 63);
            reg[i] = (c < ncol && q < rb) ? cg[(int64_t)c * n + q] : 0.0;
        }
    };
    auto chunk_to_ring = [&](int ra, const double (&reg)[NROW]) {
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            const int e = i * QB_T + t;
            ring[(e >> 6) * QB_RP + ((ra + (e & 63)) & (QB_RING - 1))] = reg[i];
        }
    };
    auto chunk_store = [&](int ra, int rb) {      // ring rows [ra, min(ra + 64, rb)) -> memory
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            const int e = i * QB_T + t;
            const int c = e >> 6, q = ra + (e & 63);
            if (c < ncol && q < rb) cg[(int64_t)c * n + q] = ring[c * QB_RP + (q & (QB_RING - 1))];
        }
    };
    auto v_load = [&](int s0, int rlo, double (&reg)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = i * QB_T + t;
            const int m = e / QB_WIN, q = e % QB_WIN;
            const int r = rlo + m;                          // first row of the reflector of sweep s0 + m in this step
            const int len = (s0 + m < n - 2 && r < n) ? min(QB_SB, n - r) : 0;
            reg[i] = (q >= m && q < m + len) ? P.v2[(int64_t)(s0 + m) * n + rlo + q] : 0.0;
        }
    };
    auto v_to_lds = [&](const double (&reg)[NV]) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = i * QB_T + t;
            vs[(e / QB_WIN) * QB_VP + e % QB_WIN] = reg[i];
        }
    };

    double rows_pf[NROW], v_pf[NV];
    for (int grp = P.ngroups - 1; grp >= 0; --grp) {
        const int s0 = grp * QB_G;
        const int rlo0 = s0 + 1;
        if (rlo0 >= n) continue;
        __syncthreads();                               // the previous group's last stores have read the ring
        // prologue: the first window (96 rows) straight into the ring, V of step 0 into vs
        {
            const int rhi0 = min(n, rlo0 + QB_WIN);
            chunk_load(rlo0, rhi0, rows_pf);
            chunk_to_ring(rlo0, rows_pf);
            chunk_load(rlo0 + QB_SB, rhi0, rows_pf);
#pragma unroll
            for (int i = 0; i < NROW; ++i) {           // rows rlo0 + 64 .. rlo0 + 95 only (the ring must keep the first chunk)
                const int e = i * QB_T + t;
                if ((e & 63) < QB_WIN - QB_SB)
                    ring[(e >> 6) * QB_RP + ((rlo0 + QB_SB + (e & 63)) & (QB_RING - 1))] = rows_pf[i];
            }
            v_load(s0, rlo0, v_pf);
            v_to_lds(v_pf);
        }
        for (int k = 0;; ++k) {
            const int rlo = rlo0 + k * QB_SB;
            const int rhi = min(n, rlo + QB_WIN);
            const int rlo_next = rlo + QB_SB;
            const bool has_next = rlo_next < n;
            const int rhi_next = min(n, rlo_next + QB_WIN);
            __syncthreads();                           // ring rows and vs of this block are in place
            // (the next block's V and the rows its window gains, [rlo + 96, rlo + 160), are prefetched below)
            // T fragments of this wave's W2 blocks (every block of a wave has the same mb = wave % 2): in flight during W1
            double tv[QB_G / 4];
            {
                const double *tq = P.tq + ((int64_t)grp * P.ks + k) * (QB_G * QB_G);
                const int mb = wave % (QB_G / 16);
#pragma unroll
                for (int ks = 0; ks < QB_G / 4; ++ks) tv[ks] = tq[(mb * 16 + lx) + (4 * ks + lk) * QB_G];
            }
            // then the prefetch for the next block (memory operations retire in issue order: what is needed first goes first)
            if (has_next) {
                if (!(P.skip & 4)) v_load(s0, rlo_next, v_pf);
                if (!(P.skip & 2)) chunk_load(rlo + QB_WIN, rhi_next, rows_pf);
            }
            // ---- W1[c][m] = sum_q Cwin[q][c] V[q][m]
            if (!(P.skip & 1))
            for (int blk = wave; blk < NW * (QB_G / 16); blk += QB_NWAVE) {
                const int cb = blk / (QB_G / 16), mb = blk % (QB_G / 16);
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                const double *ap = ring + (cb * 16 + lx) * QB_RP;
                const double *bp = vs + (mb * 16 + lx) * QB_VP + lk;
#pragma unroll
                for (int ks = 0; ks < QB_WIN / 4; ++ks) {
                    const double av = ap[(rlo + 4 * ks + lk) & (QB_RING - 1)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[4 * ks], acc, 0, 0, 0);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) w1[(cb * 16 + lk + 4 * rr) * QB_WP + mb * 16 + lx] = acc[rr];
            }
            __syncthreads();
            // ---- W2[c][m'] = sum_m W1[c][m] T[m'][m]
            if (!(P.skip & 1))
            for (int blk = wave; blk < NW * (QB_G / 16); blk += QB_NWAVE) {
                const int cb = blk / (QB_G / 16), mb = blk % (QB_G / 16);
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < QB_G / 4; ++ks) {
                    const double av = w1[(cb * 16 + lx) * QB_WP + 4 * ks + lk];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, tv[ks], acc, 0, 0, 0);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) w2[(cb * 16 + lk + 4 * rr) * QB_WP + mb * 16 + lx] = acc[rr];
            }
            __syncthreads();
            // ---- Cwin[q][c] -= sum_m V[q][m] W2[c][m]
            if (!(P.skip & 1))
            for (int blk = wave; blk < NW * (QB_WIN / 16); blk += QB_NWAVE) {
                const int cb = blk / (QB_WIN / 16), qb = blk % (QB_WIN / 16);
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                const double *ap = w2 + (cb * 16 + lx) * QB_WP + lk;
                const double *bp = vs + lk * QB_VP + qb * 16 + lx;
#pragma unroll
                for (int ks = 0; ks < QB_G / 4; ++ks)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * ks], bp[4 * ks * QB_VP], acc, 0, 0, 0);
                const int slot = (rlo + qb * 16 + lx) & (QB_RING - 1);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) ring[(cb * 16 + lk + 4 * rr) * QB_RP + slot] -= acc[rr];
            }
            __syncthreads();
            // rows that leave the window are final for this group: [rlo, rlo + 64), or everything after the last block
            if (!(P.skip & 2)) chunk_store(rlo, rhi);
            if (!has_next) {
                chunk_store(rlo + QB_SB, rhi);
                break;
            }
            __syncthreads();                           // ring reads done: the freed slots take the prefetched rows
            chunk_to_ring(rlo + QB_WIN, rows_pf);
            v_to_lds(v_pf);
        }
    }
}

size_t sbback_tq_doubles(int n, int ks) { return (size_t)((n - 2 + QB_G - 1) / QB_G + 1) * ks * QB_G * QB_G; }

// C (n x ncols, ld = n) <- Q2 C.  d_tq: sbback_tq_doubles(n, ks) doubles of workspace.
int sbback_apply_q2(hipStream_t st, const double *d_v2, const double *d_tau2, int n, int ks, double *d_c, int ncols,
                    double *d_tq, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (n <= 2 || ncols <= 0) return 0;
    const int ngroups = (n - 2 + QB_G - 1) / QB_G;
    QbParams P{d_v2, d_tau2, d_tq, d_c, n, ks, ngroups, ncols, getenv("JXGPU_QB_SKIP") ? atoi(getenv("JXGPU_QB_SKIP")) : 0};
    hipLaunchKernelGGL(sbback_tfactor_kernel, dim3(ks, ngroups), dim3(64), 0, st, P);
    JX_LAUNCH_CHECK();
    // slab width: the widest (<= 80 columns: LDS) that still gives every CU a slab
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        JX_HIP(hipGetDevice(&dev));
        JX_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    int nw = ceil_div(ceil_div(ncols, 16), cus);
    if (nw < 1) nw = 1;
    if (nw > 5) nw = 5;
    if (getenv("JXGPU_SBBACK_NW") && atoi(getenv("JXGPU_SBBACK_NW")) > 0) nw = atoi(getenv("JXGPU_SBBACK_NW"));
    if (nw > 5) nw = 5;
    const int w = nw * 16;
    const size_t lds = sizeof(double) * ((size_t)w * QB_RP + 2 * (size_t)w * QB_WP + (size_t)QB_G * QB_VP);
    const dim3 grid(ceil_div(ncols, w));
#define JX_QB_LAUNCH(NWV)                                                                                              \
    do {                                                                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            JX_HIP(hipFuncSetAttribute((const void *)sbback_apply_kernel<NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds));                                                                     \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipExtLaunchKernelGGL(sbback_apply_kernel<NWV>, grid, dim3(QB_T), lds, st, ev_start, ev_stop, 0, P);            \
    } while (0)
    switch (nw) {
        case 1: JX_QB_LAUNCH(1); break;
        case 2: JX_QB_LAUNCH(2); break;
        case 3: JX_QB_LAUNCH(3); break;
        case 4: JX_QB_LAUNCH(4); break;
        default: JX_QB_LAUNCH(5); break;
    }
#undef JX_QB_LAUNCH
    JX_LAUNCH_CHECK();
    return 0;
}

}  // namespace jx
