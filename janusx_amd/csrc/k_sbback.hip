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
#include <string.h>

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

// ---------------------------------------------------------------------------------------------------------------------
// Register-resident form.  The block operation Cwin <- (I - V T V') Cwin is written as
//   Y = U' Cwin (U = V T', precomputed per block by sbback_vu_kernel),   Cwin -= V Y
// and the slab rows never pass through LDS: a UNIT of 16 columns is owned by three waves, each holding one 32-row chunk
// of the 96-row window in registers, in the lane layout that is at once the B operand of the first product and the
// accumulator of the second (v_mfma_f64_16x16x4_f64: B lane [k = l >> 4][n = l & 15], D lane [row = (l >> 4) + 4 r]
// [col = l & 15]; the K / M index -> window row map below makes the two coincide and gives every lane two consecutive
// rows per 16-byte access).  The window is [s0 + 64 k, s0 + 64 k + 96) (s0 = first sweep of the group); chunk i of a
// group = rows s0 + 32 i, owned by wave i mod 3 of the unit, so sliding the window by 64 rows moves no data between
// waves: the owners of the two leading chunks store them (final for this group) and take the next two.
// Per block and wave: 16 MFMAs for its partial Y (summed over the unit's three waves through LDS in a fixed order),
// 16 MFMAs for its rows of the update; only the V / U operands are read from LDS (conflict-free pitches).
constexpr int QR_BLK = QB_WIN * QB_G;      // doubles of one V (or U) image: [q][m], m contiguous and swizzled
// The images are copied to LDS verbatim (LDS-DMA: one contiguous KB per wave instruction), so the bank swizzle is part
// of the memory layout: element (q, m) of V sits at q * 32 + (m ^ (2 (q & 15))), of U at q * 32 + (m ^ (16 ((q >> 1) & 1))):
// the A-operand reads of both products (16 rows x 2 columns, resp. 2 rows x 16 columns per half wave) touch every bank once.
__host__ __device__ __forceinline__ int qr_v_at(int q, int m) { return q * QB_G + (m ^ (2 * (q & 15))); }
__host__ __device__ __forceinline__ int qr_u_at(int q, int m) { return q * QB_G + (m ^ (16 * ((q >> 1) & 1))); }

struct QrParams {
    const double *v2;       // (n, n): column s = reflectors of sweep s by matrix row
    const double *tau2;     // (n, ks)
    double *vu;             // blocks ((grp - g_lo) * ks + k): V image then U image, QR_BLK doubles each
    double *ct;             // C in slab layout: [slab][row][16 NU columns] (sbback_slab_kernel), zero-padded columns
    int n, ks, ncols;
    int g_lo, g_hi;         // groups [g_lo, g_hi) of this launch (applied from g_hi - 1 down)
    int skip;               // diagnostic bit mask (JXGPU_QB_SKIP): 1 no MFMA phases, 2 no row traffic, 4 no V / U loads,
                            // 8 no row stores, 16 no row loads
};

// one workgroup (128 threads) per (k, group): V image of the block (window rows [s0 + 64 k, + 96), zero off the
// supports), T = (striu(V'V) + diag(1 / tau))^-1 with zero rows / columns for tau = 0, U = V T'
__global__ __launch_bounds__(128) void sbback_vu_kernel(QrParams P) {
    __shared__ double vs[QB_WIN][QB_G + 1];
    __shared__ double m[QB_G][QB_G + 1];     // T^-1, then T
    __shared__ double tau_s[QB_G];
    const int k = blockIdx.x, grp = P.g_lo + blockIdx.y;
    const int t = threadIdx.x;
    const int s0 = grp * QB_G;
    const int n = P.n;
    const int wb = s0 + k * QB_SB;
    if (wb + 1 >= n) return;                 // no reflector in this block: never read
    double *out = P.vu + ((int64_t)blockIdx.y * P.ks + k) * (2 * QR_BLK);
    if (t < QB_G) {
        int r, len;
        qb_support(n, s0 + t, k, r, len);
        tau_s[t] = (len > 0) ? P.tau2[(int64_t)(s0 + t) * P.ks + k] : 0.0;
    }
    __syncthreads();
    for (int e = t; e < QB_WIN * QB_G; e += 128) {
        const int i = e / QB_WIN, q = e % QB_WIN;      // column (sweep), window row
        int r, len;
        qb_support(n, s0 + i, k, r, len);
        const int row = wb + q;
        double v = 0.0;
        if (len > 0 && tau_s[i] != 0.0 && row >= r && row < r + len) v = P.v2[(int64_t)(s0 + i) * n + row];
        vs[q][i] = v;
    }
    __syncthreads();
    for (int e = t; e < QB_G * QB_G; e += 128) {
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
    double x[QB_G];
    if (t < QB_G) {                                    // column t of the inverse by back substitution
        const int j = t;
#pragma unroll
        for (int i = QB_G - 1; i >= 0; --i) {
            double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int q = i + 1; q < QB_G; ++q)
                if (q <= j) acc -= m[i][q] * x[q];
            x[i] = (i <= j) ? acc / m[i][i] : 0.0;
        }
    }
    __syncthreads();
    if (t < QB_G) {
#pragma unroll
        for (int i = 0; i < QB_G; ++i) m[i][t] = (tau_s[i] != 0.0 && tau_s[t] != 0.0) ? x[i] : 0.0;
    }
    __syncthreads();
    for (int e = t; e < QR_BLK; e += 128) {
        const int q = e / QB_G, mp = e % QB_G;
        double acc = 0.0;
#pragma unroll 8
        for (int i = 0; i < QB_G; ++i) acc = fma(vs[q][i], m[mp][i], acc);     // U[q][m'] = sum_m V[q][m] T[m'][m]
        out[qr_v_at(q, mp)] = vs[q][mp];
        out[QR_BLK + qr_u_at(q, mp)] = acc;
    }
}

typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));     // two consecutive rows of a column (any row parity)

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, which would make
// every barrier wait for the prefetches and the stores in flight
__device__ __forceinline__ void qr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NU units (16 columns each) per workgroup: three compute waves per unit plus ONE loader wave that copies the V / U images
// of the NEXT block into the other half of a double buffer by LDS-DMA (no registers, its own memory counter: the compute
// waves never wait for an image and every copy has a whole block of flight time).
// LDS: V images [2][96][32] | U images [2][96][32] | partial Y [3 NU][8][64].  Two barriers per block:
//   T(k)  loader: issue the copies of V(k+1), U(k+1)         compute: take over the prefetched chunk, store the finished
//                                                             one, prefetch the next; partial Y from U(k)
//   B2(k)                                                     compute: sum the partials, rows -= V(k) Y
//   B3(k) loader: the copies have landed (vmcnt(0)) before it arrives
template <int NU>
__global__ __launch_bounds__(NU * 192 + 64) void sbback_apply_reg_kernel(QrParams P) {
    extern __shared__ __attribute__((aligned(16))) double qb_smem[];
    double *vl = qb_smem;                                      // [2][QR_BLK]
    double *ul = vl + 2 * QR_BLK;                              // [2][QR_BLK]
    double *part = ul + 2 * QR_BLK;                            // [3 NU][8][64]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = P.n;

    if (wave == 3 * NU) {
        // ------------------------------------------------------------------------------------------------ loader wave
        // images of block (grp, k) -> buffer `buf`: 2 x 24 wave instructions of 1 KB
        auto img_copy = [&](int grp, int k, int buf) {
            const char *src = reinterpret_cast<const char *>(P.vu + ((int64_t)(grp - P.g_lo) * P.ks + k) * (2 * QR_BLK)) + lane * 16;
            const unsigned v_dst = (unsigned)(uintptr_t)(vl + buf * QR_BLK), u_dst = (unsigned)(uintptr_t)(ul + buf * QR_BLK);
#pragma unroll
            for (int i = 0; i < QR_BLK * 8 / 1024; ++i) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src + i * 1024), "s"(__builtin_amdgcn_readfirstlane(v_dst + i * 1024))
                             : "memory");
            }
#pragma unroll
            for (int i = 0; i < QR_BLK * 8 / 1024; ++i) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src + QR_BLK * 8 + i * 1024), "s"(__builtin_amdgcn_readfirstlane(u_dst + i * 1024))
                             : "memory");
            }
        };
        for (int grp = P.g_hi - 1; grp >= P.g_lo; --grp) {
            const int s0 = grp * QB_G;
            if (s0 + 1 >= n) continue;
            const int nk = (n - s0 - 1 + QB_SB - 1) / QB_SB;
            __syncthreads();                                   // G0
            img_copy(grp, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            qr_lds_barrier();                                  // G1
            for (int k = 0; k < nk; ++k) {
                if (k + 1 < nk && !(P.skip & 4)) img_copy(grp, k + 1, (k + 1) & 1);
                qr_lds_barrier();                              // B2
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                qr_lds_barrier();                              // B3
            }
        }
        return;
    }

    // ------------------------------------------------------------------------------------------------------ compute waves
    const int lx = lane & 15, lk = lane >> 4;
    const int unit = wave / 3, j = wave % 3;
    // slab layout: the rows of this workgroup's 16 NU columns are contiguous (one row = 128 NU bytes), so a workgroup streams
    // through memory linearly (one read and one write stream per workgroup instead of one per column)
    constexpr int W = NU * 16;
    double *cp = P.ct + (int64_t)blockIdx.x * n * W + unit * 16 + lx;
    const int rm_a = 8 * (lx >> 3) + 2 * (lx & 3) + ((lx >> 2) & 1);      // A-operand row lx of a 16-row block -> window row

    // chunk rows rb + [0, 32): register (h, r) <-> row rb + 16 h + 8 (r >> 1) + 2 lk + (r & 1).  The load is branch-free
    // (always a valid row) and raw: rows past the end are zeroed only when the chunk is taken over, so that the loads stay
    // in flight behind the block's arithmetic.  One instruction = 4 rows x 128 bytes.
    // The loads are asm statements hipcc does not count: its own wait for a counted load would also wait for the YOUNGER
    // stores of the finished chunk (the memory counter retires in issue order and hipcc assumes the fewest operations
    // behind a load), and a store acknowledgement under write-back pressure takes longer than a block.  chunk_wait is the
    // only wait for them: vmcnt(8) when exactly eight stores were issued behind the loads, else vmcnt(0).
    auto chunk_load = [&](int rb, double (&raw)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            const double *src = cp + (int64_t)min(row, n - 1) * W;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(raw[i]) : "v"(src) : "memory");
        }
    };
    auto chunk_wait = [&](bool eight_stores_behind, double (&raw)[8]) {
        if (eight_stores_behind)
            asm volatile("s_waitcnt vmcnt(8)"
                         : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7])
                         :
                         : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)"
                         : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7])
                         :
                         : "memory");
    };
    auto chunk_unpack = [&](int rb, const double (&raw)[8], d4 (&reg)[2]) {
        const bool inside = rb + 32 <= n;                      // whole chunk inside the matrix (wave-uniform)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            reg[i >> 2][i & 3] = (inside || row < n) ? raw[i] : 0.0;
        }
    };
    auto chunk_store = [&](int rb, const d4 (&reg)[2]) {
        if (rb + 32 <= n) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                double *dst = cp + (int64_t)(rb + 8 * (i >> 1) + 2 * lk + (i & 1)) * W;
                *dst = reg[i >> 2][i & 3];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            if (row < n) cp[(int64_t)row * W] = reg[i >> 2][i & 3];
        }
    };

    d4 cw[2];
    double pf[8];
    for (int grp = P.g_hi - 1; grp >= P.g_lo; --grp) {
        const int s0 = grp * QB_G;
        if (s0 + 1 >= n) continue;
        const int nk = (n - s0 - 1 + QB_SB - 1) / QB_SB;       // steps k with a reflector: s0 + 1 + 64 k < n
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // G0: the previous group's stores are done, LDS is free
        chunk_load(s0 + 32 * j, pf);
        chunk_wait(false, pf);
        chunk_unpack(s0 + 32 * j, pf, cw);
        bool stored8 = false;                                  // exactly eight stores were issued behind the last prefetch
        qr_lds_barrier();                                      // G1: the images of block 0 are in place
        int w = j;                                             // position of this wave's chunk in the window: (j + k) mod 3
        for (int k = 0; k < nk; ++k) {
            const int wb = s0 + k * QB_SB;
            const bool has_next = k + 1 < nk;
            // the chunk that left the window in the previous block is final for this group: store it, take over the
            // prefetched one (the wait for the prefetch comes before the stores are issued: the counter retires in order)
            // the chunk that left the window in the previous block is final for this group: take over the prefetched one,
            // prefetch the next, then store the finished one (behind the loads: nothing waits for a store but G0)
            const bool take = k > 0 && w != 0 && !(P.skip & 2);
            d4 done[2] = {cw[0], cw[1]};
            if (take) {
                chunk_wait(stored8, pf);
                chunk_unpack(wb + 32 * w, pf, cw);
            }
            if (has_next && w != 2 && !(P.skip & 18)) chunk_load(wb + QB_WIN + 32 * w, pf);
            stored8 = false;
            if (take && !(P.skip & 8)) {
                chunk_store(wb - QB_WIN + 32 * w, done);
                stored8 = wb - QB_WIN + 32 * w + 32 <= n;
            }
            d4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
            if (!(P.skip & 1)) {
                // U(q, 16 mb + lx), q = 32 w + 8 (ks >> 1) + 2 lk + (ks & 1): the swizzle bit of the row is lk & 1
                const double *up = ul + (k & 1) * QR_BLK + (32 * w + 2 * lk) * QB_G + lx;
                const int o0 = 16 * (lk & 1), o1 = 16 - o0;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int q = 8 * (ks >> 1) + (ks & 1);
                    const double b = cw[ks >> 2][ks & 3];
                    y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o0], b, y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o1], b, y1, 0, 0, 0);
                }
            }
            {
                double *pp = part + (wave * 8) * 64 + lane;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pp[r * 64] = y0[r];
                    pp[(4 + r) * 64] = y1[r];
                }
            }
            qr_lds_barrier();                                  // B2: the unit's three partial sums are in LDS
            double yn[8];
            {
                const double *pp = part + (unit * 3 * 8) * 64 + lane;
#pragma unroll
                for (int i = 0; i < 8; ++i) yn[i] = -((pp[i * 64] + pp[(8 + i) * 64]) + pp[(16 + i) * 64]);
            }
            if (!(P.skip & 1)) {
                // V(32 w + 16 h + rm_a, 4 ks + lk): the swizzle of the row is 2 rm_a
                const double *vp = vl + (k & 1) * QR_BLK + (32 * w + rm_a) * QB_G;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int m = (4 * ks + lk) ^ (2 * rm_a);
                    cw[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[m], yn[ks], cw[0], 0, 0, 0);
                    cw[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[16 * QB_G + m], yn[ks], cw[1], 0, 0, 0);
                }
            }
            qr_lds_barrier();                                  // B3: the partial sums have been read, the images of block k + 1 are in place
            if (!has_next && !(P.skip & 2)) chunk_store(wb + 32 * w, cw);
            w = (w == 2) ? 0 : w + 1;
        }
    }
}

// C (n x ncols, column-major, ld = n) <-> slab layout [slab][row][w] (w = slab width in columns; columns past ncols are
// zero on the way in and dropped on the way out).  One workgroup = 64 rows of one slab through LDS: both sides coalesced.
template <bool TO_SLAB>
__global__ __launch_bounds__(256) void sbback_slab_kernel(double *__restrict__ c, double *__restrict__ ct, int n, int ncols, int w) {
    extern __shared__ double slab_tile[];                      // [64][w + 1]
    const int r0 = blockIdx.x * 64, slab = blockIdx.y;
    const int t = threadIdx.x;
    const int64_t base = (int64_t)slab * n * w;
    if (TO_SLAB) {
        for (int e = t; e < 64 * w; e += 256) {
            const int cc = e >> 6, r = e & 63;
            const int col = slab * w + cc, row = r0 + r;
            slab_tile[r * (w + 1) + cc] = (col < ncols && row < n) ? c[(int64_t)col * n + row] : 0.0;
        }
        __syncthreads();
        for (int e = t; e < 64 * w; e += 256) {
            const int r = e / w, cc = e % w;
            if (r0 + r < n) ct[base + (int64_t)(r0 + r) * w + cc] = slab_tile[r * (w + 1) + cc];
        }
    } else {
        for (int e = t; e < 64 * w; e += 256) {
            const int r = e / w, cc = e % w;
            slab_tile[r * (w + 1) + cc] = (r0 + r < n) ? ct[base + (int64_t)(r0 + r) * w + cc] : 0.0;
        }
        __syncthreads();
        for (int e = t; e < 64 * w; e += 256) {
            const int cc = e >> 6, r = e & 63;
            const int col = slab * w + cc, row = r0 + r;
            if (col < ncols && row < n) c[(int64_t)col * n + row] = slab_tile[r * (w + 1) + cc];
        }
    }
}

static bool qb_lds_form() {
    static const bool v = getenv("JXGPU_SBBACK") && strcmp(getenv("JXGPU_SBBACK"), "lds") == 0;
    return v;
}

// groups per launch of the register form: the V / U images of a launch stay below ~12 GB (one launch up to n ~ 22000)
static int qr_groups_per_launch(int n, int ks) {
    const int ngroups = (n - 2 + QB_G - 1) / QB_G;
    const double per_group = (double)ks * 2 * QR_BLK * sizeof(double);
    int g = (int)(12.0e9 / per_group);
    if (getenv("JXGPU_SBBACK_GROUPS") && atoi(getenv("JXGPU_SBBACK_GROUPS")) > 0) g = atoi(getenv("JXGPU_SBBACK_GROUPS"));
    if (g < 1) g = 1;
    return g < ngroups ? g : ngroups;
}

size_t sbback_tq_doubles(int n, int ks) {
    if (qb_lds_form()) return (size_t)((n - 2 + QB_G - 1) / QB_G + 1) * ks * QB_G * QB_G;
    // V / U images of one launch | C in slab layout (up to 5 x 16 - 1 padding columns)
    return (size_t)qr_groups_per_launch(n, ks) * ks * 2 * QR_BLK + (size_t)(n + 96) * n;
}

static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

// register form: launches of <= qr_groups_per_launch groups, from the last group down; ev_start / ev_stop bracket the
// apply kernels (first / last launch)
static int sbback_apply_q2_reg(hipStream_t st, const double *d_v2, const double *d_tau2, int n, int ks, double *d_c, int ncols,
                               double *d_vu, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int ngroups = (n - 2 + QB_G - 1) / QB_G;
    const int gpl = qr_groups_per_launch(n, ks);
    const int units = ceil_div(ncols, 16);
    int nu = ceil_div(units, device_cus());
    if (nu < 1) nu = 1;
    if (nu > 5) nu = 5;
    if (getenv("JXGPU_SBBACK_NW") && atoi(getenv("JXGPU_SBBACK_NW")) > 0) nu = atoi(getenv("JXGPU_SBBACK_NW"));
    if (nu > 5) nu = 5;
    const size_t lds = sizeof(double) * (4 * (size_t)QR_BLK + (size_t)nu * 3 * 8 * 64);
    const dim3 grid(ceil_div(units, nu));
    const int skip = getenv("JXGPU_QB_SKIP") ? atoi(getenv("JXGPU_QB_SKIP")) : 0;
    const int w = nu * 16;
    double *d_ct = d_vu + (size_t)gpl * ks * 2 * QR_BLK;
    const dim3 sgrid(ceil_div(n, 64), grid.x);
    const size_t slds = sizeof(double) * 64 * (w + 1);
    hipLaunchKernelGGL(sbback_slab_kernel<true>, sgrid, dim3(256), slds, st, d_c, d_ct, n, ncols, w);
    JX_LAUNCH_CHECK();
    for (int g_hi = ngroups; g_hi > 0; g_hi -= gpl) {
        const int g_lo = g_hi > gpl ? g_hi - gpl : 0;
        QrParams P{d_v2, d_tau2, d_vu, d_ct, n, ks, ncols, g_lo, g_hi, skip};
        hipLaunchKernelGGL(sbback_vu_kernel, dim3(ks, g_hi - g_lo), dim3(128), 0, st, P);
        JX_LAUNCH_CHECK();
        hipEvent_t e0 = (g_hi == ngroups) ? ev_start : nullptr, e1 = (g_lo == 0) ? ev_stop : nullptr;
#define JX_QR_LAUNCH(NUV)                                                                                              \
    do {                                                                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            JX_HIP(hipFuncSetAttribute((const void *)sbback_apply_reg_kernel<NUV>,                                     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipExtLaunchKernelGGL(sbback_apply_reg_kernel<NUV>, grid, dim3(NUV * 192 + 64), lds, st, e0, e1, 0, P);             \
    } while (0)
        switch (nu) {
            case 1: JX_QR_LAUNCH(1); break;
            case 2: JX_QR_LAUNCH(2); break;
            case 3: JX_QR_LAUNCH(3); break;
            case 4: JX_QR_LAUNCH(4); break;
            default: JX_QR_LAUNCH(5); break;
        }
#undef JX_QR_LAUNCH
        JX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sbback_slab_kernel<false>, sgrid, dim3(256), slds, st, d_c, d_ct, n, ncols, w);
    JX_LAUNCH_CHECK();
    return 0;
}

// C (n x ncols, ld = n) <- Q2 C.  d_tq: sbback_tq_doubles(n, ks) doubles of workspace.
int sbback_apply_q2(hipStream_t st, const double *d_v2, const double *d_tau2, int n, int ks, double *d_c, int ncols,
                    double *d_tq, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (n <= 2 || ncols <= 0) return 0;
    if (!qb_lds_form()) return sbback_apply_q2_reg(st, d_v2, d_tau2, n, ks, d_c, ncols, d_tq, ev_start, ev_stop);
    const int ngroups = (n - 2 + QB_G - 1) / QB_G;
    QbParams P{d_v2, d_tau2, d_tq, d_c, n, ks, ngroups, ncols, getenv("JXGPU_QB_SKIP") ? atoi(getenv("JXGPU_QB_SKIP")) : 0};
    hipLaunchKernelGGL(sbback_tfactor_kernel, dim3(ks, ngroups), dim3(64), 0, st, P);
    JX_LAUNCH_CHECK();
    // slab width: the widest (<= 80 columns: LDS) that still gives every CU a slab
    const int cus = device_cus();
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
