// Back-transformation of the bulge-chasing stage (k_sb2st.hip): C <- Q2 C, Q2 = product of the reflectors H(s,k) in
// generation order, C (n x n, column-major) = eigenvectors of the tridiagonal matrix.  Last stage but one behind
// src/math/eigh.rs:1422-1528 (the reference's LAPACK dsyevd does the equivalent inside dormtr).
//
// Reflectors are applied in blocks (group of QB_G consecutive sweeps) x (step k): the reflectors of one step form a
// parallelogram V (QB_SB + QB_G rows, column i shifted down by i) and one compact-WY factor I - V T V'.  Order (proved
// in scripts/proto_twostage.py): groups descending, inside a group the steps k = 0, 1, ... ascending.  Every row of C is
// read and written once per group.  (The first version of this round kept the slab rows in an LDS ring and multiplied
// by T in the kernel: 806 ms at n = 20000 against 588 ms for the register-resident form below; git history.)
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "jx_common.h"

namespace jx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// slab layout: element (row, column cc of the slab's w columns) of a slab that starts at `base`
__host__ __device__ __forceinline__ int64_t qr_slab_at(int row, int cc, int w) {
    return (int64_t)(row >> 1) * (2 * w) + 2 * cc + (row & 1);
}

constexpr int QB_SB = 64;                  // band width (= SB of k_sy2sb.hip / BC_SB of k_sb2st.hip)
constexpr int QB_G = 32;                   // sweeps per group
constexpr int QB_WIN = QB_SB + QB_G;       // window rows (SB + G2 - 1 rounded up to a multiple of 16)

// support of the reflector of sweep s, step k: rows [r, r + len)
__device__ __forceinline__ void qb_support(int n, int s, int k, int &r, int &len) {
    r = s + 1 + k * QB_SB;
    len = (s < n - 2 && r < n) ? min(QB_SB, n - r) : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Register-resident form.  The block operation Cwin <- (I - V T V') Cwin is written as
//   Y = U' Cwin (U = V T', precomputed per block by sbback_vu_kernel),   Cwin -= V Y
// and the slab rows never pass through LDS: a UNIT of 16 columns is owned by three waves, each holding one 32-row chunk
// of the 96-row window in registers, in the lane layout that is at once the B operand of the first product and the
// accumulator of the second (v_mfma_f64_16x16x4_f64: B lane [k = l >> 4][n = l & 15], D lane [row = (l >> 4) + 4 r]
// [col = l & 15]; the K / M index -> window row map below makes the two coincide).  The window is [s0 + 64 k, s0 + 64 k + 96) (s0 = first sweep of the group); chunk i of a
// group = rows s0 + 32 i, owned by wave i mod 3 of the unit, so sliding the window by 64 rows moves no data between
// waves: the owners of the two leading chunks store them (final for this group) and take the next two.
// Per block and wave: 16 MFMAs for its partial Y (summed over the unit's three waves through LDS in a fixed order),
// 16 MFMAs for its rows of the update; only the V / U operands are read from LDS (swizzled images, conflict-free).
constexpr int QR_BLK = QB_WIN * QB_G;      // doubles of one V (or U) image: [q][m], m contiguous and swizzled
// The images are copied to LDS verbatim (LDS-DMA: one contiguous KB per wave instruction), so the bank swizzle is part
// of the memory layout: element (q, m) of V sits at q * 32 + (m ^ qr_v_swz(q)), of U at q * 32 + (m ^ (16 ((q >> 1) & 1))):
// the A-operand reads of both products (16 rows x 2 columns, resp. 2 rows x 16 columns per half wave) touch every bank once.
// V: the compiler pairs the two 16-row halves' reads into ds_read2st64_b64 (16-lane groups, banks mod 32) and leaves the reads next to
// a zero tile as ds_read_b64 (32-lane groups, banks mod 64): the term q & 15 spreads the 16 rows of a lane group over the 16 bank
// pairs, the term (q & 1) << 4 keeps the two column parities of a 32-lane group apart (rounds 3 - 5 used m ^ 2 (q & 15): free of
// conflicts for ds_read_b64 only -- 4.0e9 of the kernel's 1.28e10 LDS cycles per launch were conflict cycles, SQ_LDS_BANK_CONFLICT)
__host__ __device__ __forceinline__ int qr_v_swz(int q) { return (q & 15) ^ ((q & 1) << 4); }
__host__ __device__ __forceinline__ int qr_v_at(int q, int m) { return q * QB_G + (m ^ qr_v_swz(q)); }
__host__ __device__ __forceinline__ int qr_u_at(int q, int m) { return q * QB_G + (m ^ (16 * ((q >> 1) & 1))); }

// Zero tiles of the images (a reflector of sweep s0 + i covers the window rows i + 1 .. i + 64; column j of U = V T' starts at
// row j + 1 because T is upper triangular): 12 of the 96 MFMA operand tiles of a block are zero and their instructions are
// skipped -- V: rows 0 .. 15 x columns >= 16 and rows 80 .. 95 x columns < 16; U: rows 0 .. 15 x columns >= 16.
// w = 32-row chunk of the window, h = 16-row half of the chunk, ks = MFMA step (V: columns 4 ks .., U: rows 8 (ks >> 1) ..)
__device__ __forceinline__ constexpr bool qr_v_tile_zero(int w, int h, int ks) {
    return (w == 0 && h == 0 && ks >= 4) || (w == 2 && h == 1 && ks < 4);
}
__device__ __forceinline__ constexpr bool qr_u_tile_zero(int w, int ks) { return w == 0 && ks < 4; }   // the column half 16 .. 31

struct QrParams {
    const double *v2;       // (n, n): column s = reflectors of sweep s by matrix row
    const double *tau2;     // (n, ks)
    double *vu;             // blocks ((grp - g_lo) * ks + k): V image then U image, QR_BLK doubles each
    double *ct;             // C in slab layout: [slab][row pair][16 NU columns][2 rows] (sbback_slab_kernel), zero-padded
                            // columns: the rows 2 j, 2 j + 1 of a column are adjacent, so a lane's two rows of a register
                            // pair are ONE 16-byte access (8-byte accesses reach 0.54-0.70 of the 16-byte rate on this part)
    int n, ks, ncols;
    int units;              // 16-column units of C; slab b of the gridDim.x slabs holds the units [b units / G, (b+1) units / G)
    int g_lo, g_hi;         // groups [g_lo, g_hi) of this launch (applied from g_hi - 1 down)
    int skip;               // diagnostic bit mask (JXGPU_QB_SKIP): 1 no MFMA phases, 2 no row traffic, 4 no V / U loads,
                            // 8 no row stores, 16 no row loads, 64 lockstep units (three-waves form); balanced form only: 32 no
                            // four-column waves, 256 default wave priority, 512 full counter wait in front of B1, 4096 no L2 warm-up
    double *um;             // balanced form only (else null): third image per block, U in the order the four-column waves read it
};

// U image of the four-column waves (sbback_apply_bal_kernel): lane l = 16 kq + 4 blk + x reads U[q = 16 hh + 4 blk + kq][m = 4 tt + x],
// tt = 0 .. 7, per 16-row half hh.  Under that pattern the [q][m] image costs an 8-way bank conflict per read (a row is one bank
// cycle and a half wave only varies m through x); stored as [hh][u = tt / 2][lane][tt % 2] a lane's eight values are four 16-byte
// reads, each over 64 consecutive 16-byte slots: conflict-free (probe: the first product of those waves 433 -> 365 ms in total).
__host__ __device__ __forceinline__ int qr_um_at(int q, int m) {
    const int hh = q >> 4, blk = (q >> 2) & 3, kq = q & 3, tt = m >> 2, x = m & 3;
    return (((hh * 4 + (tt >> 1)) * 64) + 16 * kq + 4 * blk + x) * 2 + (tt & 1);
}

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
        if (P.um) P.um[((int64_t)blockIdx.y * P.ks + k) * QR_BLK + qr_um_at(q, mp)] = acc;
    }
}

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
    // Staggered units: the units >= QR_EARLY run one barrier interval behind the others, so in every interval one set of waves
    // is in its first product (then writes its partial sums) while the other reads partial sums (then runs its update): the LDS
    // exchange of one set overlaps the MFMAs of the other.  U(k) is then live for the intervals 2k, 2k + 1 and V(k) for 2k + 1,
    // 2k + 2 (2 nk + 1 intervals per group); the loader copies U(k + 1) in interval 2k and V(k + 1) in interval 2k + 1.
    const bool stag = (P.skip & 64) == 0;                      // JXGPU_QB_SKIP=64: lockstep units (547 vs 513 ms at n = 20 000)

    if (wave == 3 * NU) {
        // ------------------------------------------------------------------------------------------------ loader wave
        // images of block (grp, k) -> buffer `buf`: 2 x 24 wave instructions of 1 KB
        auto img_copy = [&](int grp, int k, int buf, int which) {          // which: 1 = V image, 2 = U image, 3 = both
            const char *src = reinterpret_cast<const char *>(P.vu + ((int64_t)(grp - P.g_lo) * P.ks + k) * (2 * QR_BLK)) + lane * 16;
            const unsigned v_dst = (unsigned)(uintptr_t)(vl + buf * QR_BLK), u_dst = (unsigned)(uintptr_t)(ul + buf * QR_BLK);
            if (which & 1)
#pragma unroll
            for (int i = 0; i < QR_BLK * 8 / 1024; ++i) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src + i * 1024), "s"(__builtin_amdgcn_readfirstlane(v_dst + i * 1024))
                             : "memory");
            }
            if (which & 2)
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
            img_copy(grp, 0, 0, 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            qr_lds_barrier();                                  // G1
            for (int k = 0; k < nk; ++k) {
                // staggered: the late units still read V(k - 1) in the first interval of block k, so only U(k + 1) may go now
                if (k + 1 < nk && !(P.skip & 4)) img_copy(grp, k + 1, (k + 1) & 1, stag ? 2 : 3);
                qr_lds_barrier();                              // B2
                if (stag && k + 1 < nk && !(P.skip & 4)) img_copy(grp, k + 1, (k + 1) & 1, 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                qr_lds_barrier();                              // B3
            }
            if (stag) qr_lds_barrier();                        // the late units' last interval
        }
        return;
    }

    // ------------------------------------------------------------------------------------------------------ compute waves
    const int lx = lane & 15, lk = lane >> 4;
    const int unit = wave / 3, j = wave % 3;
    {
        // balanced slabs: a workgroup holds NU or NU - 1 units (qr_plan); the waves of an absent unit only keep the
        // workgroup's barrier sequence (G0, G1, then B2 and B3 per block)
        const int ub = (int)((int64_t)blockIdx.x * P.units / gridDim.x);
        const int nb = (int)((int64_t)(blockIdx.x + 1) * P.units / gridDim.x) - ub;
        if (unit >= nb) {
            for (int grp = P.g_hi - 1; grp >= P.g_lo; --grp) {
                const int s0 = grp * QB_G;
                if (s0 + 1 >= n) continue;
                const int nk = (n - s0 - 1 + QB_SB - 1) / QB_SB;
                __syncthreads();                               // G0
                qr_lds_barrier();                              // G1
                for (int k = 0; k < nk; ++k) {
                    qr_lds_barrier();                          // B2
                    qr_lds_barrier();                          // B3
                }
                if (stag) qr_lds_barrier();
            }
            return;
        }
    }
    // slab layout: the rows of this workgroup's 16 NU columns are contiguous (one row = 128 NU bytes), so a workgroup streams
    // through memory linearly (one read and one write stream per workgroup instead of one per column)
    constexpr int W = NU * 16;
    const int n2 = (n + 1) & ~1;                               // rows of a slab (row pairs)
    double *cp = P.ct + (int64_t)blockIdx.x * n2 * W + 2 * (unit * 16 + lx);   // row pair j of this lane's column: cp + j 2 W
    const int rm_a = 8 * (lx >> 3) + 2 * (lx & 3) + ((lx >> 2) & 1);      // A-operand row lx of a 16-row block -> window row

    // chunk rows rb + [0, 32): register (h, r) <-> row rb + 16 h + 8 (r >> 1) + 2 lk + (r & 1).  The load is branch-free
    // (always a valid row) and raw: rows past the end are zeroed only when the chunk is taken over, so that the loads stay
    // in flight behind the block's arithmetic.  One instruction = 4 rows x 128 bytes.
    // The loads are asm statements hipcc does not count: its own wait for a counted load would also wait for the YOUNGER
    // stores of the finished chunk (the memory counter retires in issue order and hipcc assumes the fewest operations
    // behind a load), and a store acknowledgement under write-back pressure takes longer than a block.  chunk_wait is the
    // only wait for them: vmcnt(8) when exactly eight stores were issued behind the loads, else vmcnt(0).
    // (rb is even: group starts, chunk and window offsets are multiples of 32, so a register pair (2 q, 2 q + 1) is the row
    // pair (rb + 8 q + 2 lk) / 2 of the slab: four 16-byte accesses per chunk, 4 row pairs x 256 bytes per instruction)
    auto chunk_load = [&](int rb, d2 (&raw)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = rb + 8 * q + 2 * lk;
            const double *src = cp + (int64_t)(min(row, n2 - 2) >> 1) * (2 * W);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[q]) : "v"(src) : "memory");
        }
    };
    auto chunk_wait = [&](bool four_stores_behind, d2 (&raw)[4]) {
        if (four_stores_behind)
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]) : : "memory");
    };
    auto chunk_unpack = [&](int rb, const d2 (&raw)[4], d4 (&reg)[2]) {
        const bool inside = rb + 32 <= n;                      // whole chunk inside the matrix (wave-uniform)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            reg[i >> 2][i & 3] = (inside || row < n) ? raw[i >> 1][i & 1] : 0.0;
        }
    };
    auto chunk_store = [&](int rb, const d4 (&reg)[2]) {
        if (rb + 32 <= n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d2 v = {reg[q >> 1][2 * (q & 1)], reg[q >> 1][2 * (q & 1) + 1]};
                *reinterpret_cast<d2 *>(cp + (int64_t)((rb + 8 * q + 2 * lk) >> 1) * (2 * W)) = v;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            if (row < n) cp[(int64_t)(row >> 1) * (2 * W) + (row & 1)] = reg[i >> 2][i & 3];
        }
    };

    d4 cw[2];
    d2 pf[4];
    const bool late = stag && unit >= (NU + 1) / 2;
    for (int grp = P.g_hi - 1; grp >= P.g_lo; --grp) {
        const int s0 = grp * QB_G;
        if (s0 + 1 >= n) continue;
        const int nk = (n - s0 - 1 + QB_SB - 1) / QB_SB;       // steps k with a reflector: s0 + 1 + 64 k < n
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // G0: the previous group's stores are done, LDS is free
        chunk_load(s0 + 32 * j, pf);
        chunk_wait(false, pf);
        chunk_unpack(s0 + 32 * j, pf, cw);
        bool stored8 = false;                                  // exactly four stores were issued behind the last prefetch
        qr_lds_barrier();                                      // G1: the images of block 0 are in place
        if (late) qr_lds_barrier();                            // staggered: the late units start one interval behind
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
            const int wu = __builtin_amdgcn_readfirstlane(w);          // scalar: the zero-tile branches below are s_cbranch
            if (!(P.skip & 1)) {
                // U(q, 16 mb + lx), q = 32 w + 8 (ks >> 1) + 2 lk + (ks & 1): the swizzle bit of the row is lk & 1
                const double *up = ul + (k & 1) * QR_BLK + (32 * w + 2 * lk) * QB_G + lx;
                const int o0 = 16 * (lk & 1), o1 = 16 - o0;
                // one straight-line instance per chunk position (branches around single MFMAs cost more than the skipped ones)
                auto ypart = [&](auto wc) {
                    constexpr int WC = decltype(wc)::value;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const int q = 8 * (ks >> 1) + (ks & 1);
                        const double b = cw[ks >> 2][ks & 3];
                        y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o0], b, y0, 0, 0, 0);
                        if (!qr_u_tile_zero(WC, ks)) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o1], b, y1, 0, 0, 0);
                    }
                };
                if (wu == 0)
                    ypart(std::integral_constant<int, 0>{});
                else
                    ypart(std::integral_constant<int, 1>{});
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
                auto upd = [&](auto wc) {
                    constexpr int WC = decltype(wc)::value;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const int m = (4 * ks + lk) ^ qr_v_swz(rm_a);
                        if (!qr_v_tile_zero(WC, 0, ks)) cw[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[m], yn[ks], cw[0], 0, 0, 0);
                        if (!qr_v_tile_zero(WC, 1, ks))
                            cw[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[16 * QB_G + m], yn[ks], cw[1], 0, 0, 0);
                    }
                };
                if (wu == 0)
                    upd(std::integral_constant<int, 0>{});
                else if (wu == 1)
                    upd(std::integral_constant<int, 1>{});
                else
                    upd(std::integral_constant<int, 2>{});
            }
            qr_lds_barrier();                                  // B3: the partial sums have been read, the images of block k + 1 are in place
            if (!has_next && !(P.skip & 2)) chunk_store(wb + 32 * w, cw);
            w = (w == 2) ? 0 : w + 1;
        }
        if (stag && !late) qr_lds_barrier();                   // the late units' last interval
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One wave per unit (large n: at least eight 16-column units per CU).  A wave holds the WHOLE 96-row window of its unit in
// registers (three 32-row chunk sets of the layout above), so Y = U' Cwin is complete inside the wave -- its accumulator
// layout is the B-operand layout of the update, as above -- and nothing is exchanged between waves: no partial sums, ONE
// workgroup barrier per block (the hand-over of the image buffers), 96 MFMAs per wave between barriers instead of 16 + 16.
// The chunk set at window position p of step k is set (p - k) mod 3 (the set at position 2 keeps its rows when the window
// slides); the step loop is unrolled by three so that the set indices are compile-time.  Row traffic per step: the two
// leading chunks are final after the update and are stored at the END of the step (behind the prefetch of the two next
// chunks, issued at the top of the step: the wait at the top of step k + 1 is vmcnt(8) = "everything but my eight stores").
// LDS: the two image buffers only (96 KB); up to 15 compute waves + the loader wave per workgroup.
template <int NW>
__global__ __launch_bounds__((NW + 1) * 64) void sbback_apply_solo_kernel(QrParams P) {
    extern __shared__ __attribute__((aligned(16))) double qb_smem[];
    double *vl = qb_smem;                                      // [2][QR_BLK]
    double *ul = vl + 2 * QR_BLK;                              // [2][QR_BLK]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = P.n;

    if (wave == NW) {
        // ------------------------------------------------------------------------------------------------ loader wave
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
                if (k + 1 < nk) img_copy(grp, k + 1, (k + 1) & 1);      // the buffer block k - 1 used: everyone is past B(k - 1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                qr_lds_barrier();                              // B(k)
            }
        }
        return;
    }

    // ------------------------------------------------------------------------------------------------------ compute waves
    const int lx = lane & 15, lk = lane >> 4;
    const int unit = wave;
    {
        const int ub = (int)((int64_t)blockIdx.x * P.units / gridDim.x);
        const int nb = (int)((int64_t)(blockIdx.x + 1) * P.units / gridDim.x) - ub;
        if (unit >= nb) {                                      // absent unit of a narrower slab: the barrier sequence only
            for (int grp = P.g_hi - 1; grp >= P.g_lo; --grp) {
                const int s0 = grp * QB_G;
                if (s0 + 1 >= n) continue;
                const int nk = (n - s0 - 1 + QB_SB - 1) / QB_SB;
                __syncthreads();                               // G0
                qr_lds_barrier();                              // G1
                for (int k = 0; k < nk; ++k) qr_lds_barrier(); // B(k)
            }
            return;
        }
    }
    constexpr int W = NW * 16;
    const int n2 = (n + 1) & ~1;
    double *cp = P.ct + (int64_t)blockIdx.x * n2 * W + 2 * (unit * 16 + lx);
    const int rm_a = 8 * (lx >> 3) + 2 * (lx & 3) + ((lx >> 2) & 1);

    auto chunk_load = [&](int rb, d2 (&raw)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = rb + 8 * q + 2 * lk;
            const double *src = cp + (int64_t)(min(row, n2 - 2) >> 1) * (2 * W);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[q]) : "v"(src) : "memory");
        }
    };
    auto chunk_unpack = [&](int rb, const d2 (&raw)[4], d4 (&reg)[2]) {
        const bool inside = rb + 32 <= n;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            reg[i >> 2][i & 3] = (inside || row < n) ? raw[i >> 1][i & 1] : 0.0;
        }
    };
    auto chunk_store = [&](int rb, const d4 (&reg)[2]) {
        if (rb + 32 <= n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d2 v = {reg[q >> 1][2 * (q & 1)], reg[q >> 1][2 * (q & 1) + 1]};
                *reinterpret_cast<d2 *>(cp + (int64_t)((rb + 8 * q + 2 * lk) >> 1) * (2 * W)) = v;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            if (row < n) cp[(int64_t)(row >> 1) * (2 * W) + (row & 1)] = reg[i >> 2][i & 3];
        }
    };

    d4 S[3][2];
    d2 pf[2][4];
    bool stored8 = false;
    int s0 = 0, nk = 0;
    // one block: PH = k mod 3 (compile time), the set at window position p is S[(p + 3 - PH) % 3]
    auto step = [&](auto phc, int k) {
        constexpr int PH = decltype(phc)::value;
        d4 (&a0)[2] = S[(0 + 3 - PH) % 3];
        d4 (&a1)[2] = S[(1 + 3 - PH) % 3];
        d4 (&a2)[2] = S[(2 + 3 - PH) % 3];
        const int wb = s0 + k * QB_SB;
        const bool has_next = k + 1 < nk;
        if (k > 0) {
            if (stored8)
                asm volatile("s_waitcnt vmcnt(8)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]),
                             "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]) : : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]),
                             "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]) : : "memory");
            chunk_unpack(wb + 32, pf[0], a1);
            chunk_unpack(wb + 64, pf[1], a2);
        }
        if (has_next && !(P.skip & 18)) {
            chunk_load(wb + QB_WIN, pf[0]);
            chunk_load(wb + QB_WIN + 32, pf[1]);
        }
        const int buf = k & 1;
        d4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
        if (!(P.skip & 1)) {
            const int o0 = 16 * (lk & 1), o1 = 16 - o0;
            auto ypart = [&](int w, const d4 (&cw)[2]) {
                const double *up = ul + buf * QR_BLK + (32 * w + 2 * lk) * QB_G + lx;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int q = 8 * (ks >> 1) + (ks & 1);
                    const double b = cw[ks >> 2][ks & 3];
                    y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o0], b, y0, 0, 0, 0);
                    if (!qr_u_tile_zero(w, ks)) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o1], b, y1, 0, 0, 0);
                }
            };
            ypart(0, a0);
            ypart(1, a1);
            ypart(2, a2);
            double yn[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                yn[r] = -y0[r];
                yn[4 + r] = -y1[r];
            }
            auto upd = [&](int w, d4 (&cw)[2]) {
                const double *vp = vl + buf * QR_BLK + (32 * w + rm_a) * QB_G;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int m = (4 * ks + lk) ^ qr_v_swz(rm_a);
                    if (!qr_v_tile_zero(w, 0, ks)) cw[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[m], yn[ks], cw[0], 0, 0, 0);
                    if (!qr_v_tile_zero(w, 1, ks)) cw[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[16 * QB_G + m], yn[ks], cw[1], 0, 0, 0);
                }
            };
            upd(0, a0);
            upd(1, a1);
            upd(2, a2);
        }
        qr_lds_barrier();                                      // B(k): images of block k + 1 in place, buffer of block k free
        stored8 = false;
        if (!(P.skip & 10)) {
            chunk_store(wb, a0);
            chunk_store(wb + 32, a1);
            stored8 = wb + 64 <= n;
            if (!has_next) chunk_store(wb + 64, a2);
        }
    };
    for (int grp = P.g_hi - 1; grp >= P.g_lo; --grp) {
        s0 = grp * QB_G;
        if (s0 + 1 >= n) continue;
        nk = (n - s0 - 1 + QB_SB - 1) / QB_SB;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // G0: the previous group's stores are done, LDS is free
        chunk_load(s0, pf[0]);
        chunk_load(s0 + 32, pf[1]);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]), "+v"(pf[1][1]),
                     "+v"(pf[1][2]), "+v"(pf[1][3]) : : "memory");
        chunk_unpack(s0, pf[0], S[0]);
        chunk_unpack(s0 + 32, pf[1], S[1]);
        chunk_load(s0 + 64, pf[0]);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]) : : "memory");
        chunk_unpack(s0 + 64, pf[0], S[2]);
        stored8 = false;
        qr_lds_barrier();                                      // G1: the images of block 0 are in place
        int k = 0;
        for (; k + 3 <= nk; k += 3) {
            step(std::integral_constant<int, 0>{}, k);
            step(std::integral_constant<int, 1>{}, k + 1);
            step(std::integral_constant<int, 2>{}, k + 2);
        }
        if (k < nk) step(std::integral_constant<int, 0>{}, k);
        if (k + 1 < nk) step(std::integral_constant<int, 1>{}, k + 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One wave per unit, TWO sweep groups per pass over the rows.  At n = 50 000 the one-group form streams the whole of C once
// per group: 62.5 TB per decomposition, 6.9 TB/s -- the HBM roof, not the matrix pipes, bounds it.  Block (g - 1, k) touches
// the rows [32 (g - 1) + 64 k, + 96), block (g, k) the rows [32 g + 64 k, + 96): (g - 1, k) has to follow (g, k) and (g, k - 1)
// and commutes with every (g, k' > k) (disjoint rows), so the order (g, 0), (g - 1, 0), (g, 1), (g - 1, 1), ... is a valid
// one and both blocks of a step live in the 128-row window [32 (g - 1) + 64 k, + 128): four chunk sets in registers, block
// (g, k) on window positions 1 - 3, block (g - 1, k) on positions 0 - 2, then the two leading chunks are final for BOTH groups.
// Row traffic per pair of groups = that of one group before; MFMA work unchanged (the 96-row parallelograms stay).
// The two image buffers alternate between the upper and the lower group: buffer A holds (g, k), B holds (g - 1, k); the
// loader refills A with (g, k + 1) while the waves work on B and B with (g - 1, k + 1) while they work on A: one barrier per
// block, 96 KB of LDS as before.  The set at window position p of step k is set (p + 2 k) mod 4: two compile-time phases.
template <int NW>
__global__ __launch_bounds__((NW + 1) * 64) void sbback_apply_pair_kernel(QrParams P) {
    extern __shared__ __attribute__((aligned(16))) double qb_smem[];
    double *vl = qb_smem;                                      // [2][QR_BLK]: 0 = upper group's block, 1 = lower group's
    double *ul = vl + 2 * QR_BLK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = P.n;
    // passes: pairs (g, g - 1) from g_hi - 1 down; an odd count leaves the lowest group alone (upper block absent)
    auto pass_groups = [&](int pass, int &g_up, int &g_lo_grp) {
        g_up = P.g_hi - 1 - 2 * pass;
        g_lo_grp = g_up - 1;
        if (g_lo_grp < P.g_lo) {                               // single group: it plays the lower role
            g_lo_grp = g_up;
            g_up = -1;
        }
    };
    const int npass = (P.g_hi - P.g_lo + 1) / 2;
    auto steps_of = [&](int grp) { return (grp < 0 || grp * QB_G + 1 >= n) ? 0 : (n - grp * QB_G - 1 + QB_SB - 1) / QB_SB; };

    if (wave == NW) {
        // ------------------------------------------------------------------------------------------------ loader wave
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
        for (int pass = 0; pass < npass; ++pass) {
            int gu, gl;
            pass_groups(pass, gu, gl);
            const int nku = steps_of(gu), nkl = steps_of(gl);
            if (nkl == 0) continue;
            __syncthreads();                                   // G0
            if (nku > 0) img_copy(gu, 0, 0);
            img_copy(gl, 0, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            qr_lds_barrier();                                  // G1
            for (int k = 0; k < nkl; ++k) {
                qr_lds_barrier();                              // B1(k): buffer 0 consumed
                if (k + 1 < nku) img_copy(gu, k + 1, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                qr_lds_barrier();                              // B2(k): buffer 1 consumed, buffer 0 holds (gu, k + 1)
                if (k + 1 < nkl) img_copy(gl, k + 1, 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        return;
    }

    // ------------------------------------------------------------------------------------------------------ compute waves
    const int lx = lane & 15, lk = lane >> 4;
    const int unit = wave;
    {
        const int ub = (int)((int64_t)blockIdx.x * P.units / gridDim.x);
        const int nb = (int)((int64_t)(blockIdx.x + 1) * P.units / gridDim.x) - ub;
        if (unit >= nb) {                                      // absent unit of a narrower slab: the barrier sequence only
            for (int pass = 0; pass < npass; ++pass) {
                int gu, gl;
                pass_groups(pass, gu, gl);
                const int nkl = steps_of(gl);
                if (nkl == 0) continue;
                __syncthreads();                               // G0
                qr_lds_barrier();                              // G1
                for (int k = 0; k < nkl; ++k) {
                    qr_lds_barrier();                          // B1
                    qr_lds_barrier();                          // B2
                }
            }
            return;
        }
    }
    constexpr int W = NW * 16;
    const int n2 = (n + 1) & ~1;
    double *cp = P.ct + (int64_t)blockIdx.x * n2 * W + 2 * (unit * 16 + lx);
    const int rm_a = 8 * (lx >> 3) + 2 * (lx & 3) + ((lx >> 2) & 1);

    auto chunk_load = [&](int rb, d2 (&raw)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = rb + 8 * q + 2 * lk;
            const double *src = cp + (int64_t)(min(row, n2 - 2) >> 1) * (2 * W);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[q]) : "v"(src) : "memory");
        }
    };
    auto chunk_unpack = [&](int rb, const d2 (&raw)[4], d4 (&reg)[2]) {
        const bool inside = rb + 32 <= n;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            reg[i >> 2][i & 3] = (inside || row < n) ? raw[i >> 1][i & 1] : 0.0;
        }
    };
    auto chunk_store = [&](int rb, const d4 (&reg)[2]) {
        if (rb + 32 <= n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d2 v = {reg[q >> 1][2 * (q & 1)], reg[q >> 1][2 * (q & 1) + 1]};
                *reinterpret_cast<d2 *>(cp + (int64_t)((rb + 8 * q + 2 * lk) >> 1) * (2 * W)) = v;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            if (row < n) cp[(int64_t)(row >> 1) * (2 * W) + (row & 1)] = reg[i >> 2][i & 3];
        }
    };
    // one 96-row block on three chunk sets (image rows 32 w + ... of buffer `buf` <-> set cw_w)
    auto block = [&](int buf, d4 (&c0)[2], d4 (&c1)[2], d4 (&c2)[2]) {
        d4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
        const int o0 = 16 * (lk & 1), o1 = 16 - o0;
        auto ypart = [&](int w, const d4 (&cw)[2]) {
            const double *up = ul + buf * QR_BLK + (32 * w + 2 * lk) * QB_G + lx;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int q = 8 * (ks >> 1) + (ks & 1);
                const double b = cw[ks >> 2][ks & 3];
                y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o0], b, y0, 0, 0, 0);
                if (!qr_u_tile_zero(w, ks)) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o1], b, y1, 0, 0, 0);
            }
        };
        ypart(0, c0);
        ypart(1, c1);
        ypart(2, c2);
        double yn[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            yn[r] = -y0[r];
            yn[4 + r] = -y1[r];
        }
        auto upd = [&](int w, d4 (&cw)[2]) {
            const double *vp = vl + buf * QR_BLK + (32 * w + rm_a) * QB_G;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int m = (4 * ks + lk) ^ qr_v_swz(rm_a);
                if (!qr_v_tile_zero(w, 0, ks)) cw[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[m], yn[ks], cw[0], 0, 0, 0);
                if (!qr_v_tile_zero(w, 1, ks)) cw[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[16 * QB_G + m], yn[ks], cw[1], 0, 0, 0);
            }
        };
        upd(0, c0);
        upd(1, c1);
        upd(2, c2);
    };

    d4 S[4][2];
    d2 pf[2][4];
    bool stored8 = false;
    int sl = 0, nku = 0, nkl = 0;
    auto wait_pf = [&](bool eight_behind) {
        if (eight_behind)
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]),
                         "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]),
                         "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]) : : "memory");
    };
    // one step of a pass: PH = k mod 2 (compile time), the set at window position p is S[(p + 2 PH) % 4]
    auto step = [&](auto phc, int k) {
        constexpr int PH = decltype(phc)::value;
        d4 (&a0)[2] = S[(0 + 2 * PH) % 4];
        d4 (&a1)[2] = S[(1 + 2 * PH) % 4];
        d4 (&a2)[2] = S[(2 + 2 * PH) % 4];
        d4 (&a3)[2] = S[(3 + 2 * PH) % 4];
        const int wb = sl + k * QB_SB;                         // first row of the 128-row window
        const bool has_next = k + 1 < nkl;
        if (k > 0) {
            wait_pf(stored8);
            chunk_unpack(wb + 64, pf[0], a2);
            chunk_unpack(wb + 96, pf[1], a3);
        }
        if (has_next && !(P.skip & 18)) {
            chunk_load(wb + 128, pf[0]);
            chunk_load(wb + 160, pf[1]);
        }
        if (k < nku && !(P.skip & 1)) block(0, a1, a2, a3);    // block (g, k): rows wb + 32 ...
        qr_lds_barrier();                                      // B1(k)
        if (!(P.skip & 1)) block(1, a0, a1, a2);               // block (g - 1, k): rows wb ...
        qr_lds_barrier();                                      // B2(k)
        stored8 = false;
        if (!(P.skip & 10)) {
            chunk_store(wb, a0);
            chunk_store(wb + 32, a1);
            stored8 = wb + 64 <= n;
            if (!has_next) {
                chunk_store(wb + 64, a2);
                chunk_store(wb + 96, a3);
            }
        }
    };
    for (int pass = 0; pass < npass; ++pass) {
        int gu, gl;
        pass_groups(pass, gu, gl);
        nku = steps_of(gu);
        nkl = steps_of(gl);
        if (nkl == 0) continue;
        sl = gl * QB_G;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // G0: the previous pass' stores are done, LDS is free
        chunk_load(sl, pf[0]);
        chunk_load(sl + 32, pf[1]);
        wait_pf(false);
        chunk_unpack(sl, pf[0], S[0]);
        chunk_unpack(sl + 32, pf[1], S[1]);
        chunk_load(sl + 64, pf[0]);
        chunk_load(sl + 96, pf[1]);
        wait_pf(false);
        chunk_unpack(sl + 64, pf[0], S[2]);
        chunk_unpack(sl + 96, pf[1], S[3]);
        stored8 = false;
        qr_lds_barrier();                                      // G1: the images of the first two blocks are in place
        int k = 0;
        for (; k + 2 <= nkl; k += 2) {
            step(std::integral_constant<int, 0>{}, k);
            step(std::integral_constant<int, 1>{}, k + 1);
        }
        if (k < nkl) step(std::integral_constant<int, 0>{}, k);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Balanced form for FIVE units per CU (1024 < units <= 1280 on 256 CUs: n = 16 400 ... 20 480, BASELINE configs[2]).  With one
// wave per unit a CU holds five compute waves on four SIMDs: the SIMD with two of them sets the pace (measured 536 ms at
// n = 20 000, the three-waves-per-unit form 516 ms -- its 15 waves meet at two barriers per block).  Here every SIMD gets the
// same matrix-pipe load: waves 0 - 3 own one 16-column unit each (the two-groups-per-pass code of sbback_apply_pair_kernel,
// v_mfma_f64_16x16x4_f64), waves 4 - 7 own FOUR columns each of the fifth unit on v_mfma_f64_4x4x4_f64 (four 4 x 4 x 4 blocks
// per instruction at the same flop rate -- 17.3 against 64.8 cycles, scripts/probes/mfma_rate_probe.hip: a quarter of a unit's
// matrix-pipe time) and share the loader's work (each issues a quarter of the LDS-DMA copies of the next V / U images).
// 1.25 units per SIMD, no partial-sum exchange, one barrier per block.
// 4 x 4 x 4 layout (probed, scripts/probes/mfma4x4_probe.hip): lane = 16 kq + 4 blk + x; A_blk[i = x][k = kq], B[k = kq][c = 4 blk + x],
// D[i = kq][c = 4 blk + x] = sum_k A_blk(c)[i][k] B[k][c].  Window rows 16 h + 4 blk + kq of a 32-row set live at lane (kq, blk, x = column):
// at once the B operand of the first product (Y' partial per blk: rows 4 blk + kq; summed over blk by two DPP row rotations) and
// the accumulator of the second (blk = row sub-block, Y replicated over blk).
template <int CTRL>
__device__ __forceinline__ double qb_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);      // old = source: no zero-initialised register per call
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int W>                                               // slab width: 16 x (4 .. 7 units); 64: waves 4 - 7 are the loader only
__global__ __launch_bounds__(512) void sbback_apply_bal_kernel(QrParams P) {
    extern __shared__ __attribute__((aligned(16))) double qb_smem[];
    double *vl = qb_smem;                                      // [2][QR_BLK]: 0 = upper group's block, 1 = lower group's
    double *ul = vl + 2 * QR_BLK;
    double *uml = ul + 2 * QR_BLK;                             // [2][QR_BLK]: U again, in the four-column waves' order (qr_um_at)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = P.n;
    auto pass_groups = [&](int pass, int &g_up, int &g_lo_grp) {
        g_up = P.g_hi - 1 - 2 * pass;
        g_lo_grp = g_up - 1;
        if (g_lo_grp < P.g_lo) {                               // single group: it plays the lower role
            g_lo_grp = g_up;
            g_up = -1;
        }
    };
    const int npass = (P.g_hi - P.g_lo + 1) / 2;
    auto steps_of = [&](int grp) { return (grp < 0 || grp * QB_G + 1 >= n) ? 0 : (n - grp * QB_G - 1 + QB_SB - 1) / QB_SB; };
    const int ub = (int)((int64_t)blockIdx.x * P.units / gridDim.x);
    const int nb = (int)((int64_t)(blockIdx.x + 1) * P.units / gridDim.x) - ub;
    const int n2 = (n + 1) & ~1;

    if (wave >= 4) {
        // ------------------------------------------------------------------------ four-column waves (+ a quarter of the loader)
        const int mw = wave - 4;
        const int kq = lane >> 4, blk = (lane >> 2) & 3, x = lane & 3;
        // the short 4 x 4 x 4 instructions of this wave go first whenever they are ready (JXGPU_QB_SKIP=256: default priority): the
        // 16-column wave on the same SIMD is never short of ready work, and with equal priorities the older wave wins every
        // arbitration, so this wave's products would only start when the other one has reached the barrier
        if (!(P.skip & 256)) __builtin_amdgcn_s_setprio(3);
        const bool have = W > 64 && nb >= 5 && !(P.skip & 32);            // a slab of four units has no fifth: loader duty only
        auto dma_part = [&](int grp, int k, int buf) {
            const char *src = reinterpret_cast<const char *>(P.vu + ((int64_t)(grp - P.g_lo) * P.ks + k) * (2 * QR_BLK)) + lane * 16;
            const unsigned v_dst = (unsigned)(uintptr_t)(vl + buf * QR_BLK), u_dst = (unsigned)(uintptr_t)(ul + buf * QR_BLK);
            constexpr int PER = QR_BLK * 8 / 1024 / 4;         // 6 of the 24 KB-sized copies of each image
#pragma unroll
            for (int ii = 0; ii < PER; ++ii) {
                const int i = mw * PER + ii;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src + i * 1024), "s"(__builtin_amdgcn_readfirstlane(v_dst + i * 1024))
                             : "memory");
            }
#pragma unroll
            for (int ii = 0; ii < PER; ++ii) {
                const int i = mw * PER + ii;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src + QR_BLK * 8 + i * 1024), "s"(__builtin_amdgcn_readfirstlane(u_dst + i * 1024))
                             : "memory");
            }
            if (W == 64) return;
            const char *srcm = reinterpret_cast<const char *>(P.um + ((int64_t)(grp - P.g_lo) * P.ks + k) * QR_BLK) + lane * 16;
            const unsigned m_dst = (unsigned)(uintptr_t)(uml + buf * QR_BLK);
#pragma unroll
            for (int ii = 0; ii < PER; ++ii) {
                const int i = mw * PER + ii;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(srcm + i * 1024), "s"(__builtin_amdgcn_readfirstlane(m_dst + i * 1024))
                             : "memory");
            }
        };
        // this wave's share: four columns of each unit beyond the fourth of the slab (mc = nb - 4 units, at most MCMAX)
        constexpr int MCMAX = W / 16 - 4 > 0 ? W / 16 - 4 : 1;
        const int mc = have ? nb - 4 : 0;
        double *cpm = P.ct + (int64_t)blockIdx.x * n2 * W + 2 * (64 + 4 * mw + x);      // + 32 qd doubles for the unit 4 + qd
        // a 32-row set: registers [qd][h] <-> row rb + 16 h + 4 blk + kq of unit 4 + qd; one 16-byte access = the row pair of that row
        auto set_load = [&](int rb, d2 (&raw)[MCMAX][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int qd = 0; qd < MCMAX; ++qd) {
                if (qd >= mc) continue;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = rb + 16 * h + 4 * blk + kq;
                    // a plain load (the compiler tracks it): the asm loads of the 16-column code leave the register unwritten until
                    // the data lands, which is only safe while the compiler never copies the value in between -- with two or
                    // three units per wave it does (v_mov_b64 of the in-flight registers: garbage).  The memory-clobbering asm
                    // statements around keep the load where it is written; the compiler's own wait sits in front of the first use,
                    // behind the counter wait of the barrier before
                    const double *src = cpm + 32 * qd + (int64_t)(min(row, n2 - 2) >> 1) * (2 * W);
                    raw[qd][h] = *reinterpret_cast<const d2 *>(src);
                }
            }
        };
        auto set_unpack = [&](int rb, const d2 (&raw)[MCMAX][2], double (&reg)[MCMAX][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int qd = 0; qd < MCMAX; ++qd)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = rb + 16 * h + 4 * blk + kq;
                    const double el = (kq & 1) ? raw[qd][h][1] : raw[qd][h][0];      // (a run-time vector index would go through scratch)
                    reg[qd][h] = (row < n) ? el : 0.0;
                }
        };
        auto set_store = [&](int rb, const double (&reg)[MCMAX][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int qd = 0; qd < MCMAX; ++qd) {
                if (qd >= mc) continue;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = rb + 16 * h + 4 * blk + kq;
                    // asm: a store the compiler tracked would make it wait (by its own count, which does not know the image copies)
                    // before the next step's loads reuse a register -- i.e. for the copies just issued and the stores' acknowledgements
                    double *dst = cpm + 32 * qd + (int64_t)(row >> 1) * (2 * W) + (row & 1);
                    if (row < n) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(dst), "v"(reg[qd][h]) : "memory");
                }
            }
        };
        // LDS addresses of the V operands: the bank swizzle of the image (qr_v_at) depends on the lane only, not on the 16-row half
        // hh of the window: eight lane terms + hh * 512 doubles as an immediate offset
        const int vsw = qr_v_swz(4 * blk + x);
        int voff[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) voff[i] = (4 * blk + x) * QB_G + ((4 * i + kq) ^ vsw);       // V[q = 16 hh + 4 blk + x][m = 4 ks + kq]
        // one 96-row block on three sets (image rows 32 w + 16 h + ...), every unit of this wave with the same operands
        auto mini_block = [&](int buf, double (&c0)[MCMAX][2], double (&c1)[MCMAX][2], double (&c2)[MCMAX][2]) __attribute__((always_inline)) {
            const double *vbuf = vl + buf * QR_BLK;
            // operands one 16-row half (first product) / one 4-column step (second product) ahead of the instructions that use
            // them: eight or six LDS reads in flight behind the previous step's products.  (All 84 reads of the block at once
            // made the four waves of this kind queue ~4800 LDS cycles right behind every barrier -- the 8-way bank conflicts of
            // the [q][m] U image under this access pattern included --, and the 16-column waves waited for their own operands
            // behind that queue: the block interval grew by about that much.)
            double ua[2][8], va[2][6];
            const d2 *umb = reinterpret_cast<const d2 *>(uml + buf * QR_BLK) + lane;
            auto u_load = [&](int hh, double (&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (hh == 0 && u >= 2) continue;           // U[q][m] = 0 for q <= m: rows 0 .. 15 against columns >= 16
                    const d2 pr = umb[(hh * 4 + u) * 64];
                    dst[2 * u] = pr[0];
                    dst[2 * u + 1] = pr[1];
                }
            };
            auto v_load = [&](int ks, double (&dst)[6]) __attribute__((always_inline)) {
#pragma unroll
                for (int hh = 0; hh < 6; ++hh)
                    if (!((hh == 0 && ks >= 4) || (hh == 5 && ks < 4))) dst[hh] = vbuf[voff[ks] + hh * 16 * QB_G];   // V[q][m] != 0: m < q <= m + 64
            };
            double y[MCMAX][8];
#pragma unroll
            for (int qd = 0; qd < MCMAX; ++qd)
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) y[qd][tt] = 0.0;
            u_load(0, ua[0]);
#pragma unroll
            for (int hh = 0; hh < 6; ++hh) {
                if (hh + 1 < 6) u_load(hh + 1, ua[(hh + 1) & 1]);
                else v_load(0, va[0]);
#pragma unroll
                for (int qd = 0; qd < MCMAX; ++qd) {
                    if (qd >= mc) continue;
                    const double b = hh < 2 ? c0[qd][hh] : (hh < 4 ? c1[qd][hh - 2] : c2[qd][hh - 4]);
#pragma unroll
                    for (int tt = 0; tt < 8; ++tt) {
                        if (hh == 0 && tt >= 4) continue;
                        y[qd][tt] = __builtin_amdgcn_mfma_f64_4x4x4f64(ua[hh & 1][tt], b, y[qd][tt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int qd = 0; qd < MCMAX; ++qd) {
                if (qd >= mc) continue;
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) {
                    double sm = y[qd][tt];
                    sm += qb_dpp<0x124>(sm);                   // row_ror:4
                    sm += qb_dpp<0x128>(sm);                   // row_ror:8: the sum over the four blocks in every lane
                    y[qd][tt] = -sm;
                }
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) v_load(ks + 1, va[(ks + 1) & 1]);
#pragma unroll
                for (int qd = 0; qd < MCMAX; ++qd) {
                    if (qd >= mc) continue;
#pragma unroll
                    for (int hh = 0; hh < 6; ++hh) {           // six independent accumulators per step
                        if ((hh == 0 && ks >= 4) || (hh == 5 && ks < 4)) continue;
                        double &cw = hh < 2 ? c0[qd][hh] : (hh < 4 ? c1[qd][hh - 2] : c2[qd][hh - 4]);
                        cw = __builtin_amdgcn_mfma_f64_4x4x4f64(va[ks & 1][hh], y[qd][ks], cw, 0, 0, 0);
                    }
                }
            }
        };
        double M[4][MCMAX][2];
        d2 pfm[2][MCMAX][2];
        // wait until at most `younger` of this wave's vector-memory operations are outstanding (4 per unit and kind: the counts
        // are those of mc = 1 .. 3 units)
        auto wait_vm = [&](int younger) __attribute__((always_inline)) {
            switch (younger) {
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        };
        // everything landed; the prefetch registers pass through the statement so that their unpacking cannot be scheduled in front
        // of it (the selects have no other dependence: hoisted into the second block they took the compiler's wait with them)
        auto wait_all_pf = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int qd = 0; qd < MCMAX; ++qd)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(pfm[0][qd][0]), "+v"(pfm[0][qd][1]), "+v"(pfm[1][qd][0]), "+v"(pfm[1][qd][1]) : : "memory");
        };
        for (int pass = 0; pass < npass; ++pass) {
            int gu, gl;
            pass_groups(pass, gu, gl);
            const int nku = steps_of(gu), nkl = steps_of(gl);
            if (nkl == 0) continue;
            const int sl = gl * QB_G;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                   // G0
            if (nku > 0) dma_part(gu, 0, 0);
            dma_part(gl, 0, 1);
            if (have) {
                set_load(sl, pfm[0]);
                set_load(sl + 32, pfm[1]);
                wait_all_pf();
                set_unpack(sl, pfm[0], M[0]);
                set_unpack(sl + 32, pfm[1], M[1]);
                set_load(sl + 64, pfm[0]);
                set_load(sl + 96, pfm[1]);
                wait_all_pf();
                set_unpack(sl + 64, pfm[0], M[2]);
                set_unpack(sl + 96, pfm[1], M[3]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            qr_lds_barrier();                                  // G1: the images of the first two blocks are in place
            auto step = [&](auto phc, int k) __attribute__((always_inline)) {
                constexpr int PH = decltype(phc)::value;
                double (&a0)[MCMAX][2] = M[(0 + 2 * PH) % 4];
                double (&a1)[MCMAX][2] = M[(1 + 2 * PH) % 4];
                double (&a2)[MCMAX][2] = M[(2 + 2 * PH) % 4];
                double (&a3)[MCMAX][2] = M[(3 + 2 * PH) % 4];
                const int wb = sl + k * QB_SB;
                const bool has_next = k + 1 < nkl;
                if (!have) {                                   // loader duty only
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    qr_lds_barrier();                          // B1(k)
                    if (k + 1 < nku && !(P.skip & 4)) dma_part(gu, k + 1, 0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    qr_lds_barrier();                          // B2(k)
                    if (k + 1 < nkl && !(P.skip & 4)) dma_part(gl, k + 1, 1);
                    return;
                }
                // The row prefetch and its unpacking are UNCONDITIONAL on this path (row addresses are clamped; the last step of
                // a pass loads 4 mc row pairs nobody uses): the compiler tracks these loads with its own counter, and on a path
                // where loads might be pending it waits -- by its count, which knows nothing of the image copies -- before the
                // next loads reuse the registers; with conditions it cannot correlate that wait landed on every step.
                set_load(wb + 128, pfm[0]);
                set_load(wb + 160, pfm[1]);
                if (k < nku) mini_block(0, a1, a2, a3);        // block (g, k): rows wb + 32 ...
                // the copies into buffer 1 were issued behind B2 of the previous step; younger than them are only that step's 4 mc
                // row stores and this step's 4 mc prefetch loads (the counter retires in issue order): waiting for everything would
                // put an HBM round trip of the prefetch in front of every barrier
                if (k > 0 && !(P.skip & 512)) wait_vm(8 * mc);
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                qr_lds_barrier();                              // B1(k): buffer 0 consumed, buffer 1 holds (gl, k)
                if (k + 1 < nku) dma_part(gu, k + 1, 0);
                mini_block(1, a0, a1, a2);                     // block (g - 1, k): rows wb ...
                wait_all_pf();
                // the prefetched rows are taken out of their load registers HERE, behind the wait that has just drained the counter:
                // at the top of the next step the compiler's own wait in front of the first use would also cover the image copies
                // issued behind B2 (invisible to it), i.e. the copies' latency would no longer hide behind the first block
                double nx[2][MCMAX][2];
                set_unpack(wb + 128, pfm[0], nx[0]);
                set_unpack(wb + 160, pfm[1], nx[1]);
#pragma unroll
                for (int qd = 0; qd < MCMAX; ++qd)             // materialise the values here (the scheduler would sink the selects -- and
#pragma unroll                                                 // with them its wait -- to their use behind the copies and the stores)
                    for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(nx[0][qd][h]), "+v"(nx[1][qd][h]));
                qr_lds_barrier();                              // B2(k): buffer 1 consumed, buffer 0 holds (gu, k + 1)
                if (k + 1 < nkl) dma_part(gl, k + 1, 1);
                set_store(wb, a0);                             // 4 mc stores (the last step's other two sets: behind the loop)
                set_store(wb + 32, a1);
                if (!has_next) {
                    set_store(wb + 64, a2);
                    set_store(wb + 96, a3);
                }
#pragma unroll
                for (int qd = 0; qd < MCMAX; ++qd)             // the two leading sets are stored: they become the next step's a2, a3
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        a0[qd][h] = nx[0][qd][h];
                        a1[qd][h] = nx[1][qd][h];
                    }
            };
            int k = 0;
            for (; k + 2 <= nkl; k += 2) {
                step(std::integral_constant<int, 0>{}, k);
                step(std::integral_constant<int, 1>{}, k + 1);
            }
            if (k < nkl) step(std::integral_constant<int, 0>{}, k);
        }
        return;
    }

    // ------------------------------------------------------------------------------------------------ 16-column waves
    const int lx = lane & 15, lk = lane >> 4;
    const int unit = wave;
    if (unit >= nb) {                                          // absent unit of a narrower slab: the barrier sequence only
        for (int pass = 0; pass < npass; ++pass) {
            int gu, gl;
            pass_groups(pass, gu, gl);
            const int nkl = steps_of(gl);
            if (nkl == 0) continue;
            __syncthreads();                                   // G0
            qr_lds_barrier();                                  // G1
            for (int k = 0; k < nkl; ++k) {
                qr_lds_barrier();                              // B1
                qr_lds_barrier();                              // B2
            }
        }
        return;
    }
    double *cp = P.ct + (int64_t)blockIdx.x * n2 * W + 2 * (unit * 16 + lx);
    const int rm_a = 8 * (lx >> 3) + 2 * (lx & 3) + ((lx >> 2) & 1);

    auto chunk_load = [&](int rb, d2 (&raw)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = rb + 8 * q + 2 * lk;
            const double *src = cp + (int64_t)(min(row, n2 - 2) >> 1) * (2 * W);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[q]) : "v"(src) : "memory");
        }
    };
    auto chunk_unpack = [&](int rb, const d2 (&raw)[4], d4 (&reg)[2]) {
        const bool inside = rb + 32 <= n;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            reg[i >> 2][i & 3] = (inside || row < n) ? raw[i >> 1][i & 1] : 0.0;
        }
    };
    auto chunk_store = [&](int rb, const d4 (&reg)[2]) {
        if (rb + 32 <= n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d2 v = {reg[q >> 1][2 * (q & 1)], reg[q >> 1][2 * (q & 1) + 1]};
                *reinterpret_cast<d2 *>(cp + (int64_t)((rb + 8 * q + 2 * lk) >> 1) * (2 * W)) = v;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rb + 8 * (i >> 1) + 2 * lk + (i & 1);
            if (row < n) cp[(int64_t)(row >> 1) * (2 * W) + (row & 1)] = reg[i >> 2][i & 3];
        }
    };
    // one 96-row block on three chunk sets (image rows 32 w + ... of buffer `buf` <-> set cw_w)
    auto block = [&](int buf, d4 (&c0)[2], d4 (&c1)[2], d4 (&c2)[2]) {
        d4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
        const int o0 = 16 * (lk & 1), o1 = 16 - o0;
        auto ypart = [&](int w, const d4 (&cw)[2]) {
            const double *up = ul + buf * QR_BLK + (32 * w + 2 * lk) * QB_G + lx;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int q = 8 * (ks >> 1) + (ks & 1);
                const double b = cw[ks >> 2][ks & 3];
                y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o0], b, y0, 0, 0, 0);
                if (!qr_u_tile_zero(w, ks)) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(up[q * QB_G + o1], b, y1, 0, 0, 0);
            }
        };
        ypart(0, c0);
        ypart(1, c1);
        ypart(2, c2);
        double yn[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            yn[r] = -y0[r];
            yn[4 + r] = -y1[r];
        }
        auto upd = [&](int w, d4 (&cw)[2]) {
            const double *vp = vl + buf * QR_BLK + (32 * w + rm_a) * QB_G;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int m = (4 * ks + lk) ^ qr_v_swz(rm_a);
                if (!qr_v_tile_zero(w, 0, ks)) cw[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[m], yn[ks], cw[0], 0, 0, 0);
                if (!qr_v_tile_zero(w, 1, ks)) cw[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(vp[16 * QB_G + m], yn[ks], cw[1], 0, 0, 0);
            }
        };
        upd(0, c0);
        upd(1, c1);
        upd(2, c2);
    };

    d4 S[4][2];
    d2 pf[2][4];
    bool stored8 = false;
    int sl = 0, nku = 0, nkl = 0;
    // L2 warm-up of the images two steps ahead.  Every block interval is gated by the LDS-DMA copies of the next block's images,
    // which all workgroups issue at the same moment for the same, never-touched lines: an HBM round trip (~2.6 us: the interval
    // measured without any arithmetic).  Wave 0 of every workgroup therefore touches a 1 / 32 slice (the workgroups of one XCD
    // share an L2 and run in step) of the images of step k + 2 -- five 1 KB LDS-DMA loads into a dummy KB, no register written,
    // nobody waits for them; they only count in this wave's memory counter (13 instead of 8 younger operations at the wait).
    const bool l2warm = unit == 0 && !(P.skip & 4096);
    constexpr int NWARM = W > 64 ? 5 : 3;                      // 144 (96 without the third image) KB-sized pieces over 32 workgroups
    constexpr int NPIECE = W > 64 ? 72 : 48;                   // pieces per group
    int cur_gu = -1, cur_gl = -1;
    auto warm = [&](int k2) {
        const int j = (int)(blockIdx.x >> 3) & 31;
        const unsigned dst = (unsigned)(uintptr_t)(uml + 2 * QR_BLK);
#pragma unroll
        for (int i = 0; i < NWARM; ++i) {
            const int c = (j + 32 * i) % (2 * NPIECE);         // 2 groups x (48 KB of V | U + 24 KB of the third image)
            const bool up = c < NPIECE && cur_gu >= 0 && nku > 0;
            const int cc = c < NPIECE ? c : c - NPIECE;
            const int grp = up ? cur_gu : cur_gl;
            const int kk = min(k2, (up ? nku : nkl) - 1);
            const int64_t blkid = (int64_t)(grp - P.g_lo) * P.ks + kk;
            const char *src = (cc < 48 ? reinterpret_cast<const char *>(P.vu + blkid * (2 * QR_BLK)) + cc * 1024
                                       : reinterpret_cast<const char *>(P.um + blkid * QR_BLK) + (cc - 48) * 1024) + lane * 16;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst))
                         : "memory");
        }
    };
    auto wait_pf = [&](bool eight_behind) {
        if (eight_behind && l2warm && W > 64)
            asm volatile("s_waitcnt vmcnt(13)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]),
                         "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]) : : "memory");
        else if (eight_behind && l2warm)
            asm volatile("s_waitcnt vmcnt(11)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]),
                         "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]) : : "memory");
        else if (eight_behind)
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]),
                         "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[0][2]), "+v"(pf[0][3]), "+v"(pf[1][0]),
                         "+v"(pf[1][1]), "+v"(pf[1][2]), "+v"(pf[1][3]) : : "memory");
    };
    // one step of a pass: PH = k mod 2 (compile time), the set at window position p is S[(p + 2 PH) % 4]
    auto step = [&](auto phc, int k) {
        constexpr int PH = decltype(phc)::value;
        d4 (&a0)[2] = S[(0 + 2 * PH) % 4];
        d4 (&a1)[2] = S[(1 + 2 * PH) % 4];
        d4 (&a2)[2] = S[(2 + 2 * PH) % 4];
        d4 (&a3)[2] = S[(3 + 2 * PH) % 4];
        const int wb = sl + k * QB_SB;                         // first row of the 128-row window
        const bool has_next = k + 1 < nkl;
        if (k > 0) {
            wait_pf(stored8);
            chunk_unpack(wb + 64, pf[0], a2);
            chunk_unpack(wb + 96, pf[1], a3);
        }
        if (has_next && !(P.skip & 18)) {
            chunk_load(wb + 128, pf[0]);
            chunk_load(wb + 160, pf[1]);
            if (l2warm) warm(k + 2);
        }
        if (k < nku && !(P.skip & 1)) block(0, a1, a2, a3);    // block (g, k): rows wb + 32 ...
        qr_lds_barrier();                                      // B1(k)
        if (!(P.skip & 1)) block(1, a0, a1, a2);               // block (g - 1, k): rows wb ...
        qr_lds_barrier();                                      // B2(k)
        stored8 = false;
        if (!(P.skip & 10)) {
            chunk_store(wb, a0);
            chunk_store(wb + 32, a1);
            stored8 = wb + 64 <= n;
            if (!has_next) {
                chunk_store(wb + 64, a2);
                chunk_store(wb + 96, a3);
            }
        }
    };
    for (int pass = 0; pass < npass; ++pass) {
        int gu, gl;
        pass_groups(pass, gu, gl);
        nku = steps_of(gu);
        nkl = steps_of(gl);
        if (nkl == 0) continue;
        cur_gu = gu;
        cur_gl = gl;
        sl = gl * QB_G;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // G0: the previous pass' stores are done, LDS is free
        chunk_load(sl, pf[0]);
        chunk_load(sl + 32, pf[1]);
        wait_pf(false);
        chunk_unpack(sl, pf[0], S[0]);
        chunk_unpack(sl + 32, pf[1], S[1]);
        chunk_load(sl + 64, pf[0]);
        chunk_load(sl + 96, pf[1]);
        wait_pf(false);
        chunk_unpack(sl + 64, pf[0], S[2]);
        chunk_unpack(sl + 96, pf[1], S[3]);
        stored8 = false;
        qr_lds_barrier();                                      // G1: the images of the first two blocks are in place
        int k = 0;
        for (; k + 2 <= nkl; k += 2) {
            step(std::integral_constant<int, 0>{}, k);
            step(std::integral_constant<int, 1>{}, k + 1);
        }
        if (k < nkl) step(std::integral_constant<int, 0>{}, k);
    }
}

// C (n x ncols, column-major, ld = n) <-> slab layout [slab][row][w] (w = slab width in columns; columns past ncols are
// zero on the way in and dropped on the way out).  One workgroup = 64 rows of one slab through LDS: both sides coalesced.
template <bool TO_SLAB>
__global__ __launch_bounds__(256) void sbback_slab_kernel(double *__restrict__ c, double *__restrict__ ct, int n, int ncols, int w,
                                                          int units) {
    extern __shared__ double slab_tile[];                      // [64][w + 1]
    const int r0 = blockIdx.x * 64, slab = blockIdx.y;
    const int t = threadIdx.x;
    const int64_t base = (int64_t)slab * ((n + 1) & ~1) * w;
    // columns of this slab: units [ub, ub + nb) of C, the rest of the w columns is padding
    const int ub = (int)((int64_t)slab * units / gridDim.y);
    const int wv = ((int)((int64_t)(slab + 1) * units / gridDim.y) - ub) * 16;
    const int cbase = ub * 16;
    if (TO_SLAB) {
        for (int e = t; e < 64 * w; e += 256) {
            const int cc = e >> 6, r = e & 63;
            const int col = cbase + cc, row = r0 + r;
            slab_tile[r * (w + 1) + cc] = (cc < wv && col < ncols && row < n) ? c[(int64_t)col * n + row] : 0.0;
        }
        __syncthreads();
        for (int e = t; e < 64 * w; e += 256) {
            const int r = e / w, cc = e % w;
            if (r0 + r < n) ct[base + qr_slab_at(r0 + r, cc, w)] = slab_tile[r * (w + 1) + cc];
        }
    } else {
        for (int e = t; e < 64 * w; e += 256) {
            const int r = e / w, cc = e % w;
            slab_tile[r * (w + 1) + cc] = (r0 + r < n) ? ct[base + qr_slab_at(r0 + r, cc, w)] : 0.0;
        }
        __syncthreads();
        for (int e = t; e < 64 * w; e += 256) {
            const int cc = e >> 6, r = e & 63;
            const int col = cbase + cc, row = r0 + r;
            if (cc < wv && col < ncols && row < n) c[(int64_t)col * n + row] = slab_tile[r * (w + 1) + cc];
        }
    }
}

// groups per launch of the register form: the V / U images of a launch stay below ~12 GB (one launch up to n ~ 22000)
static bool qr_bal(int ncols);
static int qr_groups_per_launch(int n, int ks) {
    const int ngroups = (n - 2 + QB_G - 1) / QB_G;
    // balanced form: three images per block, and the whole workspace (images + slabs) stays below 6 GB -- a hipMalloc of 17 GB
    // did not hide behind the band reduction any more (366 -> 985 ms until the stage's synchronisation); a launch boundary costs
    // the drain and refill of one block interval
    const bool bal = qr_bal(n);
    const double per_group = (double)ks * (bal ? 3 : 2) * QR_BLK * sizeof(double);
    int g = (int)(((bal && n <= 20480) ? 2.5e9 : 12.0e9) / per_group);
    if (getenv("JXGPU_SBBACK_GROUPS") && atoi(getenv("JXGPU_SBBACK_GROUPS")) > 0) g = atoi(getenv("JXGPU_SBBACK_GROUPS"));
    if (g < 1) g = 1;
    return g < ngroups ? g : ngroups;
}

static int device_cus();

// Slabs of the register form for `ncols` columns: G workgroups of at most NU <= 5 units (16 columns) each, units dealt
// evenly (a slab holds NU or NU - 1).  Up to 5 units per CU: one round with NU = ceil(units / CUs).  Beyond: R = ceil(units /
// (5 CUs)) rounds of CUs workgroups, so that every round is full and as short as its widest slab (n = 50 000: 3125 units ->
// 768 slabs of 4 or 5 units instead of 625 of 5 = 2.44 rounds each as long as a full one).
// one wave per unit (sbback_apply_solo_kernel) from eight units per CU (n >= 32768 on 256 CUs); JXGPU_SBBACK_SOLO=0 / 1
// switches it off / on from four units per CU
static bool qr_solo(int ncols) {
    const int units = (ncols + 15) / 16;
    static const int env = getenv("JXGPU_SBBACK_SOLO") ? atoi(getenv("JXGPU_SBBACK_SOLO")) : -1;
    if (env == 0) return false;
    return units >= (env == 1 ? 4 : 8) * device_cus();
}
constexpr int QR_SOLO_MAX = 11;           // units per workgroup of the one-group solo form: 12 waves = 3 per SIMD (168 - 179 registers)
constexpr int QR_PAIR_MAX = 7;            // of the two-group form: 8 waves = 2 per SIMD (four chunk sets: ~200 registers)
// two sweep groups per pass (sbback_apply_pair_kernel) whenever the solo form applies; JXGPU_SBBACK_PAIR=0: one group per pass
static bool qr_pair() {
    static const int env = getenv("JXGPU_SBBACK_PAIR") ? atoi(getenv("JXGPU_SBBACK_PAIR")) : 1;
    return env != 0;
}

// balanced five-units-per-CU form (sbback_apply_bal_kernel): more than four and at most five units per CU; JXGPU_SBBACK_BAL5=0: off
static bool qr_bal(int ncols) {
    const int units = (ncols + 15) / 16;
    static const int env = getenv("JXGPU_SBBACK_BAL5") ? atoi(getenv("JXGPU_SBBACK_BAL5")) : 1;
    static const int lo = getenv("JXGPU_SBBACK_BAL_MIN") ? atoi(getenv("JXGPU_SBBACK_BAL_MIN")) : 2;   // more than `lo` units per CU (measured: 84.8 against 89.3 ms at n = 10 000, slower at 8000)
    // one round of slabs only (at most seven units per CU, n <= 28 672): with two rounds at n = 50 000 the four-column waves carry
    // 2 - 3 of a slab's 6 - 7 units through their 8-byte row stores and the stage is as much HBM- as pipe-bound there -- 7.71 s
    // against 6.82 s for the two-groups-per-pass form with one wave per unit
    constexpr int hi = 7;                                        // at most seven units per CU
    return env != 0 && units > lo * device_cus() && units <= hi * device_cus();
}
// slabs of the balanced form: R = ceil(units / (7 CUs)) full rounds of CUs workgroups with the units dealt evenly, 4 .. 7 units
// per slab (four 16-column waves + the four-column waves on the units beyond the fourth); below four units per CU whole slabs of four
static void qr_bal_plan(int ncols, int *g, int *per) {
    const int units = (ncols + 15) / 16, cus = device_cus();
    int gg = ((units + 7 * cus - 1) / (7 * cus)) * cus;
    int pp = (units + gg - 1) / gg;
    if (pp < 4) {
        pp = 4;
        gg = (units + 3) / 4;
    }
    static const int force = getenv("JXGPU_SBBACK_BAL_PER") ? atoi(getenv("JXGPU_SBBACK_BAL_PER")) : 0;   // diagnostic: units per slab
    if (force >= 4 && force <= 7) {
        pp = force;
        gg = (units + pp - 1) / pp;
    }
    *g = gg;
    *per = pp;
}
static int qr_bal_nu(int ncols) {
    int g, per;
    qr_bal_plan(ncols, &g, &per);
    return per;
}

static void qr_plan(int ncols, int *g_out, int *nu_out) {
    const int units = (ncols + 15) / 16;
    const int cus = device_cus();
    int nu, g;
    if (qr_bal(ncols)) {
        qr_bal_plan(ncols, g_out, nu_out);
        return;
    }
    if (qr_solo(ncols)) {
        // R = ceil(units / (15 CUs)) full rounds of CUs workgroups, units dealt evenly (n = 50 000: 256 slabs of 12 or 13)
        const int wmax = qr_pair() ? QR_PAIR_MAX : QR_SOLO_MAX;
        const int rounds = (units + wmax * cus - 1) / (wmax * cus);
        g = rounds * cus;
        nu = (units + g - 1) / g;
        *g_out = std::max(g, 1);
        *nu_out = std::max(nu, 4);
        return;
    }
    if (getenv("JXGPU_SBBACK_NW") && atoi(getenv("JXGPU_SBBACK_NW")) > 0) {       // fixed width, whole slabs (diagnostic)
        nu = std::min(atoi(getenv("JXGPU_SBBACK_NW")), 5);
        g = (units + nu - 1) / nu;
    } else if (units <= 5 * cus) {
        nu = std::max((units + cus - 1) / cus, 1);
        g = (units + nu - 1) / nu;
    } else {
        const int rounds = (units + 5 * cus - 1) / (5 * cus);
        g = rounds * cus;
        nu = (units + g - 1) / g;
    }
    *g_out = std::max(g, 1);
    *nu_out = nu;
}

size_t sbback_tq_doubles(int n, int ks) {
    // V / U images of one launch | C in slab layout (G slabs of NU x 16 columns, padding included)
    int g, nu;
    qr_plan(n, &g, &nu);
    const size_t padded = std::max((size_t)g * nu * 16, (size_t)n + 96);
    // (+ the third image per block of the balanced form, behind the slabs)
    return (size_t)qr_groups_per_launch(n, ks) * ks * (2 + (qr_bal(n) ? 1 : 0)) * QR_BLK + padded * (size_t)(n + 1);
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
extern float g_last_ms[24];   // [16] apply launches of the last back-transformation, [17] its form: 0 three waves per unit, 1 one
                              // wave per unit, 2 one wave per unit and two groups per pass, 3 balanced (five / four units per CU)
static int sbback_apply_q2_reg(hipStream_t st, const double *d_v2, const double *d_tau2, int n, int ks, double *d_c, int ncols,
                               double *d_vu, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int ngroups = (n - 2 + QB_G - 1) / QB_G;
    const int gpl = qr_groups_per_launch(n, ks);
    int units = ceil_div(ncols, 16);
    int nu, gslabs;
    qr_plan(ncols, &gslabs, &nu);
    if (getenv("JXGPU_SBBACK_BAL") && atoi(getenv("JXGPU_SBBACK_BAL")) == 0) {   // diagnostic: whole slabs of 5 units at most
        nu = std::min(std::max(ceil_div(units, device_cus()), 1), 5);
        gslabs = ceil_div(units, nu);
        units = gslabs * nu;
    }
    {
        int gn, nun;                                           // the workspace was sized for ncols = n (sbback_tq_doubles)
        qr_plan(n, &gn, &nun);
        if ((size_t)gslabs * nu * 16 > std::max((size_t)gn * nun * 16, (size_t)n + 96))
            return fail("sbback_apply_q2: slab plan exceeds the workspace");
    }
    const bool bal = qr_bal(ncols) && qr_bal(n) && nu == qr_bal_nu(ncols) && !(getenv("JXGPU_SBBACK_BAL") && atoi(getenv("JXGPU_SBBACK_BAL")) == 0);
    const bool solo = !bal && qr_solo(ncols) && !(getenv("JXGPU_SBBACK_BAL") && atoi(getenv("JXGPU_SBBACK_BAL")) == 0);
    const size_t lds = bal ? sizeof(double) * (6 * (size_t)QR_BLK + 128)
                           : (solo ? sizeof(double) * 4 * (size_t)QR_BLK : sizeof(double) * (4 * (size_t)QR_BLK + (size_t)nu * 3 * 8 * 64));
    const dim3 grid(gslabs);
    const int skip = getenv("JXGPU_QB_SKIP") ? atoi(getenv("JXGPU_QB_SKIP")) : 0;
    const int w = nu * 16;
    double *d_ct = d_vu + (size_t)gpl * ks * 2 * QR_BLK;
    double *d_um = nullptr;                                    // third image per block of the balanced form: behind the slabs
    {
        int gn, nun;
        qr_plan(n, &gn, &nun);
        d_um = d_ct + std::max((size_t)gn * nun * 16, (size_t)n + 96) * (size_t)(n + 1);
    }
    const dim3 sgrid(ceil_div(n, 64), grid.x);
    const size_t slds = sizeof(double) * 64 * (w + 1);
    {
        static bool slab_attr = false;                         // slabs of the solo form are up to 240 columns wide: 123 KB tiles
        if (!slab_attr) {
            JX_HIP(hipFuncSetAttribute((const void *)sbback_slab_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(double) * 64 * (QR_SOLO_MAX * 16 + 1))));
            JX_HIP(hipFuncSetAttribute((const void *)sbback_slab_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(double) * 64 * (QR_SOLO_MAX * 16 + 1))));
            slab_attr = true;
        }
    }
    hipLaunchKernelGGL(sbback_slab_kernel<true>, sgrid, dim3(256), slds, st, d_c, d_ct, n, ncols, w, units);
    JX_LAUNCH_CHECK();
    g_last_ms[16] = (float)ceil_div(ngroups, gpl);
    g_last_ms[17] = bal ? 3.f : (solo ? (qr_pair() ? 2.f : 1.f) : 0.f);
    for (int g_hi = ngroups; g_hi > 0; g_hi -= gpl) {
        const int g_lo = g_hi > gpl ? g_hi - gpl : 0;
        QrParams P{d_v2, d_tau2, d_vu, d_ct, n, ks, ncols, units, g_lo, g_hi, skip, (bal && nu >= 5) ? d_um : nullptr};
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
#define JX_QR_SOLO(NWV)                                                                                                \
    do {                                                                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            JX_HIP(hipFuncSetAttribute((const void *)sbback_apply_solo_kernel<NWV>,                                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipExtLaunchKernelGGL(sbback_apply_solo_kernel<NWV>, grid, dim3((NWV + 1) * 64), lds, st, e0, e1, 0, P);        \
    } while (0)
        if (bal) {
            static bool attr_set = false;
            if (!attr_set) {
                JX_HIP(hipFuncSetAttribute((const void *)sbback_apply_bal_kernel<80>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                JX_HIP(hipFuncSetAttribute((const void *)sbback_apply_bal_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                JX_HIP(hipFuncSetAttribute((const void *)sbback_apply_bal_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                JX_HIP(hipFuncSetAttribute((const void *)sbback_apply_bal_kernel<112>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_set = true;
            }
            switch (nu) {
                case 4: hipExtLaunchKernelGGL(sbback_apply_bal_kernel<64>, grid, dim3(512), lds, st, e0, e1, 0, P); break;
                case 5: hipExtLaunchKernelGGL(sbback_apply_bal_kernel<80>, grid, dim3(512), lds, st, e0, e1, 0, P); break;
                case 6: hipExtLaunchKernelGGL(sbback_apply_bal_kernel<96>, grid, dim3(512), lds, st, e0, e1, 0, P); break;
                case 7: hipExtLaunchKernelGGL(sbback_apply_bal_kernel<112>, grid, dim3(512), lds, st, e0, e1, 0, P); break;
                default: return fail("sbback_apply_q2: balanced slab width out of range");
            }
        } else if (solo && qr_pair()) {
#define JX_QR_PAIR(NWV)                                                                                                \
    do {                                                                                                               \
        static bool attr_set = false;                                                                                  \
        if (!attr_set) {                                                                                               \
            JX_HIP(hipFuncSetAttribute((const void *)sbback_apply_pair_kernel<NWV>,                                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipExtLaunchKernelGGL(sbback_apply_pair_kernel<NWV>, grid, dim3((NWV + 1) * 64), lds, st, e0, e1, 0, P);        \
    } while (0)
            switch (nu) {
                case 4: JX_QR_PAIR(4); break;
                case 5: JX_QR_PAIR(5); break;
                case 6: JX_QR_PAIR(6); break;
                case 7: JX_QR_PAIR(7); break;
                default: return fail("sbback_apply_q2: pair slab width out of range");
            }
#undef JX_QR_PAIR
        } else if (solo) {
            switch (nu) {
                case 4: JX_QR_SOLO(4); break;
                case 5: JX_QR_SOLO(5); break;
                case 6: JX_QR_SOLO(6); break;
                case 7: JX_QR_SOLO(7); break;
                case 8: JX_QR_SOLO(8); break;
                case 9: JX_QR_SOLO(9); break;
                case 10: JX_QR_SOLO(10); break;
                case 11: JX_QR_SOLO(11); break;
                default: return fail("sbback_apply_q2: solo slab width out of range");
            }
        } else {
            switch (nu) {
                case 1: JX_QR_LAUNCH(1); break;
                case 2: JX_QR_LAUNCH(2); break;
                case 3: JX_QR_LAUNCH(3); break;
                case 4: JX_QR_LAUNCH(4); break;
                default: JX_QR_LAUNCH(5); break;
            }
        }
#undef JX_QR_SOLO
#undef JX_QR_LAUNCH
        JX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sbback_slab_kernel<false>, sgrid, dim3(256), slds, st, d_c, d_ct, n, ncols, w, units);
    JX_LAUNCH_CHECK();
    return 0;
}

// C (n x ncols, ld = n) <- Q2 C.  d_tq: sbback_tq_doubles(n, ks) doubles of workspace.
int sbback_apply_q2(hipStream_t st, const double *d_v2, const double *d_tau2, int n, int ks, double *d_c, int ncols,
                    double *d_tq, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (n <= 2 || ncols <= 0) return 0;
    return sbback_apply_q2_reg(st, d_v2, d_tau2, n, ks, d_c, ncols, d_tq, ev_start, ev_stop);
}

}  // namespace jx
