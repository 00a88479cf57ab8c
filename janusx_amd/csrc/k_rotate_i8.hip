// Eigenvector rotation G~ = G U for EXACT design rows (allele counts {0,1,2}, no missing call) on the int8 matrix pipes.
//
// Reference: rotate_snp_block_with_ut_blas (src/stats/lmm.rs:728-784: an f32 SGEMM of the decoded design rows with U^T),
// design decode decode_centered_block_packed_f32 (src/decode/decode.rs:192-271).
//
// A design row without missing calls is  g_r = beta_r + s_r c_r  (c_r = counts of the payload's second allele, s_r = -1 when
// the scan LUT is flipped), so  g_r U = s_r (c_r U) + (beta_r + 2 [s_r < 0]) usum,  usum_j = sum_i U_ij  (the offset term as in
// rotate_f16x2_kernel).  The product c U has an exact small-integer operand; U is quantised once per model into THREE int8
// planes with one scale per eigenvector,
//     U_ij = umax_j ( q1 / 127 + q2 / (127 254) + q3 / (127 254^2) ) + O(umax_j 2^-24),   |q| <= 127,
// and  c U  = umax_j ( (c q1) / 127 + (c q2) / (127 254) + (c q3) / (127 254^2) )  with three v_mfma_i32_32x32x32_i8 products
// whose i32 sums are exact (|c q| summed over n <= 66 000 samples stays below 2^24); the planes are combined in f64 in the
// epilogue.  Measured on a family-structured panel (n = 3000) the quantisation error of the rotated values is 1.6e-8 of their
// range at most (fp16 hi/lo split: 1.0e-7; the f32 SGEMM of the reference: 1.8e-6).  Three int8 products cost 0.75 of the two
// fp16 products of the exact-row form of rotate256_kernel, the operand images are bytes (A: 1 B instead of 2 B per element,
// B: 3 B instead of 4 B), and the design decode is pure VALU (no LUT, no selector table: k_grm_i8.hip).
//
// Shape: 512 threads = 8 waves (2 x 4) on 256 SNP rows x 128 eigenvector columns, 128 x 32 per wave (3 planes x 4 tiles of
// 32 x 32 i32 accumulators = 192 registers); images [row][64 k] bytes with the 16-byte chunk index XOR-swizzled by
// (row >> 2) & 3, double buffered, one barrier per 64-sample step; the k order inside a 16-sample group is the decode's
// (byte 4 q + b = sample 4 b + q) and the planes are stored in that order.  The exact rows of a block are passed as a compacted
// list of positions (`sel`), the other rows go to the fp16 kernel the same way: a row takes the same path, and gets the same
// bits, whatever rows surround it (chunked scans equal unchunked ones, the reference's smoke invariant).
#include <stdlib.h>

#include "jx_common.h"

namespace jx {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));

constexpr int RI_TM = 256;                // SNP rows per workgroup
constexpr int RI_TN = 128;                // eigenvector columns per workgroup
constexpr int RI_BK = 64;                 // samples per step
constexpr int RI_IMG_A = RI_TM * RI_BK;   // bytes
constexpr int RI_IMG_B = RI_TN * RI_BK;   // one plane
constexpr int RI_SET = RI_IMG_A + 3 * RI_IMG_B;

__device__ __forceinline__ int ri_chunk_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// 16 two-bit codes -> 16 count bytes, byte 4 q + b = sample 4 b + q (same as k_grm_i8.hip)
__device__ __forceinline__ u32x4q ri_decode16(uint32_t w) {
    const uint32_t hi = (w >> 1) & 0x55555555u;
    const uint32_t v = hi + (hi & w);
    u32x4q o;
    o.x = v & 0x03030303u;
    o.y = (v >> 2) & 0x03030303u;
    o.z = (v >> 4) & 0x03030303u;
    o.w = (v >> 6) & 0x03030303u;
    return o;
}

// 16 two-bit codes -> 16 bytes that are 1 at a missing call (code 01) and 0 elsewhere, same byte order
__device__ __forceinline__ u32x4q ri_decode16_missing(uint32_t w) {
    const uint32_t v = w & ~(w >> 1) & 0x55555555u;
    u32x4q o;
    o.x = v & 0x03030303u;
    o.y = (v >> 2) & 0x03030303u;
    o.z = (v >> 4) & 0x03030303u;
    o.w = (v >> 6) & 0x03030303u;
    return o;
}

// ---- quantisation of U^T (n x n f32, row j = eigenvector j) into three int8 planes (npad x npad each) ---------------------
__global__ __launch_bounds__(256) void ut_rowmax_kernel(const float *__restrict__ ut, int n, int npad, float *__restrict__ umax) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= npad) return;
    float a = 0.0f;
    if (j < n)
        for (int i = lane; i < n; i += 64) a = fmaxf(a, fabsf(ut[(int64_t)j * n + i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a = fmaxf(a, __shfl_xor(a, off, 64));
    if (lane == 0) umax[j] = (a > 0.0f && a < 3.0e38f) ? a : 1.0f;
}

__global__ __launch_bounds__(256) void ut_quant3_kernel(const float *__restrict__ ut, int n, int64_t npad,
                                                        const float *__restrict__ umax, int8_t *__restrict__ q) {
    // thread = (row j, 16-sample group): writes the group's 16 bytes of every plane in the decode's order
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t groups = npad / 16;
    if (idx >= npad * groups) return;
    const int64_t j = idx / groups, gq = idx - j * groups;
    int8_t o[3][16];
    const double inv = (j < n) ? 1.0 / (double)umax[j] : 0.0;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        const int smp = ((pos & 3) << 2) | (pos >> 2);           // byte 4 q + b holds sample 4 b + q
        const int64_t i = gq * 16 + smp;
        double u = 0.0;
        if (j < n && i < n) u = (double)ut[j * (int64_t)n + i] * inv;
        const double a1 = u * 127.0;
        double q1 = rint(a1);
        const double a2 = (a1 - q1) * 254.0;
        double q2 = rint(a2);
        const double a3 = (a2 - q2) * 254.0;
        double q3 = rint(a3);
        q1 = fmin(fmax(q1, -127.0), 127.0);
        q2 = fmin(fmax(q2, -127.0), 127.0);
        q3 = fmin(fmax(q3, -127.0), 127.0);
        o[0][pos] = (int8_t)q1;
        o[1][pos] = (int8_t)q2;
        o[2][pos] = (int8_t)q3;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<uint4 *>(q + pl * npad * npad + j * npad + gq * 16) = *reinterpret_cast<const uint4 *>(o[pl]);
}

// ---- rotation --------------------------------------------------------------------------------------------------------
// MODE 0: the rotation of exact design rows (above).  MODE 1: the missing-call term of rows that keep the exact path although
// they hold missing calls -- the SAME product with the indicator of the missing calls as the integer operand (e U, exact) and
// the epilogue  out += d_r (e_r U),  d_r = `rowoff[pos]` (the value jxg_lut_split_rows_m leaves in d_rowmiss: imputed value minus
// what the count form puts at a missing call): `lut16` / `usum` are not read.  One more pass of the int8 kernel over those
// rows = the cost of the fp16 hi / lo kernel they took before, but exact (the fp16 rotation of a row with missing calls was
// measurably noisier than the reference's f32 SGEMM: 4.3e-6 against 1.1e-6 on beta at n = 20 000).
template <int MODE>
__global__ __launch_bounds__(512, 2) void rotate_i8_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                           const int32_t *__restrict__ rows, int nrows,
                                                           const uint4 *__restrict__ lut16, const float *__restrict__ rowoff,
                                                           const float *__restrict__ usum, const int8_t *__restrict__ q,
                                                           const float *__restrict__ umax, int64_t npad, int n,
                                                           float *__restrict__ out, int64_t ldo,
                                                           const int32_t *__restrict__ sel) {
    // `sel` (optional): positions of this launch's rows inside the block (rows / lut16 / rowoff / out are indexed by position):
    // the exact rows of a block as a compacted list, so that a row takes the same path whatever rows surround it
    extern __shared__ __attribute__((aligned(16))) uint8_t ri_smem[];       // 2 sets | row offsets | row signs | positions (1 KB each)
    float *sOff = reinterpret_cast<float *>(ri_smem + 2 * RI_SET);
    float *sSgn = sOff + RI_TM;
    int *sPos = reinterpret_cast<int *>(sSgn + RI_TM);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // 1-D grid, XCD-aware: every XCD owns the column tiles ct = 8 g + xcd and walks the row tiles fastest
    const int nrt = (nrows + RI_TM - 1) / RI_TM;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int ct = (local / nrt) * 8 + xcd;
    if ((int64_t)ct * RI_TN >= npad) return;
    const int j0 = ct * RI_TN;
    const int r0 = (local % nrt) * RI_TM;

    // A panel: thread = (row tid >> 1, 16-byte pair of chunks): two payload dwords (32 samples) per step
    const int arow = tid >> 1, ahalf = tid & 1;
    const uint8_t *arec = p32;
    int row_exact = 1;
    {
        const int r = r0 + arow;
        float boff = 0.0f, sgn = 1.0f;
        int pos = -1;
        if (r < nrows) {
            pos = sel ? sel[r] : r;
            const int64_t rec = rows ? (int64_t)rows[pos] : (int64_t)pos;
            arec = p32 + rec * 32;
            const float t = rowoff[pos];
            if (MODE == 1) {
                boff = t;                                   // d_r
            } else {
                row_exact = (t == t) ? 1 : 0;
                // integer LUT {c(00), 0, c(10), c(11)} = {0, 0, 1, 2} or flipped {2, 0, 1, 0} (fp16 in lut16.x low half)
                const bool flipped = (lut16[pos].x & 0xffffu) != 0u;
                sgn = flipped ? -1.0f : 1.0f;
                boff = row_exact ? (t + (flipped ? 2.0f : 0.0f)) : 0.0f;
            }
        }
        if (ahalf == 0) {
            sOff[arow] = boff;
            sSgn[arow] = sgn;
            sPos[arow] = pos;
        }
    }
    if (__syncthreads_and(row_exact) == 0) return;        // a row that does not qualify: the fp16 kernel takes the tile
    arec += 8 * ahalf;                                     // this thread's 8 payload bytes of a 16-byte (64-sample) piece

    // B planes: thread = (row tid >> 2, 16-byte chunk tid & 3) of each of the three planes
    const int brow = tid >> 2, bpart = tid & 3;
    const bool bok = j0 + brow < npad;
    const int8_t *bsrc = q + (int64_t)(bok ? j0 + brow : 0) * npad + bpart * 16;
    const uint32_t bmask = bok ? 0xffffffffu : 0u;
    const int64_t plane = npad * npad;
    const int boff_lds = ri_chunk_off(brow, bpart);

    uint2 wa = make_uint2(0, 0);
    u32x4q b0, b1, b2;
    auto load_a = [&](int kstep) {        // samples 64 kstep .. + 63: 16 payload bytes of the row, 8 per thread
        const int tile = kstep >> 1, sub = kstep & 1;
        wa = *reinterpret_cast<const uint2 *>(arec + (int64_t)tile * m_total * 32 + sub * 16);
    };
    auto load_b = [&](int64_t kcol) {
        b0 = *reinterpret_cast<const u32x4q *>(bsrc + kcol);
        b1 = *reinterpret_cast<const u32x4q *>(bsrc + plane + kcol);
        b2 = *reinterpret_cast<const u32x4q *>(bsrc + 2 * plane + kcol);
    };
    auto stage = [&](uint8_t *set) {
        *reinterpret_cast<u32x4q *>(set + ri_chunk_off(arow, 2 * ahalf)) = MODE == 1 ? ri_decode16_missing(wa.x) : ri_decode16(wa.x);
        *reinterpret_cast<u32x4q *>(set + ri_chunk_off(arow, 2 * ahalf + 1)) = MODE == 1 ? ri_decode16_missing(wa.y) : ri_decode16(wa.y);
        const u32x4q mk = {bmask, bmask, bmask, bmask};
        *reinterpret_cast<u32x4q *>(set + RI_IMG_A + boff_lds) = b0 & mk;
        *reinterpret_cast<u32x4q *>(set + RI_IMG_A + RI_IMG_B + boff_lds) = b1 & mk;
        *reinterpret_cast<u32x4q *>(set + RI_IMG_A + 2 * RI_IMG_B + boff_lds) = b2 & mk;
    };

    i32x16 acc[3][4];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pl][mi][r] = 0;

    const int h = lane >> 5, frow = lane & 31;
    auto mfma_ks = [&](const uint8_t *set, int ks) {
        // two A fragments at a time (the accumulators take 192 of the 256 registers)
        i32x4 b[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            b[pl] = *reinterpret_cast<const i32x4 *>(set + RI_IMG_A + pl * RI_IMG_B + ri_chunk_off(wn * 32 + frow, 2 * ks + h));
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
            i32x4 a[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                a[u] = *reinterpret_cast<const i32x4 *>(set + ri_chunk_off(wm * 128 + (2 * mp + u) * 32 + frow, 2 * ks + h));
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    acc[pl][2 * mp + u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u], b[pl], acc[pl][2 * mp + u], 0, 0, 0);
            if (mp == 0) __builtin_amdgcn_sched_barrier(0);
        }
    };

    const int nk = (int)(npad / RI_BK);
    load_a(0);
    load_b(0);
    stage(ri_smem);
    {
        const int k1 = nk > 1 ? 1 : 0;
        load_a(k1);
        load_b((int64_t)k1 * RI_BK);
    }
    __syncthreads();
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        const uint8_t *rd = ri_smem + cur * RI_SET;
        uint8_t *wr = ri_smem + (cur ^ 1) * RI_SET;
        const int kn = ks + 2 < nk ? ks + 2 : nk - 1;      // clamped: the last iterations re-load the last step, nothing reads it
        mfma_ks(rd, 0);
        stage(wr);                                         // step ks + 1 (in registers since the previous iteration)
        load_a(kn);
        load_b((int64_t)kn * RI_BK);
        __builtin_amdgcn_sched_barrier(0);
        mfma_ks(rd, 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: planes combined in f64; C/D layout of the 32 x 32 shapes: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
    const int gj = j0 + wn * 32 + frow;
    const bool colok = gj < n;
    const double um = colok ? (double)umax[gj] : 0.0;
    const double w1 = um / 127.0, w2 = um / (127.0 * 254.0), w3 = um / (127.0 * 254.0 * 254.0);
    const float us = (MODE == 0 && colok) ? usum[gj] : 0.0f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int gr = sPos[lr];
            if (gr >= 0 && colok) {
                const double v = (double)acc[0][mi][r] * w1 + (double)acc[1][mi][r] * w2 + (double)acc[2][mi][r] * w3;
                float *dst = out + (int64_t)gr * ldo + gj;
                if (MODE == 1) *dst = fmaf(sOff[lr], (float)v, *dst);
                else *dst = fmaf(sOff[lr], us, sSgn[lr] * (float)v);
            }
        }
    }
}

// ---- the same product with the operands copied by the LDS DMA (round 6) -------------------------------------------------------
// rotate_i8_kernel stages the DECODED rows through LDS (16 KB per 64-sample step, written by VALU stores, read back as 4 of the 7
// ds_read_b128 a wave issues per 12 products) and the planes through registers: its LDS port is busy ~0.8 of the time the matrix
// pipes need.  Here a step is one 128-sample payload tile and
//   * the RAW payload records of the 256 rows (32 B each: 8 KB instead of 32 KB decoded) and the three planes (48 KB) are copied
//     global -> LDS by global_load_lds_dwordx4, no register staging, double buffered, ONE barrier per 128 samples;
//   * a lane reads its 16 payload bytes per 32-row fragment ONCE per step (lane (row, h) takes the dwords 4 h .. 4 h + 3 of the
//     record: the k slot (ks, h) of the MFMA is the 16-sample group 4 h + ks of the tile, on both operands) and decodes them in
//     registers in front of the products (11 VALU instructions per fragment, four waves repeat the decode of the rows they share);
//   * per step and wave 4 + 12 ds_read_b128 for 48 products (28 before).
// Plane image in LDS: [column][128 bytes], 16-byte chunk index XOR (column >> 1) & 7 (a 16-lane group of a fragment read covers all
// 64 banks); the DMA writes 1 KB pieces (8 columns) whose lanes pick the source chunk that belongs at their linear position.
// Same exact i32 sums and the same epilogue as rotate_i8_kernel: the outputs are bit-identical.
constexpr int RD_BK = 128;                  // samples per step = one payload tile
constexpr int RD_A = 8 * 1024;              // raw records: piece w = rows 32 w .. + 31, [half][row][16 B]
constexpr int RD_B = RI_TN * RD_BK;         // one plane
constexpr int RD_STAGE = RD_A + 3 * RD_B;   // 56 KB

// 16 bytes per lane from `base` (wave-uniform) + `off` (per lane, 32 bits) to LDS at lds_dst + 16 lane
__device__ __forceinline__ void rd_glds(const void *base, uint32_t off, unsigned lds_dst) {
    unsigned keep;
    const uint64_t b = (uint64_t)(uintptr_t)base;
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    const uint64_t sb = ((uint64_t)bhi << 32) | blo;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(off), "s"(sb), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}

// 16 two-bit codes -> the MFMA operand of one k slot (same byte order as ri_decode16)
template <int MODE> __device__ __forceinline__ i32x4 rd_decode(uint32_t w) {
    const u32x4q o = MODE == 1 ? ri_decode16_missing(w) : ri_decode16(w);
    i32x4 r;
    r.x = (int)o.x;
    r.y = (int)o.y;
    r.z = (int)o.z;
    r.w = (int)o.w;
    return r;
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void rotate_i8_dma_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                               const int32_t *__restrict__ rows, int nrows,
                                                               const uint4 *__restrict__ lut16, const float *__restrict__ rowoff,
                                                               const float *__restrict__ usum, const int8_t *__restrict__ q,
                                                               const float *__restrict__ umax, int64_t npad, int n,
                                                               float *__restrict__ out, int64_t ldo,
                                                               const int32_t *__restrict__ sel) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rd_smem[];       // 2 stages | row offsets | row signs | positions
    float *sOff = reinterpret_cast<float *>(rd_smem + 2 * RD_STAGE);
    float *sSgn = sOff + RI_TM;
    int *sPos = reinterpret_cast<int *>(sSgn + RI_TM);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nrt = (nrows + RI_TM - 1) / RI_TM;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int ct = (local / nrt) * 8 + xcd;
    if ((int64_t)ct * RI_TN >= npad) return;
    const int j0 = ct * RI_TN;
    const int r0 = (local % nrt) * RI_TM;

    // payload piece of this wave: rows 32 wave .. + 31, lane = (row lane & 31, record half lane >> 5); byte offsets of a lane's
    // sources are 32-bit (m_total < 2^27: the launcher checks), the bases wave-uniform
    uint32_t aoff = 0;
    int row_exact = 1;
    {
        const int arow = 32 * wave + (lane & 31);
        const int r = r0 + arow;
        float boff = 0.0f, sgn = 1.0f;
        int pos = -1;
        if (r < nrows) {
            pos = sel ? sel[r] : r;
            aoff = (uint32_t)(rows ? rows[pos] : pos) * 32u;
            const float t = rowoff[pos];
            if (MODE == 1) {
                boff = t;
            } else {
                row_exact = (t == t) ? 1 : 0;
                const bool flipped = (lut16[pos].x & 0xffffu) != 0u;
                sgn = flipped ? -1.0f : 1.0f;
                boff = row_exact ? (t + (flipped ? 2.0f : 0.0f)) : 0.0f;
            }
        }
        if (lane < 32) {
            sOff[arow] = boff;
            sSgn[arow] = sgn;
            sPos[arow] = pos;
        }
    }
    if (__syncthreads_and(row_exact) == 0) return;
    aoff += 16u * (uint32_t)(lane >> 5);

    // plane pieces of this wave: the column groups 2 wave + e (8 columns each), e = 0, 1, of every plane
    uint32_t boffs[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int col = 16 * wave + 8 * e + (lane >> 3);
        const int c = (lane & 7) ^ ((col >> 1) & 7);
        boffs[e] = (uint32_t)col * (uint32_t)npad + (uint32_t)c * 16u;
    }
    const int8_t *qtile = q + (int64_t)j0 * npad;
    const int64_t plane = npad * npad;
    const int64_t atile = m_total * 32;
    const unsigned lds0 = (unsigned)(uintptr_t)rd_smem;
    auto issue = [&](int t, int buf) {
        const unsigned st = lds0 + buf * RD_STAGE;
        rd_glds(p32 + (int64_t)t * atile, aoff, st + wave * 1024);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                rd_glds(qtile + pl * plane + (int64_t)t * RD_BK, boffs[e], st + RD_A + pl * RD_B + (2 * wave + e) * 1024);
    };

    i32x16 acc[3][4];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pl][mi][r] = 0;

    const int h = lane >> 5, frow = lane & 31;
    const int afrag = wm * 4 * 1024 + h * 512 + frow * 16;                 // + mi * 1024
    const int bcol = (wn * 32 + frow) * RD_BK;
    const int bx0 = (4 * h) ^ ((frow >> 1) & 7);                             // chunk of k slot (ks, h): (bx0 ^ ks)

    const int nk = (int)(npad / RD_BK);
    issue(0, 0);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // this wave's pieces of step t landed
        __builtin_amdgcn_s_barrier();                                          // everybody's did, and nobody reads the other stage any more
        asm volatile("" ::: "memory");
        issue(t + 1 < nk ? t + 1 : nk - 1, buf ^ 1);                           // past the end: a copy nothing reads
        const uint8_t *st = rd_smem + buf * RD_STAGE;
        i32x4 araw[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) araw[mi] = *reinterpret_cast<const i32x4 *>(st + afrag + mi * 1024);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            i32x4 b[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                b[pl] = *reinterpret_cast<const i32x4 *>(st + RD_A + pl * RD_B + bcol + ((bx0 ^ ks) << 4));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const i32x4 a = rd_decode<MODE>((uint32_t)araw[mi][ks]);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    acc[pl][mi] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[pl], acc[pl][mi], 0, 0, 0);
            }
        }
        buf ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // no DMA may outlive the workgroup's LDS allocation

    const int gj = j0 + wn * 32 + frow;
    const bool colok = gj < n;
    const double um = colok ? (double)umax[gj] : 0.0;
    const double w1 = um / 127.0, w2 = um / (127.0 * 254.0), w3 = um / (127.0 * 254.0 * 254.0);
    const float us = (MODE == 0 && colok) ? usum[gj] : 0.0f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int gr = sPos[lr];
            if (gr >= 0 && colok) {
                const double v = (double)acc[0][mi][r] * w1 + (double)acc[1][mi][r] * w2 + (double)acc[2][mi][r] * w3;
                float *dst = out + (int64_t)gr * ldo + gj;
                if (MODE == 1) *dst = fmaf(sOff[lr], (float)v, *dst);
                else *dst = fmaf(sOff[lr], us, sSgn[lr] * (float)v);
            }
        }
    }
}

// 1 (default): the DMA form; JXGPU_ROT_I8_DMA=0: the register-staged form (same bits)
static bool rot_i8_dma() {
    static const bool on = !(getenv("JXGPU_ROT_I8_DMA") && atoi(getenv("JXGPU_ROT_I8_DMA")) == 0);
    return on;
}

extern float g_last_ms[24];

}  // namespace jx

using namespace jx;

namespace jx {
int launch_rotate_i8_missing(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const int32_t *d_sel,
                             int nsel, const float *d_rowmiss, const int8_t *d_q, const float *d_umax, float *d_out, int64_t ld_out);
}
// Missing-call term of design rows that keep the exact (int8) rotation, as ONE MORE int8 product (indicator of the missing calls
// x the three planes of U) instead of a gather per missing call: d_out[d_sel[i]] += d_rowmiss[d_sel[i]] * (e U) for the nsel
// rows listed in d_sel (positions inside the block d_rows / d_rowmiss / d_out are indexed by).  Used when the rows of a scan
// hold more than n / 300 missing calls on average (jxg_rot_miss_max > 256), where the gather form (jxg_rotate_missing_correct)
// costs more than the product.
extern "C" int jxg_rotate_missing_dense(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const int32_t *d_sel,
                                        int nsel, const float *d_rowmiss, const int8_t *d_q, const float *d_umax, float *d_out,
                                        int64_t ld_out, void *stream) {
    return launch_rotate_i8_missing((hipStream_t)stream, d_p32, m_total, n, d_rows, d_sel, nsel, d_rowmiss, d_q, d_umax, d_out, ld_out);
}

// U^T (n x n f32, row j = eigenvector j) -> three int8 planes (npad x npad each, npad = 128 ceil(n / 128); plane p at
// d_q + p npad^2; k order inside 16-sample groups as decoded by the rotation kernel) + umax (npad) f32
extern "C" int jxg_ut_quant3(const float *d_ut, int n, int8_t *d_q, float *d_umax, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int64_t npad = (int64_t)num_tiles(n) * JXG_TILE;
    hipLaunchKernelGGL(ut_rowmax_kernel, dim3((unsigned)((npad + 3) / 4)), dim3(256), 0, st, d_ut, n, (int)npad, d_umax);
    JX_LAUNCH_CHECK();
    const int64_t items = npad * (npad / 16);
    if ((items + 255) / 256 > 0x7fffffffLL) return fail("jxg_ut_quant3: grid too large");
    hipLaunchKernelGGL(ut_quant3_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, d_ut, n, npad, d_umax, d_q);
    JX_LAUNCH_CHECK();
    return 0;
}

namespace jx {
// int8 kernel over the `nsel` rows d_sel (positions inside the block; every one of them must be an exact row)
int launch_rotate_i8(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const int32_t *d_sel,
                     int nsel, const void *d_lut16, const float *d_rowoff, const float *d_usum, const int8_t *d_q,
                     const float *d_umax, float *d_out, int64_t ld_out) {
    if (nsel <= 0) return 0;
    const int nt = num_tiles(n);
    const int64_t npad = (int64_t)nt * JXG_TILE;
    if (!d_q || !d_umax || !d_rowoff || !d_usum || n > 66000) return fail("launch_rotate_i8: planes missing or n > 66000");
    const int nrows = nsel;
    const int nct = (int)((npad + RI_TN - 1) / RI_TN), nrt = (nrows + RI_TM - 1) / RI_TM;
    static bool attr = false;
    const int lds = 2 * RI_SET + 3072;
    if (!attr) {
        JX_HIP(hipFuncSetAttribute((const void *)rotate_i8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        JX_HIP(hipFuncSetAttribute((const void *)rotate_i8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    dim3 grid((unsigned)(((nct + 7) / 8) * 8 * nrt));
    if (rot_i8_dma() && m_total < (1LL << 27)) {          // 32-bit byte offsets of the payload records inside a tile
        static bool attr_dma = false;
        const int lds_dma = 2 * RD_STAGE + 3072;
        if (!attr_dma) {
            JX_HIP(hipFuncSetAttribute((const void *)rotate_i8_dma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_dma));
            attr_dma = true;
        }
        hipLaunchKernelGGL(rotate_i8_dma_kernel<0>, grid, dim3(512), lds_dma, st, d_p32, m_total, d_rows, nrows,
                           (const uint4 *)d_lut16, d_rowoff, d_usum, d_q, d_umax, npad, n, d_out, ld_out, d_sel);
        JX_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(rotate_i8_kernel<0>, grid, dim3(512), lds, st, d_p32, m_total, d_rows, nrows, (const uint4 *)d_lut16,
                       d_rowoff, d_usum, d_q, d_umax, npad, n, d_out, ld_out, d_sel);
    JX_LAUNCH_CHECK();
    return 0;
}

// the missing-call term of the `nsel` rows d_sel (positions inside the block): d_out[row] += d_rowmiss[row] * (indicator row) U
int launch_rotate_i8_missing(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const int32_t *d_sel,
                             int nsel, const float *d_rowmiss, const int8_t *d_q, const float *d_umax, float *d_out, int64_t ld_out) {
    if (nsel <= 0) return 0;
    const int nt = num_tiles(n);
    const int64_t npad = (int64_t)nt * JXG_TILE;
    if (!d_q || !d_umax || !d_rowmiss || !d_sel || n > 66000) return fail("launch_rotate_i8_missing: planes / list missing or n > 66000");
    const int nct = (int)((npad + RI_TN - 1) / RI_TN), nrt = (nsel + RI_TM - 1) / RI_TM;
    static bool attr = false;
    const int lds = 2 * RI_SET + 3072;
    if (!attr) {
        JX_HIP(hipFuncSetAttribute((const void *)rotate_i8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    dim3 grid((unsigned)(((nct + 7) / 8) * 8 * nrt));
    if (rot_i8_dma() && m_total < (1LL << 27)) {
        static bool attr_dma = false;
        const int lds_dma = 2 * RD_STAGE + 3072;
        if (!attr_dma) {
            JX_HIP(hipFuncSetAttribute((const void *)rotate_i8_dma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_dma));
            attr_dma = true;
        }
        hipLaunchKernelGGL(rotate_i8_dma_kernel<1>, grid, dim3(512), lds_dma, st, d_p32, m_total, d_rows, nsel,
                           (const uint4 *)nullptr, d_rowmiss, (const float *)nullptr, d_q, d_umax, npad, n, d_out, ld_out, d_sel);
        JX_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(rotate_i8_kernel<1>, grid, dim3(512), lds, st, d_p32, m_total, d_rows, nsel, (const uint4 *)nullptr,
                       d_rowmiss, (const float *)nullptr, d_q, d_umax, npad, n, d_out, ld_out, d_sel);
    JX_LAUNCH_CHECK();
    return 0;
}
}  // namespace jx
