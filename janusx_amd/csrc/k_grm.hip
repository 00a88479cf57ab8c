// GRM construction on MFMA: acc(lower tiles) += Z Z^T, Z decoded on the fly from the 2-bit P32 payload.
//
// Reference path being replaced: decode_additive_grm_block_f32 (src/decode/decode.rs:728-886) ->
// cblas_ssyrk per SNP block into an f32 scratch -> f64 merge (src/stats/grm.rs:1638-1667, 1700-1772) ->
// scale + mirror (src/stats/grm.rs:2771-2785).
//
// Design (gfx950):
//  * one 256-thread workgroup (4 waves, 2x2) owns a 128x128 output tile (ti >= tj) for one SNP chunk;
//  * per step of 32 SNPs each thread loads ONE dword of each panel (16 samples of one SNP, a coalesced 1 KiB
//    chunk of the P32 layout) plus that SNP's 16-byte fp16 LUT, software-prefetched one step ahead;
//  * decode = byte -> two v_perm selectors (256-entry LDS table) -> v_perm_b32 on the SNP's 4-entry fp16 LUT,
//    for the hi and the lo plane of the value split z = hi + lo (fp16 + fp16 = 22 significant bits);
//  * LDS images are [k = SNP][sample] (the natural decode order); MFMA operands (8 consecutive k per lane) are
//    fetched with ds_read_b64_tr_b16, the hardware transpose read; pitch 320 B makes them conflict-free;
//  * three v_mfma_f32_32x32x16_f16 products per k-step (hi*hi + hi*lo + lo*hi) into one f32 accumulator,
//    flushed into the f64 HBM accumulator after <= kchunk SNPs (f32 block / f64 merge like the reference).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "jx_common.h"

namespace jx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int G_BK = 32;        // SNPs per step

// value LUT (mk,4) f32 -> (mk) x {hi[4], lo[4]} fp16; flags[0] |= 1 if a value leaves the safe fp16 range.
__global__ __launch_bounds__(256) void lut_split_kernel(const float *__restrict__ lut, int64_t mk,
                                                        uint4 *__restrict__ out, float prescale,
                                                        int *__restrict__ flags) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= mk) return;
    uint16_t hi[4], lo[4];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float v = lut[k * 4 + c] * prescale;
        if (!(fabsf(v) <= 30000.0f)) bad = true;
        const __half h = __float2half_rn(v);
        const float r = v - __half2float(h);
        const __half l = __float2half_rn(r);
        hi[c] = __half_as_ushort(h);
        lo[c] = __half_as_ushort(l);
    }
    uint4 o;
    o.x = (uint32_t)hi[0] | ((uint32_t)hi[1] << 16);
    o.y = (uint32_t)hi[2] | ((uint32_t)hi[3] << 16);
    o.z = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16);
    o.w = (uint32_t)lo[2] | ((uint32_t)lo[3] << 16);
    out[k] = o;
    if (bad) atomicOr(flags, 1);
}

// selector pair for one payload byte: samples (0,1) -> .x, (2,3) -> .y; each 16-bit result picks bytes
// (2c, 2c+1) of the 8-byte LUT {S0 = entries 2,3 ; S1 = entries 0,1}.
__device__ __forceinline__ uint2 make_selectors(uint32_t byte) {
    const uint32_t c0 = byte & 3u, c1 = (byte >> 2) & 3u, c2 = (byte >> 4) & 3u, c3 = (byte >> 6) & 3u;
    uint2 s;
    s.x = (2u * c0) | ((2u * c0 + 1u) << 8) | ((2u * c1) << 16) | ((2u * c1 + 1u) << 24);
    s.y = (2u * c2) | ((2u * c2 + 1u) << 8) | ((2u * c3) << 16) | ((2u * c3 + 1u) << 24);
    return s;
}

// 16 samples (one payload dword) -> 16 hi + 16 lo fp16 values, written as 2+2 ds_write_b128.
// `half_off` = byte distance between the two 16-byte halves of a plane (see the slot layout in the kernel).
// Selectors come from a 16-entry table indexed by a payload NIBBLE (two samples): 16 dwords sit on 16 distinct
// banks, so the lookup is conflict-free for any genotype distribution (the 256-entry byte table measured ~8-way
// conflicts: common genotype bytes share their low five bits).
__device__ __forceinline__ void decode16_to_lds(uint32_t w, const uint4 L, const uint32_t *__restrict__ seltab,
                                                uint8_t *dst_hi, uint8_t *dst_lo, int half_off) {
    u32x4 h0, h1, l0, l1;
    uint32_t s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = seltab[(w >> (4 * i)) & 15u];
    h0.x = __builtin_amdgcn_perm(L.y, L.x, s[0]);
    h0.y = __builtin_amdgcn_perm(L.y, L.x, s[1]);
    h0.z = __builtin_amdgcn_perm(L.y, L.x, s[2]);
    h0.w = __builtin_amdgcn_perm(L.y, L.x, s[3]);
    h1.x = __builtin_amdgcn_perm(L.y, L.x, s[4]);
    h1.y = __builtin_amdgcn_perm(L.y, L.x, s[5]);
    h1.z = __builtin_amdgcn_perm(L.y, L.x, s[6]);
    h1.w = __builtin_amdgcn_perm(L.y, L.x, s[7]);
    l0.x = __builtin_amdgcn_perm(L.w, L.z, s[0]);
    l0.y = __builtin_amdgcn_perm(L.w, L.z, s[1]);
    l0.z = __builtin_amdgcn_perm(L.w, L.z, s[2]);
    l0.w = __builtin_amdgcn_perm(L.w, L.z, s[3]);
    l1.x = __builtin_amdgcn_perm(L.w, L.z, s[4]);
    l1.y = __builtin_amdgcn_perm(L.w, L.z, s[5]);
    l1.z = __builtin_amdgcn_perm(L.w, L.z, s[6]);
    l1.w = __builtin_amdgcn_perm(L.w, L.z, s[7]);
    *reinterpret_cast<u32x4 *>(dst_hi) = h0;
    *reinterpret_cast<u32x4 *>(dst_hi + half_off) = h1;
    *reinterpret_cast<u32x4 *>(dst_lo) = l0;
    *reinterpret_cast<u32x4 *>(dst_lo + half_off) = l1;
}

// hi plane only (exact-integer steps: the lo plane is identically zero)
__device__ __forceinline__ void decode16_hi_to_lds(uint32_t w, const uint4 L, const uint32_t *__restrict__ seltab,
                                                   uint8_t *dst_hi, int half_off) {
    u32x4 h0, h1;
    uint32_t s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = seltab[(w >> (4 * i)) & 15u];
    h0.x = __builtin_amdgcn_perm(L.y, L.x, s[0]);
    h0.y = __builtin_amdgcn_perm(L.y, L.x, s[1]);
    h0.z = __builtin_amdgcn_perm(L.y, L.x, s[2]);
    h0.w = __builtin_amdgcn_perm(L.y, L.x, s[3]);
    h1.x = __builtin_amdgcn_perm(L.y, L.x, s[4]);
    h1.y = __builtin_amdgcn_perm(L.y, L.x, s[5]);
    h1.z = __builtin_amdgcn_perm(L.y, L.x, s[6]);
    h1.w = __builtin_amdgcn_perm(L.y, L.x, s[7]);
    *reinterpret_cast<u32x4 *>(dst_hi) = h0;
    *reinterpret_cast<u32x4 *>(dst_hi + half_off) = h1;
}

// MFMA operand (8 consecutive k for this lane's sample) from a [k][sample] image: two transposed reads.
template <int PITCH>
__device__ __forceinline__ half8 tr_frag(const uint8_t *img_lane_base) {
    typedef __attribute__((address_space(3))) fp16x4 lds_fp16x4;
    const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4 *)(img_lane_base));
    const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4 *)(img_lane_base + 4 * PITCH));
    u32x4 r;
    const u32x2 ua = __builtin_bit_cast(u32x2, a);
    const u32x2 ub = __builtin_bit_cast(u32x2, b);
    r.x = ua.x;
    r.y = ua.y;
    r.z = ub.x;
    r.w = ub.y;
    return __builtin_bit_cast(half8, r);
}

// TM x TN output tile per workgroup, WM x WN per wave (multiples of 32), 64 * (TM/WM) * (TN/WN) threads.
//   <128,128,64,64>  : 4 waves, small n (many tiles needed to fill the chip)
//   <256,128,128,64> : 4 waves of 128x64: each LDS fragment read feeds 1.33x and each decoded byte 1.33x more MFMAs
//                      (the 128x128 shape is LDS-bound: ~930 LDS cycles vs 768 MFMA cycles per 32-SNP step);
//                      2 workgroups per CU keep one decoding while the other issues MFMAs
//   DBUF: two LDS image sets; the decode of step k+1 is issued in the same barrier interval as the MFMAs of step k
//   (one barrier per step, VALU/LDS-write work rides in the MFMA issue gaps) at 2 workgroups per CU.
//   EXACT: every LUT value of the SNP range is a small integer (lo plane zero): one image per panel, one MFMA
//   product per k-step; `corr` (r[0..ld), B at [ld]) adds the affine terms r_i + r_j + B of the integer
//   factorisation z = beta + c to the first chunk's merge (see jxg_grm_accumulate).
template <int TM, int TN, int WM, int WN, bool DBUF, bool EXACT = false, int BK = G_BK>
__global__ __launch_bounds__(64 * (TM / WM) * (TN / WN), (DBUF ? (EXACT ? 4 : 2) : (TM == 128 ? (EXACT ? 4 : 3) : 2))) void grm_f16x2_kernel(
    const uint8_t *__restrict__ p32, int64_t m_total, const int32_t *__restrict__ rows,
    const uint4 *__restrict__ lut16, int64_t k_begin, int64_t k_end, int kchunk, int nt128,
    double *__restrict__ acc, int64_t ld, int use_atomic, const double *__restrict__ corr, int tile_base) {
    constexpr int NWN = TN / WN;
    constexpr int NTHREADS = 64 * (TM / WM) * NWN;
    constexpr int MI = WM / 32, NI = WN / 32;
    // LDS image row (one SNP, TM samples as fp16): the 16 decoded samples of payload dword d are stored as two
    // 16-byte halves in slots d and DW + d (DW = dwords per row), so the DW lanes that decode one SNP row write
    // DW * 16 contiguous bytes per ds_write_b128 (conflict-free; the natural order is 2-way conflicted), and the
    // pitch (row bytes + 32) puts the four rows of a transposed read on disjoint banks (pitch/4 = 8 mod 64).
    constexpr int PA = TM * 2 + 32;   // bytes per SNP row of the A images
    constexpr int PB = TN * 2 + 32;
    constexpr int IMG_A = BK * PA, IMG_B = BK * PB;
    constexpr int DWA = TM / 16, DWB = TN / 16;  // payload dwords per SNP row of each panel
    constexpr int NA = BK * DWA / NTHREADS, NB = BK * DWB / NTHREADS;  // payload dwords per thread per step
    static_assert(NA * NTHREADS == BK * DWA && NB * NTHREADS == BK * DWB, "panel dwords must divide evenly");
    constexpr int NPL = EXACT ? 1 : 2;           // planes per panel
    constexpr int SET = NPL * (IMG_A + IMG_B);   // one set of the images: A hi [A lo] B hi [B lo]
    __shared__ __attribute__((aligned(16))) uint8_t smem[(DBUF ? 2 : 1) * SET + 64];
    uint32_t *seltab = reinterpret_cast<uint32_t *>(smem + (DBUF ? 2 : 1) * SET);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;

    if (tid < 16) seltab[tid] = make_selectors((uint32_t)tid).x;  // selector of codes (t & 3, t >> 2)

    // output tiles covering the lower triangle: row block ti (TM rows) x column blocks tj (TN columns) with
    // tj * TN < (ti + 1) * TM; RATIO = TM / TN column blocks per row-block step
    constexpr int RATIO = TM / TN;
    const int t = blockIdx.x + tile_base;     // tile_base > 0: a panel of tile rows (sparse GRM of large n, row panels)
    int ti, tj;
    if (RATIO == 1) {
        ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((int64_t)ti * (ti + 1) / 2 > t) --ti;
        while ((int64_t)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
        tj = t - (int)((int64_t)ti * (ti + 1) / 2);
    } else {  // RATIO == 2: row block ti owns 2 * ti + 2 column blocks, ti * (ti + 1) precede it
        ti = (int)((sqrtf(4.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((int64_t)ti * (ti + 1) > t) --ti;
        while ((int64_t)(ti + 1) * (ti + 2) <= t) ++ti;
        tj = t - (int)((int64_t)ti * (ti + 1));
    }

    const int64_t k0 = k_begin + (int64_t)blockIdx.y * kchunk;
    const int64_t k1 = (k0 + kchunk < k_end) ? (k0 + kchunk) : k_end;

    // decode mapping: payload dword idx = tid + u * NTHREADS -> (SNP kk of the step, dword d of the panel row)
    uint32_t wA[NA], wB[NB];
    uint4 LA[NA], LB[NB];
    auto prefetch = [&](int64_t kbase) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + u * NTHREADS;
            const int kk = idx / DWA, d = idx % DWA;
            const int rec128 = ti * (TM / 128) + (d >> 3);
            const int64_t k = kbase + kk;
            wA[u] = 0x55555555u;
            LA[u] = make_uint4(0, 0, 0, 0);
            if (k < k1) {
                const int64_t rec = rows ? (int64_t)rows[k] : k;
                if (rec128 < nt128)
                    wA[u] = *reinterpret_cast<const uint32_t *>(p32 + ((int64_t)rec128 * m_total + rec) * 32 +
                                                                4 * (d & 7));
                LA[u] = lut16[k];
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + u * NTHREADS;
            const int kk = idx / DWB, d = idx % DWB;
            const int rec128 = tj * (TN / 128) + (d >> 3);
            const int64_t k = kbase + kk;
            wB[u] = 0x55555555u;
            LB[u] = make_uint4(0, 0, 0, 0);
            if (k < k1) {
                const int64_t rec = rows ? (int64_t)rows[k] : k;
                if (rec128 < nt128)
                    wB[u] = *reinterpret_cast<const uint32_t *>(p32 + ((int64_t)rec128 * m_total + rec) * 32 +
                                                                4 * (d & 7));
                LB[u] = lut16[k];
            }
        }
    };

    floatx16 c[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[mi][ni][r] = 0.0f;

    // per-lane transposed-read geometry
    const int g = lane >> 4;          // 16-lane group
    const int h = lane >> 5;          // k half of the MFMA operand
    const int q = (lane & 15) >> 2;   // row of the 4x16 block this lane addresses
    const int pp = lane & 3;          // 4-column group this lane addresses
    // byte offset of samples 4*pp..4*pp+3 of 16-sample block `blk` inside a row: ((pp >> 1) * DW + blk) * 16 + (pp & 1) * 8
    const int offA = (8 * h + q) * PA + ((pp >> 1) * DWA + (g & 1)) * 16 + (pp & 1) * 8;
    const int offB = (8 * h + q) * PB + ((pp >> 1) * DWB + (g & 1)) * 16 + (pp & 1) * 8;

    auto decode_to = [&](uint8_t *base) {
        uint8_t *sAh = base, *sAl = base + IMG_A, *sBh = base + NPL * IMG_A, *sBl = base + NPL * IMG_A + IMG_B;
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + u * NTHREADS;
            const int kk = idx / DWA, d = idx % DWA;
            if constexpr (EXACT)
                decode16_hi_to_lds(wA[u], LA[u], seltab, sAh + kk * PA + d * 16, DWA * 16);
            else
                decode16_to_lds(wA[u], LA[u], seltab, sAh + kk * PA + d * 16, sAl + kk * PA + d * 16, DWA * 16);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + u * NTHREADS;
            const int kk = idx / DWB, d = idx % DWB;
            if constexpr (EXACT)
                decode16_hi_to_lds(wB[u], LB[u], seltab, sBh + kk * PB + d * 16, DWB * 16);
            else
                decode16_to_lds(wB[u], LB[u], seltab, sBh + kk * PB + d * 16, sBl + kk * PB + d * 16, DWB * 16);
        }
    };
    auto mfma_from = [&](const uint8_t *base) {
        const uint8_t *sAh = base, *sAl = base + IMG_A, *sBh = base + NPL * IMG_A, *sBl = base + NPL * IMG_A + IMG_B;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int off = ks * 16 * PA + offA + ((wm * WM + mi * 32) / 16) * 16;
                ah[mi] = tr_frag<PA>(sAh + off);
                if constexpr (!EXACT) al[mi] = tr_frag<PA>(sAl + off);
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int off = ks * 16 * PB + offB + ((wn * WN + ni * 32) / 16) * 16;
                bh[ni] = tr_frag<PB>(sBh + off);
                if constexpr (!EXACT) bl[ni] = tr_frag<PB>(sBl + off);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    c[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], c[mi][ni], 0, 0, 0);
                    if constexpr (!EXACT) {
                        c[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], c[mi][ni], 0, 0, 0);
                        c[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], c[mi][ni], 0, 0, 0);
                    }
                }
        }
    };

    prefetch(k0);
    __syncthreads();  // selector table ready
    if constexpr (!DBUF) {
        for (int64_t kbase = k0; kbase < k1; kbase += BK) {
            decode_to(smem);
            __syncthreads();
            prefetch(kbase + BK);
            mfma_from(smem);
            __syncthreads();
        }
    } else {
        decode_to(smem);
        prefetch(k0 + BK);
        __syncthreads();
        int cur = 0;
        for (int64_t kbase = k0; kbase < k1; kbase += BK) {
            // same barrier interval: decode step k+1 into the other image set, MFMAs of step k from this one
            if (kbase + BK < k1) decode_to(smem + (cur ^ 1) * SET);
            prefetch(kbase + 2 * BK);
            mfma_from(smem + cur * SET);
            __syncthreads();
            cur ^= 1;
        }
    }

    // f64 merge: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool add_corr = EXACT && corr != nullptr && k0 == k_begin;
    const double corr_b = add_corr ? corr[ld] : 0.0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int64_t gj = (int64_t)tj * TN + wn * WN + ni * 32 + (lane & 31);
            const double corr_j = (add_corr && gj < ld) ? corr[gj] + corr_b : 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = (int64_t)ti * TM + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gi < ld && gj < ld) {
                    double *dst = acc + gi * ld + gj;
                    double v = (double)c[mi][ni][r];
                    if (add_corr) v += corr[gi] + corr_j;
                    if (use_atomic) {
                        unsafeAtomicAdd(dst, v);
                    } else {
                        *dst += v;
                    }
                }
            }
        }
}


// ---- exact-integer factorisation of design rows -------------------------------------------------------------------
// A SNP whose three genotype values are beta + {0,1,2} (method 1: g - 2p, either allele orientation) and that has
// no missing call among the selected samples contributes
//     sum_j (beta_j + c_ij)(beta_j + c_kj) = sum_j c_ij c_kj + r_i + r_k + B,   r = C beta,  B = sum_j beta_j^2,
// with c in {0,1,2}: the Gram term is exact in fp16 x fp16 -> f32 (one MFMA product instead of three, sums
// <= 4 * kchunk < 2^24) and the affine terms are one f64 matrix-vector product over the payload.  The reference
// rounds g - 2p to f32 before its SSYRK (src/decode/decode.rs:813-839), i.e. differs from beta + c by <= 1 ulp(2).

// flag[k] = 1 if SNP k qualifies; beta[k] (f64), ilut[k] = integer LUT {c(00), 0, c(10), c(11)} as f32.
__global__ __launch_bounds__(256) void grm_classify_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                           const int32_t *__restrict__ rows,
                                                           const float *__restrict__ lut, int64_t mk, int n_sel,
                                                           int nt128, int32_t *__restrict__ flag,
                                                           double *__restrict__ beta, float *__restrict__ ilut,
                                                           unsigned long long *__restrict__ miss_total) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= mk) return;
    const float v0 = lut[k * 4 + 0], v2 = lut[k * 4 + 2], v3 = lut[k * 4 + 3];
    const float b = fminf(v0, v3);
    const float c0 = v0 - b, c2 = v2 - b, c3 = v3 - b;
    const float tol = 4e-6f;
    bool ok = fabsf(c2 - 1.0f) <= tol &&
              ((fabsf(c0) <= tol && fabsf(c3 - 2.0f) <= tol) || (fabsf(c0 - 2.0f) <= tol && fabsf(c3) <= tol)) &&
              fabsf(b) <= 4.0f;
    const float r0 = (c0 > 1.0f) ? 2.0f : 0.0f, r3 = 2.0f - r0;
    int fl = 1;
    if (ok) {
        const int64_t rec = rows ? (int64_t)rows[k] : k;
        uint32_t any = 0, nmiss = 0;
        for (int t = 0; t < nt128; ++t) {
            const uint4 *q = reinterpret_cast<const uint4 *>(p32 + ((int64_t)t * m_total + rec) * 32);
            const uint4 a = q[0], c = q[1];
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            const int valid = n_sel - t * 128;   // real samples in this tile (pads are code 01)
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                uint32_t miss = w[d] & ~(w[d] >> 1) & 0x55555555u;
                const int left = valid - d * 16;
                if (left <= 0) miss = 0;
                else if (left < 16) miss &= (1u << (2 * left)) - 1u;
                any |= miss;
                nmiss += __popc(miss);
            }
        }
        // flag 2: affine in the count but with missing calls -- int8 Gram + the sparse correction of k_grm_miss.hip
        // (the missing call's own LUT value must be finite)
        if (any != 0) {
            // the missing call's value as a count, c* = s (z_missing - b): inside [0, 2] for a centred design (2 p of the counted
            // allele), which is what the dense two-Gram path quantises
            const float bta = ((v0 - r0) + (v2 - 1.0f) + (v3 - r3)) / 3.0f;
            const float cst = (r0 == 0.0f ? 1.0f : -1.0f) * (lut[k * 4 + 1] - (bta + r0));
            fl = (isfinite(lut[k * 4 + 1]) && cst >= -0.004f && cst <= 2.004f) ? 2 : 0;
            if (fl == 2 && miss_total) atomicAdd(miss_total, (unsigned long long)nmiss);
        }
        ok = fl != 0;
    }
    flag[k] = ok ? fl : 0;
    beta[k] = ok ? ((double)(v0 - r0) + (double)(v2 - 1.0f) + (double)(v3 - r3)) / 3.0 : 0.0;
    ilut[k * 4 + 0] = r0;
    ilut[k * 4 + 1] = 0.0f;
    ilut[k * 4 + 2] = 1.0f;
    ilut[k * 4 + 3] = r3;
}

// stable partition positions: SNPs with flag 1 first, then flag 2, then the others.  One workgroup; info[0] = number with a
// non-zero flag, info[2] = number with flag 1.
__global__ __launch_bounds__(1024) void grm_partition_kernel(const int32_t *__restrict__ flag, int64_t mk,
                                                             int32_t *__restrict__ pos, int32_t *__restrict__ info) {
    __shared__ int64_t part1[1024], part2[1024];
    const int tid = threadIdx.x;
    const int64_t per = (mk + 1023) / 1024;
    const int64_t b = tid * per, e = (b + per < mk) ? b + per : mk;
    int64_t c1 = 0, c2 = 0;
    for (int64_t k = b; k < e; ++k) {
        c1 += flag[k] == 1;
        c2 += flag[k] == 2;
    }
    part1[tid] = c1;
    part2[tid] = c2;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // inclusive scans
        const int64_t v1 = (tid >= off) ? part1[tid - off] : 0, v2 = (tid >= off) ? part2[tid - off] : 0;
        __syncthreads();
        part1[tid] += v1;
        part2[tid] += v2;
        __syncthreads();
    }
    const int64_t t1 = part1[1023], t2 = part2[1023];
    int64_t n1 = part1[tid] - c1, n2 = part2[tid] - c2;   // rows of either kind before b
    for (int64_t k = b; k < e; ++k) {
        if (flag[k] == 1) pos[k] = (int32_t)n1++;
        else if (flag[k] == 2) pos[k] = (int32_t)(t1 + n2++);
        else pos[k] = (int32_t)(t1 + t2 + (k - n1 - n2));
    }
    if (tid == 0) {
        info[0] = (int32_t)(t1 + t2);
        info[2] = (int32_t)t1;
    }
}

// the rows with missing calls go to the general kernel after all (panel mode, too many missing calls, JXGPU_GRM_MISS=0)
__global__ __launch_bounds__(256) void grm_demote_kernel(int32_t *__restrict__ flag, int64_t mk) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < mk && flag[k] == 2) flag[k] = 0;
}

// reordered SNP list: rows2, split LUT (integer LUT below the 32-aligned exact prefix), f32 integer LUT and beta for
// the affine terms (zero beyond the prefix); info[1] |= 1 if a general value leaves the safe fp16 range.
__global__ __launch_bounds__(256) void grm_gather_kernel(const int32_t *__restrict__ rows, const float *__restrict__ lut,
                                                         const float *__restrict__ ilut, const double *__restrict__ beta,
                                                         const int32_t *__restrict__ pos, int64_t mk,
                                                         int32_t *__restrict__ info, int32_t *__restrict__ rows2,
                                                         uint4 *__restrict__ lut16, float *__restrict__ ilut2,
                                                         double *__restrict__ beta2, int i8mode,
                                                         const int32_t *__restrict__ flag, uint8_t *__restrict__ miss2,
                                                         double *__restrict__ wl, uint32_t *__restrict__ lut_a,
                                                         uint32_t *__restrict__ lut_b, double *__restrict__ dfix,
                                                         uint32_t *__restrict__ lut_c, uint32_t *__restrict__ lut_f) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= mk) return;
    const int32_t nex = info[0] & ~63;
    const int32_t dst = pos[k];
    const bool exact = dst < nex;
    // dense missing-call path (lut_a != nullptr): z = b + s c~ with c~ = the unflipped count, c* = s (z_missing - b) at a missing
    // call, written as 56 c~ ~ P1 + P2 / 31: P1 = rint(56 c*) in [0, 112], P2 = rint(31 (56 c* - P1)) in [-15, 15] (P1 + P2 is
    // an int8); byte LUTs of the two Grams: B = (0, P1, 56, 112), A = (0, P1 + P2, 56, 112);
    // c~ c~' ~ ((1 - 1/31) B B' + (1/31) A A') / 3136 up to the product of two P2 (both samples missing at the SNP: 3e-7 of the
    // mean diagonal at 1 % missing calls off the diagonal; ON the diagonal dfix[k] = c*^2 - ((30 P1^2 + (P1 + P2)^2) / 31) / 3136
    // restores the exact value).  With two digits c* is resolved to 1 / 1736 of a count: 1.7e-4 rms per missing call, which
    // random-walks to sqrt(2 rate m) 1.7e-4 0.55 / (0.3 m) of the mean diagonal -- 2e-7 rms at m = 200 000 and 1 %, 1.3e-6 as
    // the LARGEST entry error at m = 20 000 / 0.2 % (measured, round 6) -- and an error of that size in K moves the per-SNP
    // beta of a whole run by 1.7e-5 (scripts/diag_e2e_two_stage.py).  Hence a THIRD digit (lut_c != nullptr, the default):
    // 56 c~ ~ P1 + P2 / 31 + P3 / 961, P3 = rint(31 (31 (56 c* - P1) - P2)) in [-15, 15], byte LUT C = (0, P1 + P3, 56, 112),
    //   c~ c~' ~ ((1 - 1/31 - 1/961) B B' + (1/31) A A' + (1/961) C C') / 3136
    // -- the resolution 31 x finer (5.4e-6 rms per missing call).  That alone left the LARGEST entry error where it was: the
    // dropped product of two P2 -- both samples missing at the SNP, 2.3e-3 count^2 per coincidence, a dozen coincidences on the
    // worst of 12 million pairs.  It is an exact Gram too: F = P2 e (byte LUT (0, P2, 0, 0)) and
    //   + (1/961 - 1/31) F F'
    // turns the P2 P2' / 31 that A A' brings into the P2 P2' / 961 of the exact product.  Four int8 Grams over the rows with
    // missing calls; what is still dropped are products with a P3 (31 x smaller per coincidence).
    double cstar = 0.0;
    if (lut_a) {
        uint32_t la = 0u, lb = 0u, lc = 0u, lf = 0u;
        double fix = 0.0;
        if (exact) {
            la = lb = lc = (56u << 16) | (112u << 24);
            if (flag[k] == 2) {
                const double r0d = (double)ilut[k * 4 + 0];
                const double bb = beta[k] + r0d, ss = (r0d == 0.0) ? 1.0 : -1.0;
                cstar = ss * ((double)lut[k * 4 + 1] - bb);
                double p1 = rint(56.0 * cstar);
                p1 = fmin(fmax(p1, 0.0), 112.0);
                double p2 = rint(31.0 * (56.0 * cstar - p1));
                p2 = fmin(fmax(p2, -15.0), 15.0);
                double p3 = lut_c ? rint(31.0 * (31.0 * (56.0 * cstar - p1) - p2)) : 0.0;
                p3 = fmin(fmax(p3, -15.0), 15.0);
                lb |= ((uint32_t)(int)p1 & 0xffu) << 8;
                la |= ((uint32_t)(int)(p1 + p2) & 0xffu) << 8;
                lc |= ((uint32_t)(int)(p1 + p3) & 0xffu) << 8;
                lf = ((uint32_t)(int)p2 & 0xffu) << 8;
                const double wa = 1.0 / 31.0, wc = lut_c ? 1.0 / 961.0 : 0.0, wb = 1.0 - wa - wc;
                const double wf = lut_f ? 1.0 / 961.0 - 1.0 / 31.0 : 0.0;
                fix = cstar * cstar -
                      (wb * p1 * p1 + wa * (p1 + p2) * (p1 + p2) + wc * (p1 + p3) * (p1 + p3) + wf * p2 * p2) / 3136.0;
            }
        }
        lut_a[dst] = la;
        lut_b[dst] = lb;
        if (lut_c) lut_c[dst] = lc;
        if (lut_f) lut_f[dst] = lf;
        dfix[dst] = fix;
    }
    if (miss2) {
        // table of the sparse correction (k_grm_miss.hip): clean-form value b + s c by code, d = lut(missing) - b
        const bool m2 = exact && flag[k] == 2;
        miss2[dst] = m2 ? 1 : 0;
        if (exact) {
            const double r0d = (double)ilut[k * 4 + 0];
            const double bb = beta[k] + r0d, ss = (r0d == 0.0) ? 1.0 : -1.0;
            const double d = m2 ? (double)lut[k * 4 + 1] - bb : 0.0;
            double *t = wl + (int64_t)dst * 4;
            t[0] = d * bb;
            t[1] = d * (bb + 0.5 * d);
            t[2] = d * (bb + ss);
            t[3] = d * (bb + 2.0 * ss);
        }
    }
    rows2[dst] = rows ? rows[k] : (int32_t)k;
    uint16_t hi[4], lo[4];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float v = exact ? ilut[k * 4 + c] : lut[k * 4 + c];
        if (!(fabsf(v) <= 30000.0f)) bad = true;
        const __half h = __float2half_rn(v);
        const __half l = __float2half_rn(v - __half2float(h));
        hi[c] = __half_as_ushort(h);
        lo[c] = __half_as_ushort(l);
        // int8 Gram (k_grm_i8.hip): z = b + s c with the UNFLIPPED count c of the payload (LUT 0, 0, 1, 2); else the
        // flipped integer LUT of the fp16 exact variant (z = beta + c~)
        // (dense missing-call path: the missing call carries its own count c*)
        ilut2[(int64_t)dst * 4 + c] = i8mode ? (c == 0 ? 0.0f : (c == 1 ? (float)cstar : (float)(c - 1))) : ilut[k * 4 + c];
    }
    uint4 o;
    o.x = (uint32_t)hi[0] | ((uint32_t)hi[1] << 16);
    o.y = (uint32_t)hi[2] | ((uint32_t)hi[3] << 16);
    o.z = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16);
    o.w = (uint32_t)lo[2] | ((uint32_t)lo[3] << 16);
    lut16[dst] = o;
    if (i8mode) {
        // b = beta + c~(00) (0, or 2 when the reference flips the SNP), s = -1 for a flipped SNP: weight of the affine term
        // r = C (b s); sum b^2 = sum (b s)^2 is formed from the same array (grm_sumsq_kernel)
        const double r0 = (double)ilut[k * 4 + 0];
        beta2[dst] = exact ? (beta[k] + r0) * (r0 == 0.0 ? 1.0 : -1.0) : 0.0;
    } else {
        beta2[dst] = exact ? beta[k] : 0.0;
    }
    if (bad) atomicOr(&info[1], 1);
}

// corr[ld] = sum_k beta2[k]^2, k < info[0] & ~31 (one workgroup)
__global__ __launch_bounds__(1024) void grm_sumsq_kernel(const double *__restrict__ beta2, const int32_t *__restrict__ info,
                                                         double *__restrict__ out) {
    __shared__ double sh[1024];
    const int nex = info[0] & ~63;
    double a = 0.0;
    for (int k = threadIdx.x; k < nex; k += 1024) a += beta2[k] * beta2[k];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// K = acc * inv_scale mirrored to both triangles, cast to the output type (n x n, row-major).
template <typename OutT>
__global__ __launch_bounds__(256) void grm_finalize_kernel(const double *__restrict__ acc, int64_t ld, int n,
                                                           double inv_scale, OutT *__restrict__ out) {
    // 32x32 tiles; blocks with by < bx are skipped. Transposed store goes through LDS for coalescing.
    __shared__ double tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;
    if (bx > by) return;  // (row block by) >= (col block bx): lower triangle
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = by * 32 + r, j = bx * 32 + tx;
        double v = 0.0;
        if (i < n && j < n) v = acc[(int64_t)i * ld + j] * inv_scale;
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = by * 32 + r, j = bx * 32 + tx;
        if (i < n && j < n && (bx < by || j <= i)) out[(int64_t)i * n + j] = (OutT)tile[r][tx];
    }
    // mirrored part: out[j][i] = tile value at (i, j), i > j
    for (int r = ty; r < 32; r += 8) {
        const int jrow = bx * 32 + r;  // row of the upper-triangle element
        const int icol = by * 32 + tx;
        if (jrow < n && icol < n && (bx < by || icol > jrow)) out[(int64_t)jrow * n + icol] = (OutT)tile[tx][r];
    }
}

}  // namespace jx

using namespace jx;

namespace jx {
float g_last_ms[24] = {0.f};  // 0: GRM MFMA kernel(s), 1: rotation kernel, 2-3: symv sample, 4-10: eigensolver stages (eigh.cpp), 11: scan form, 12: int8 share of the last GRM
int g_timer_pending[4] = {0, 0, 0, 0};
extern hipEvent_t g_rot_a, g_rot_b;
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    int init() {
        if (!a) {
            JX_HIP(hipEventCreate(&a));
            JX_HIP(hipEventCreate(&b));
        }
        return 0;
    }
};
static EventPair g_grm_ev;
}  // namespace jx

namespace jx {
// acc[i][i] += v[i]
__global__ void grm_diag_add_kernel(double *__restrict__ acc, int64_t ld, int n, const double *__restrict__ v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[(int64_t)i * ld + i] += v[i];
}
// (k, 4) f32 = (0, 1, 0, 0): indicator of the missing code, the LUT of the diagonal term of the dense missing-call path
__global__ void grm_elut_kernel(float *__restrict__ e, int64_t mk) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < mk) *reinterpret_cast<float4 *>(e + k * 4) = make_float4(0.f, 1.f, 0.f, 0.f);
}
int launch_grm_i8_lut(hipStream_t st, const uint8_t *p32, int64_t m_total, const int32_t *rows, const uint32_t *luts, int64_t r0,
                      int64_t r1, int nt128, double *d_acc, int64_t ld, const double *corr, double gscale);
int launch_grm_i8(hipStream_t st, const uint8_t *p32, int64_t m_total, const int32_t *rows, int64_t r0, int64_t r1, int nt128,
                  double *d_acc, int64_t ld, const double *corr, bool panel, int tile_row_begin, int tile_row_end);
int grm_missing_correction(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n_sel, int nt, const int32_t *rows2,
                           const uint8_t *miss2, const double *wl, int64_t nex, double *d_acc, int64_t ld);
}

extern "C" int jxg_debug_occupancy(int *out) {
    int a = -1, b = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, (const void *)grm_f16x2_kernel<128, 128, 64, 64, false>, 256, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void *)grm_f16x2_kernel<128, 128, 64, 64, true>, 256, 0);
    hipFuncAttributes fa, fb;
    (void)hipFuncGetAttributes(&fa, (const void *)grm_f16x2_kernel<128, 128, 64, 64, false>);
    (void)hipFuncGetAttributes(&fb, (const void *)grm_f16x2_kernel<128, 128, 64, 64, true>);
    out[0] = a; out[1] = b; out[2] = fa.numRegs; out[3] = fb.numRegs; out[4] = (int)fa.sharedSizeBytes; out[5] = (int)fb.sharedSizeBytes;
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    out[6] = (int)pr.maxSharedMemoryPerMultiProcessor; out[7] = (int)pr.sharedMemPerBlock; out[8] = pr.regsPerMultiprocessor; out[9] = pr.regsPerBlock;
    return 0;
}

extern "C" int jxg_packed_dot(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                              const float *d_lut, const double *d_beta, double *d_out, void *stream);

extern "C" float jxg_last_kernel_ms(int which) {
    if (which < 0 || which >= 24) return 0.f;
    if (which == 1 && g_timer_pending[1] && g_rot_b) {
        if (hipEventSynchronize(g_rot_b) == hipSuccess) (void)hipEventElapsedTime(&g_last_ms[1], g_rot_a, g_rot_b);
        g_timer_pending[1] = 0;
    }
    return g_last_ms[which];
}

extern "C" int jxg_grm_accumulate_rows(const uint8_t *d_p32, int64_t m_total, int n_sel, const int32_t *d_rows,
                                       const float *d_lut, int64_t mk, double *d_acc, int kchunk, int precision,
                                       int tile_row_begin, int tile_row_end, void *stream);

extern "C" int jxg_grm_accumulate(const uint8_t *d_p32, int64_t m_total, int n_sel, const int32_t *d_rows,
                                  const float *d_lut, int64_t mk, double *d_acc, int kchunk, int precision,
                                  void *stream) {
    return jxg_grm_accumulate_rows(d_p32, m_total, n_sel, d_rows, d_lut, mk, d_acc, kchunk, precision, 0, -1, stream);
}

// Tile rows [tile_row_begin, tile_row_end) of the lower triangle only (128-row tiles; end < 0: all).  d_acc is then the
// PANEL buffer: (tile_row_end - tile_row_begin) * 128 rows of the full leading dimension, row 0 = sample row
// tile_row_begin * 128 -- the form the sparse GRM takes when the n x n f64 accumulator does not fit in HBM.
extern "C" int jxg_grm_accumulate_rows(const uint8_t *d_p32, int64_t m_total, int n_sel, const int32_t *d_rows,
                                       const float *d_lut, int64_t mk, double *d_acc, int kchunk, int precision,
                                       int tile_row_begin, int tile_row_end, void *stream) {
    if (mk <= 0) return 0;
    if (precision != 0 && precision != 2) return fail("jxg_grm_accumulate: precision=1 (f32 MFMA) path not built yet");
    hipStream_t st = (hipStream_t)stream;
    const int nt = num_tiles(n_sel);
    const int64_t ld = (int64_t)nt * JXG_TILE;
    if (kchunk <= 0) kchunk = 8192;
    kchunk = ((kchunk + 63) / 64) * 64;

    // classify / reorder: SNPs that factor as beta + {0,1,2} without missing calls first (exact single-product path)
    static const int exact_env = getenv("JXGPU_GRM_EXACT") ? atoi(getenv("JXGPU_GRM_EXACT")) : 1;
    static const int exact_bk = getenv("JXGPU_GRM_EXACT_BK") ? atoi(getenv("JXGPU_GRM_EXACT_BK")) : 64;
    // exact prefix on the int8 matrix pipes (k_grm_i8.hip); JXGPU_GRM_I8=0 keeps the fp16 single-product variant
    static const int i8_env = getenv("JXGPU_GRM_I8") ? atoi(getenv("JXGPU_GRM_I8")) : 1;
    if (mk > 0x7fffffffLL) return fail("jxg_grm_accumulate: too many SNPs in one call");
    const int miss_env = getenv("JXGPU_GRM_MISS") ? atoi(getenv("JXGPU_GRM_MISS")) : 1;
    // largest share of missing calls (over the rows that hold any) for which the sparse correction is taken: its cost grows
    // with the number of missing calls (nnz n table lookups), the split kernel's does not; measured at n = 20000,
    // m = 200000: 54 ms + 11.5 ms per 0.1 % against 229 ms, i.e. a crossover at ~1.5 %
    const double miss_max = getenv("JXGPU_GRM_MISS_MAX") ? atof(getenv("JXGPU_GRM_MISS_MAX")) : 0.012;
    DevBuf lut16, flagb, betab, ilutb, posb, infob, rows2b, ilut2b, beta2b, corrb, miss2b, wlb, misstotb, lutab, lutbb, lutcb, lutfb, dfixb, elutb, dvecb;
    // digits of the missing call's count in the dense form: 3 + the exact product of two second digits (default; see
    // grm_gather_kernel), JXGPU_GRM_MISS_DIGITS=2: the two-Gram form of rounds 4 - 5
    const bool three_digits = !(getenv("JXGPU_GRM_MISS_DIGITS") && atoi(getenv("JXGPU_GRM_MISS_DIGITS")) == 2);
    // Rows that are affine in the count but hold missing calls, two forms: the SPARSE correction behind the
    // clean int8 Gram (k_grm_miss.hip: nnz n table lookups -- 54 ms + 11.5 ms per 0.1 % of missing calls at n = 20 000,
    // m = 200 000) up to JXGPU_GRM_MISS_DENSE_MIN (0.15 %) of missing calls, the DENSE two-Gram form above it: the missing
    // call's count c* in two int8 digits, (30 B B' + A A') / 31 / 3136 with the byte LUTs B = (0, P1, 56, 112) and
    // A = (0, P1 + P2, 56, 112) -- two int8 Gram products whatever the rate, 2e-7 of the mean diagonal at m = 200 000 (the
    // quantisation noise of c* falls with sqrt(m): from JXGPU_GRM_MISS_DENSE_ROWS = 16 384 SNPs on).
    const double dense_min = getenv("JXGPU_GRM_MISS_DENSE_MIN") ? atof(getenv("JXGPU_GRM_MISS_DENSE_MIN")) : 0.0015;
    const int64_t dense_rows = getenv("JXGPU_GRM_MISS_DENSE_ROWS") ? atoll(getenv("JXGPU_GRM_MISS_DENSE_ROWS")) : 16384;
    bool use_dense = false;
    if (lut16.alloc(sizeof(uint4) * (size_t)mk) || flagb.alloc(sizeof(int32_t) * (size_t)mk) ||
        betab.alloc(sizeof(double) * (size_t)mk) || ilutb.alloc(sizeof(float) * 4 * (size_t)mk) ||
        posb.alloc(sizeof(int32_t) * (size_t)mk) || infob.alloc(4 * sizeof(int32_t)) ||
        rows2b.alloc(sizeof(int32_t) * (size_t)mk) || ilut2b.alloc(sizeof(float) * 4 * (size_t)mk) ||
        beta2b.alloc(sizeof(double) * (size_t)mk) || corrb.alloc(sizeof(double) * (size_t)(ld + 1)))
        return 1;
    JX_HIP(hipMemsetAsync(infob.p, 0, 4 * sizeof(int32_t), st));
    const unsigned gk = (unsigned)((mk + 255) / 256);
    // rows that are affine in the count but hold missing calls: int8 Gram + sparse correction (k_grm_miss.hip) when the whole
    // triangle is built here (not a row panel), the int8 path is on and the missing calls are few enough (miss_max)
    bool use_miss = exact_env && i8_env && miss_env && tile_row_end < 0 && precision != 2;
    if (use_miss && misstotb.alloc(sizeof(unsigned long long))) return 1;
    if (use_miss) JX_HIP(hipMemsetAsync(misstotb.p, 0, sizeof(unsigned long long), st));
    if (exact_env) {
        hipLaunchKernelGGL(grm_classify_kernel, dim3(gk), dim3(256), 0, st, d_p32, m_total, d_rows, d_lut, mk, n_sel, nt,
                           flagb.as<int32_t>(), betab.as<double>(), ilutb.as<float>(),
                           use_miss ? misstotb.as<unsigned long long>() : (unsigned long long *)nullptr);
        JX_LAUNCH_CHECK();
        if (use_miss) {
            unsigned long long hm = 0;
            JX_HIP(hipMemcpyAsync(&hm, misstotb.p, sizeof(hm), hipMemcpyDeviceToHost, st));
            JX_HIP(hipStreamSynchronize(st));
            const double rate = (double)hm / ((double)mk * (double)n_sel);
            if (hm > 0 && rate >= dense_min && rate <= 0.06 && mk >= dense_rows) {
                use_dense = true;          // flagged rows keep flag 2 and take the two-Gram form
                use_miss = false;
            }
            if (!use_dense && (hm == 0 || (double)hm > miss_max * (double)mk * (double)n_sel || hm > 0x7fffffffULL)) use_miss = false;
            if (use_miss) {
                // the correction needs an n_pad^2 f64 buffer + the lists (20 GB at n = 50 000, 80 GB at n = 100 000): decided
                // HERE, while the flagged rows can still be handed to the general kernel -- once the int8 kernel has added
                // their clean form there is no way back
                size_t fr = 0, tot = 0;
                const size_t need = sizeof(double) * (size_t)ld * (size_t)ld + sizeof(int32_t) * (size_t)hm +
                                    sizeof(int64_t) * ((size_t)ld * 32 + 2) + 5 * (size_t)mk + sizeof(double) * 4 * (size_t)mk;
                if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < need + ((size_t)2 << 30)) use_miss = false;
            }
        }
        if (!use_miss && !use_dense) {
            hipLaunchKernelGGL(grm_demote_kernel, dim3(gk), dim3(256), 0, st, flagb.as<int32_t>(), mk);
            JX_LAUNCH_CHECK();
        }
    } else {
        JX_HIP(hipMemsetAsync(flagb.p, 0, sizeof(int32_t) * (size_t)mk, st));
        JX_HIP(hipMemsetAsync(ilutb.p, 0, sizeof(float) * 4 * (size_t)mk, st));
        JX_HIP(hipMemsetAsync(betab.p, 0, sizeof(double) * (size_t)mk, st));
    }
    hipLaunchKernelGGL(grm_partition_kernel, dim3(1), dim3(1024), 0, st, flagb.as<int32_t>(), mk, posb.as<int32_t>(),
                       infob.as<int32_t>());
    JX_LAUNCH_CHECK();
    if (use_miss && (miss2b.alloc((size_t)mk) || wlb.alloc(sizeof(double) * 4 * (size_t)mk))) return 1;
    if (use_dense && three_digits && (lutcb.alloc(sizeof(uint32_t) * (size_t)mk) || lutfb.alloc(sizeof(uint32_t) * (size_t)mk))) return 1;
    if (use_dense && (lutab.alloc(sizeof(uint32_t) * (size_t)mk) || lutbb.alloc(sizeof(uint32_t) * (size_t)mk) ||
                      dfixb.alloc(sizeof(double) * (size_t)mk) || elutb.alloc(sizeof(float) * 4 * (size_t)mk) ||
                      dvecb.alloc(sizeof(double) * (size_t)(ld + 1))))
        return 1;
    hipLaunchKernelGGL(grm_gather_kernel, dim3(gk), dim3(256), 0, st, d_rows, d_lut, ilutb.as<float>(),
                       betab.as<double>(), posb.as<int32_t>(), mk, infob.as<int32_t>(), rows2b.as<int32_t>(),
                       lut16.as<uint4>(), ilut2b.as<float>(), beta2b.as<double>(), i8_env, flagb.as<int32_t>(),
                       use_miss ? miss2b.as<uint8_t>() : (uint8_t *)nullptr, use_miss ? wlb.as<double>() : (double *)nullptr,
                       use_dense ? lutab.as<uint32_t>() : (uint32_t *)nullptr, use_dense ? lutbb.as<uint32_t>() : (uint32_t *)nullptr,
                       use_dense ? dfixb.as<double>() : (double *)nullptr,
                       (use_dense && three_digits) ? lutcb.as<uint32_t>() : (uint32_t *)nullptr,
                       (use_dense && three_digits) ? lutfb.as<uint32_t>() : (uint32_t *)nullptr);
    JX_LAUNCH_CHECK();
    int32_t hinfo[4] = {0, 0, 0, 0};
    JX_HIP(hipMemcpyAsync(hinfo, infob.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    if (hinfo[1]) return fail("jxg_grm_accumulate: design values exceed the fp16 split range (|z| > 3e4)");
    const int64_t n_exact = (int64_t)(hinfo[0] & ~63);
    g_last_ms[12] = (i8_env && mk > 0) ? (float)((double)n_exact / (double)mk) : 0.f;   // share of the SNPs on the int8 path
    const int32_t *rows2 = rows2b.as<int32_t>();

    // tile shape: 256x128 (4 waves of 128x64) once there are enough tiles, else 128x128 (4 waves of 64x64)
    static const int tile_env = getenv("JXGPU_GRM_TILE") ? atoi(getenv("JXGPU_GRM_TILE")) : 0;
    const bool panel = tile_row_end >= 0;
    if (panel && (tile_row_begin < 0 || tile_row_end > nt || tile_row_begin >= tile_row_end))
        return fail("jxg_grm_accumulate_rows: tile row range out of bounds");
    const bool big = (tile_env && !panel) ? (tile_env >= 256) : false;
    const int tdim = big ? (nt + 1) / 2 : nt;  // row blocks
    const int64_t tile_base = panel ? (int64_t)tile_row_begin * (tile_row_begin + 1) / 2 : 0;
    const int64_t ntiles = panel ? (int64_t)tile_row_end * (tile_row_end + 1) / 2 - tile_base
                                 : (big ? (int64_t)tdim * (tdim + 1) : (int64_t)tdim * (tdim + 1) / 2);
    if (tile_base + ntiles > 0x7fffffffLL) return fail("jxg_grm_accumulate: too many tiles");
    // the kernels index the accumulator by global sample row: shift the panel buffer accordingly
    d_acc -= panel ? (int64_t)tile_row_begin * JXG_TILE * ld : 0;
    if (ntiles > 0x7fffffffLL) return fail("jxg_grm_accumulate: too many tiles");
    const int64_t slots = big ? 512 : 768;  // resident workgroups on 256 CUs
    if (g_grm_ev.init()) return 1;
    JX_HIP(hipEventRecord(g_grm_ev.a, st));
    if (n_exact > 0) {   // affine terms of the exact prefix: r = C beta (f64), B = sum beta^2
        JX_HIP(hipMemsetAsync(corrb.p, 0, sizeof(double) * (size_t)(ld + 1), st));
        if (jxg_packed_dot(d_p32, m_total, n_sel, rows2, (int)n_exact, ilut2b.as<float>(), beta2b.as<double>(),
                           corrb.as<double>(), st))
            return 1;
        hipLaunchKernelGGL(grm_sumsq_kernel, dim3(1), dim3(1024), 0, st, beta2b.as<double>(), infob.as<int32_t>(),
                           corrb.as<double>() + ld);
        JX_LAUNCH_CHECK();
    }
    // split (three-product) path with 256 x 256 tiles, 8 waves, two image sets (one barrier per 32-SNP step): measured
    // SLOWER than the 128 x 128 kernel at 4 workgroups per CU (235 vs 223 ms at n = 20000, m = 200000 with 1 % missing
    // calls; round 3) -- kept behind JXGPU_GRM_TILE=257 for comparison only.
    const int nt256 = (nt + 1) / 2;
    const int64_t tiles256 = (int64_t)nt256 * (nt256 + 1) / 2;
    const bool split256 = !panel && tile_env == 257;
    auto launch = [&](bool exact, dim3 grid, int64_t kb, int64_t ke, int kc, int atomic, const double *corr) {
        if (!exact && split256)
            hipLaunchKernelGGL((grm_f16x2_kernel<256, 256, 128, 64, true, false, 32>), dim3((unsigned)tiles256, grid.y),
                               dim3(512), 0, st, d_p32, m_total, rows2, lut16.as<uint4>(), kb, ke, kc, nt, d_acc, ld, atomic,
                               corr, 0);
        else if (exact && big)
            hipLaunchKernelGGL((grm_f16x2_kernel<256, 128, 128, 64, false, true>), grid, dim3(256), 0, st, d_p32, m_total,
                               rows2, lut16.as<uint4>(), kb, ke, kc, nt, d_acc, ld, atomic, corr, (int)tile_base);
        else if (exact)
            if (exact_bk == 33)
                hipLaunchKernelGGL((grm_f16x2_kernel<128, 128, 64, 64, true, true>), grid, dim3(256), 0, st, d_p32,
                                   m_total, rows2, lut16.as<uint4>(), kb, ke, kc, nt, d_acc, ld, atomic, corr, (int)tile_base);
            else if (exact_bk == 64)
                hipLaunchKernelGGL((grm_f16x2_kernel<128, 128, 64, 64, false, true, 64>), grid, dim3(256), 0, st, d_p32,
                                   m_total, rows2, lut16.as<uint4>(), kb, ke, kc, nt, d_acc, ld, atomic, corr, (int)tile_base);
            else
                hipLaunchKernelGGL((grm_f16x2_kernel<128, 128, 64, 64, false, true>), grid, dim3(256), 0, st, d_p32,
                                   m_total, rows2, lut16.as<uint4>(), kb, ke, kc, nt, d_acc, ld, atomic, corr, (int)tile_base);
        else if (big)
            hipLaunchKernelGGL((grm_f16x2_kernel<256, 128, 128, 64, false>), grid, dim3(256), 0, st, d_p32, m_total,
                               rows2, lut16.as<uint4>(), kb, ke, kc, nt, d_acc, ld, atomic, corr, (int)tile_base);
        else if (tile_env == 129)
            hipLaunchKernelGGL((grm_f16x2_kernel<128, 128, 64, 64, true>), grid, dim3(256), 0, st, d_p32, m_total,
                               rows2, lut16.as<uint4>(), kb, ke, kc, nt, d_acc, ld, atomic, corr, (int)tile_base);
        else
            hipLaunchKernelGGL((grm_f16x2_kernel<128, 128, 64, 64, false>), grid, dim3(256), 0, st, d_p32, m_total,
                               rows2, lut16.as<uint4>(), kb, ke, kc, nt, d_acc, ld, atomic, corr, (int)tile_base);
    };
    // SNP range [r0, r1) of the reordered list with one kernel variant
    auto run_range = [&](bool exact, int64_t r0, int64_t r1) -> int {
        const int64_t cnt = r1 - r0;
        if (cnt <= 0) return 0;
        const double *corr = exact ? corrb.as<double>() : nullptr;
        // Few tiles: spread SNP chunks over blockIdx.y with f64 atomics so the chip is filled.
        // Many tiles: one launch per chunk, the owning workgroup does a plain f64 read-modify-write.
        const bool atomic_mode = ntiles < 2 * slots && cnt > 2048 && !(split256 && !exact);
        if (atomic_mode) {
            // shrink chunks if that is what it takes to reach ~2 rounds of resident workgroups
            int64_t want = (2 * slots + ntiles - 1) / ntiles;
            int64_t kc = kchunk;
            while ((cnt + kc - 1) / kc < want && kc > 1024) kc /= 2;
            kc = ((kc + 63) / 64) * 64;
            const int64_t ny = (cnt + kc - 1) / kc;
            if (ny > 65535) return fail("jxg_grm_accumulate: too many chunks");
            launch(exact, dim3((unsigned)ntiles, (unsigned)ny), r0, r1, (int)kc, 1, corr);
            JX_LAUNCH_CHECK();
        } else {
            // exact sums stay below 2^24 for 4 M SNPs: one launch per 2^20 SNPs instead of one per kchunk
            const int64_t kcl = exact ? (int64_t)(1 << 20) : (int64_t)kchunk;
            for (int64_t kb = r0; kb < r1; kb += kcl) {
                const int64_t ke = (kb + kcl < r1) ? kb + kcl : r1;
                launch(exact, dim3((unsigned)ntiles, 1), kb, ke, (int)kcl, 0, kb == r0 ? corr : nullptr);
                JX_LAUNCH_CHECK();
            }
        }
        return 0;
    };
    if (i8_env && use_dense) {
        // clean rows [0, n_clean): counts as they are; rows with missing calls [n_clean, n_exact): the two Grams
        const int64_t n_clean = std::min<int64_t>(hinfo[2], n_exact);
        if (launch_grm_i8(st, d_p32, m_total, rows2, 0, n_clean, nt, d_acc, ld, n_clean > 0 ? corrb.as<double>() : nullptr,
                          panel, tile_row_begin, tile_row_end))
            return 1;
        if (n_exact > n_clean) {
            const double w_a = 1.0 / 31.0, w_c = three_digits ? 1.0 / 961.0 : 0.0, w_b = 1.0 - w_a - w_c;
            if (launch_grm_i8_lut(st, d_p32, m_total, rows2, lutbb.as<uint32_t>(), n_clean, n_exact, nt, d_acc, ld,
                                  n_clean > 0 ? nullptr : corrb.as<double>(), w_b / 3136.0))
                return 1;
            if (launch_grm_i8_lut(st, d_p32, m_total, rows2, lutab.as<uint32_t>(), n_clean, n_exact, nt, d_acc, ld, nullptr,
                                  w_a / 3136.0))
                return 1;
            if (three_digits && launch_grm_i8_lut(st, d_p32, m_total, rows2, lutcb.as<uint32_t>(), n_clean, n_exact, nt, d_acc, ld,
                                                  nullptr, w_c / 3136.0))
                return 1;
            // the product of two second digits exactly: F = P2 e, weight 1/961 - 1/31 (see grm_gather_kernel)
            if (three_digits && launch_grm_i8_lut(st, d_p32, m_total, rows2, lutfb.as<uint32_t>(), n_clean, n_exact, nt, d_acc, ld,
                                                  nullptr, (1.0 / 961.0 - 1.0 / 31.0) / 3136.0))
                return 1;
            // the diagonal exactly: sum over the SNPs a sample misses of dfix
            hipLaunchKernelGGL(grm_elut_kernel, dim3((unsigned)((n_exact - n_clean + 255) / 256)), dim3(256), 0, st,
                               elutb.as<float>(), n_exact - n_clean);
            JX_LAUNCH_CHECK();
            if (jxg_packed_dot(d_p32, m_total, n_sel, rows2 + n_clean, (int)(n_exact - n_clean), elutb.as<float>(),
                               dfixb.as<double>() + n_clean, dvecb.as<double>(), st))
                return 1;
            hipLaunchKernelGGL(grm_diag_add_kernel, dim3((n_sel + 255) / 256), dim3(256), 0, st, d_acc, ld, n_sel, dvecb.as<double>());
            JX_LAUNCH_CHECK();
        }
    } else if (i8_env) {
        if (launch_grm_i8(st, d_p32, m_total, rows2, 0, n_exact, nt, d_acc, ld, n_exact > 0 ? corrb.as<double>() : nullptr,
                          panel, tile_row_begin, tile_row_end))
            return 1;
        if (use_miss && n_exact > 0) {
            const int rc = grm_missing_correction(st, d_p32, m_total, n_sel, nt, rows2, miss2b.as<uint8_t>(), wlb.as<double>(),
                                                  n_exact, d_acc, ld);
            if (rc == 2) return fail("jxg_grm_accumulate: not enough device memory for the missing-call correction "
                                     "(set JXGPU_GRM_MISS=0 to take the general kernel for rows with missing calls)");
            if (rc) return 1;
        }
    } else if (run_range(true, 0, n_exact)) {
        return 1;
    }
    if (run_range(false, n_exact, mk)) return 1;
    JX_HIP(hipEventRecord(g_grm_ev.b, st));
    JX_HIP(hipStreamSynchronize(st));  // lut16 is freed on return
    JX_HIP(hipEventElapsedTime(&g_last_ms[0], g_grm_ev.a, g_grm_ev.b));
    return 0;
}

extern "C" int jxg_grm_finalize(const double *d_acc, int n, double inv_scale, void *d_out, int out_is_f64,
                                void *stream) {
    const int nt = num_tiles(n);
    const int64_t ld = (int64_t)nt * JXG_TILE;
    const int nb = (n + 31) / 32;
    dim3 grid(nb, nb);
    if (out_is_f64) {
        hipLaunchKernelGGL(grm_finalize_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, d_acc, ld, n,
                           inv_scale, (double *)d_out);
    } else {
        hipLaunchKernelGGL(grm_finalize_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, d_acc, ld, n,
                           inv_scale, (float *)d_out);
    }
    JX_LAUNCH_CHECK();
    return 0;
}
