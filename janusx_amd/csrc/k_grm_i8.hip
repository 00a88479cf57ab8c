// Exact-integer GRM Gram term on the int8 matrix pipes: acc(lower tiles) += C C^T, C = allele counts {0,1,2} decoded on the
// fly from the 2-bit P32 payload.
//
// Reference path being replaced: decode_additive_grm_block_f32 (src/decode/decode.rs:728-886) -> cblas_ssyrk per SNP block
// -> f64 merge (src/stats/grm.rs:1638-1667, 1700-1772).  With method 1 the design row of a SNP without missing calls among
// the selected samples is z_ij = b_j + s_j c_ij (c = count of the payload's second allele, s_j = -1 when the reference
// flips the SNP, b_j the centring term), so
//     sum_j z_ij z_kj = sum_j c_ij c_kj + r_i + r_k + B,   r = C (b .* s),   B = sum_j b_j^2
// (s_j^2 = 1: the Gram term does not see the flips).  The Gram term has operands in {0,1,2} and integer sums: it runs on
// v_mfma_i32_32x32x32_i8 (twice the f16 rate, one byte per operand element, exact for 2^29 SNPs per chunk) and the affine
// terms are one f64 matrix-vector product over the payload (jxg_packed_dot), added in the first chunk's merge.
//
// Design (gfx950):
//  * decode is pure VALU, 11 instructions per payload dword (16 samples): value fields V = hi + (hi & w), hi = (w >> 1) &
//    0x5555..., (code 00 -> 0, 10 -> 1, 11 -> 2, 01 = missing / pad -> 0), then the four dwords (V >> 2q) & 0x03030303:
//    byte 4q + b of the 16-byte group holds sample 4b + q.  No table lookups, no per-SNP value LUT, no flips.  The sample
//    order inside a 16-group is therefore a fixed 4 x 4 transposition `pos_to_sample`, applied once in the merge;
//  * LDS images are [k = SNP][position] with one BYTE per element (half the LDS traffic of the fp16 images), written with
//    one ds_write_b128 per payload dword; MFMA operands (16 consecutive k of one position per lane) come from two
//    ds_read_b64_tr_b8 transposed reads (lane map measured with scripts/probes/i8_probe.hip: inside a 16-lane group
//    supplier lane 2 e + c provides row e, columns 8 c .. 8 c + 7, and lane i receives column i); the pitch (row + 32 B)
//    puts the eight rows of a transposed read on disjoint banks;
//  * large n: 256 x 256 output tile per 512-thread workgroup (8 waves of 128 x 64), one workgroup per CU, two image sets:
//    the decode of step k + 1 shares a barrier interval with the MFMAs of step k (decode bytes and fragment reads per
//    MFMA are half those of a 128 x 128 tile); few tiles / row panels: 128 x 128 per 256-thread workgroup, 4 per CU.
#include <stdlib.h>

#include <algorithm>

#include "jx_common.h"

namespace jx {

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// position inside a tile -> sample inside the tile (involution: 4 x 4 transposition inside every group of 16)
__device__ __forceinline__ int pos_to_sample(int p) { return (p & ~15) | ((p & 3) << 2) | ((p >> 2) & 3); }

// 16 two-bit codes -> 16 count bytes (see the header for the order)
__device__ __forceinline__ u32x4v decode16_counts(uint32_t w) {
    const uint32_t hi = (w >> 1) & 0x55555555u;
    const uint32_t v = hi + (hi & w);                      // per 2-bit field: b1 + (b1 & b0)
    u32x4v o;
    o.x = v & 0x03030303u;
    o.y = (v >> 2) & 0x03030303u;
    o.z = (v >> 4) & 0x03030303u;
    o.w = (v >> 6) & 0x03030303u;
    return o;
}

// 16 two-bit codes -> 16 bytes through a per-SNP byte LUT (byte c of `lut` = value of code c): v_perm_b32 with the code bytes
// as selectors.  Same byte order as decode16_counts.
__device__ __forceinline__ u32x4v decode16_lut(uint32_t w, uint32_t lut) {
    u32x4v o;
    o.x = __builtin_amdgcn_perm(lut, lut, w & 0x03030303u);
    o.y = __builtin_amdgcn_perm(lut, lut, (w >> 2) & 0x03030303u);
    o.z = __builtin_amdgcn_perm(lut, lut, (w >> 4) & 0x03030303u);
    o.w = __builtin_amdgcn_perm(lut, lut, (w >> 6) & 0x03030303u);
    return o;
}

// MFMA operand of this lane (16 consecutive k of one position) from a [k][position] byte image: two transposed reads.
// `lane_base` = image + (k0 + 16 (lane >> 5) + ((lane & 15) >> 1)) * PITCH + pos0 + 16 ((lane >> 4) & 1) + 8 (lane & 1).
template <int PITCH>
__device__ __forceinline__ i32x4 tr8_frag(const uint8_t *lane_base) {
    typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
    const i32x2 a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2 *)(lane_base));
    const i32x2 b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2 *)(lane_base + 8 * PITCH));
    i32x4 r;
    r.x = a.x;
    r.y = a.y;
    r.z = b.x;
    r.w = b.y;
    return r;
}

// LUT = true: the operand bytes come from a per-SNP byte LUT `luts[k]` (byte c = value of the 2-bit code c; k = position in the
// row list) instead of the counts {0, 0, 1, 2}, and the i32 Gram is scaled by `gscale` in the merge: the two-Gram form of rows
// with missing calls (k_grm.hip, "dense missing-call path").
template <int TM, int TN, int WM, int WN, int BK, bool DBUF, int MINW, bool LUT = false>
__global__ __launch_bounds__(64 * (TM / WM) * (TN / WN), MINW) void grm_i8_kernel(
    const uint8_t *__restrict__ p32, int64_t m_total, const int32_t *__restrict__ rows, int64_t k_begin, int64_t k_end,
    int64_t kchunk, int nt128, double *__restrict__ acc, int64_t ld, int use_atomic, const double *__restrict__ corr,
    int tile_base, const uint32_t *__restrict__ luts, double gscale) {
    static_assert(TM == TN, "square workgroup tiles");
    constexpr int NWN = TN / WN;
    constexpr int NTHREADS = 64 * (TM / WM) * NWN;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int PITCH = TM + 32;                 // bytes per SNP row of an image
    constexpr int IMG = BK * PITCH;
    constexpr int DW = TM / 16;                    // payload dwords per SNP row of a panel
    constexpr int NL = BK * DW / NTHREADS;         // payload dwords per thread, step and panel
    static_assert(NL * NTHREADS == BK * DW && BK % 32 == 0, "panel dwords must divide evenly");
    constexpr int SET = 2 * IMG;                   // A image, B image
    __shared__ __attribute__((aligned(16))) uint8_t smem[(DBUF ? 2 : 1) * SET];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;

    // lower-triangle tile (ti >= tj) of TM x TM blocks
    const int t = blockIdx.x + tile_base;
    int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((int64_t)ti * (ti + 1) / 2 > t) --ti;
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - (int)((int64_t)ti * (ti + 1) / 2);

    const int64_t k0 = k_begin + (int64_t)blockIdx.y * kchunk;
    const int64_t k1 = (k0 + kchunk < k_end) ? (k0 + kchunk) : k_end;

    // decode mapping: dword idx = tid + u NTHREADS -> (SNP kk = idx / DW of the step, dword d = idx % DW of the panel row);
    // both panels use the same (kk, d), so one record index per u serves both
    const int d_of = tid % DW;
    const int kk_of = tid / DW;                    // + u * (NTHREADS / DW)
    constexpr int KSTRIDE = NTHREADS / DW;
    const int recA128 = ti * (TM / 128) + (d_of >> 3), recB128 = tj * (TM / 128) + (d_of >> 3);
    // Nothing in the step loop is conditional: loads go to clamped (always valid) addresses and are consumed one step
    // later, where rows beyond the chunk / tiles beyond the panel are zeroed with a mask (codes 00 = count 0).  A guarded
    // load is a branch around the load plus a wait right behind it, and branches keep the decode out of the MFMA gaps.
    const uint32_t maskA = recA128 < nt128 ? 0xffffffffu : 0u, maskB = recB128 < nt128 ? 0xffffffffu : 0u;
    const uint8_t *const baseA = p32 + (int64_t)(recA128 < nt128 ? recA128 : nt128 - 1) * m_total * 32 + 4 * (d_of & 7);
    const uint8_t *const baseB = p32 + (int64_t)(recB128 < nt128 ? recB128 : nt128 - 1) * m_total * 32 + 4 * (d_of & 7);

    int32_t rec[NL];                               // payload record (SNP row), loaded two steps ahead of its payload's use
    uint32_t lutn[NL], lutw[NL];                   // LUT: the SNP's byte LUT, travelling with rec (lutn) and with the payload (lutw)
    uint32_t wA[NL], wB[NL];                       // payload dwords, loaded one step ahead of their decode
    auto load_rec = [&](int u, int64_t kbase) {
        const int64_t k = kbase + kk_of + u * KSTRIDE;
        rec[u] = rows[k < k1 ? k : k1 - 1];
        if constexpr (LUT) lutn[u] = luts[k < k1 ? k : k1 - 1];
    };
    auto load_payload = [&](int u) {               // consumes rec[u]
        wA[u] = *reinterpret_cast<const uint32_t *>(baseA + (int64_t)rec[u] * 32);
        wB[u] = *reinterpret_cast<const uint32_t *>(baseB + (int64_t)rec[u] * 32);
        if constexpr (LUT) lutw[u] = lutn[u];
    };
    auto decode_to = [&](uint8_t *base, int u, int64_t kbase) {     // payload of the step that starts at kbase
        const bool valid = kbase + kk_of + u * KSTRIDE < k1;
        const int o = (kk_of + u * KSTRIDE) * PITCH + d_of * 16;
        if constexpr (LUT) {
            // code 00 -> byte 0 of the LUT = 0: masked rows / tiles vanish as in the count decode
            *reinterpret_cast<u32x4v *>(base + o) = decode16_lut(wA[u] & (valid ? maskA : 0u), lutw[u]);
            *reinterpret_cast<u32x4v *>(base + IMG + o) = decode16_lut(wB[u] & (valid ? maskB : 0u), lutw[u]);
        } else {
            *reinterpret_cast<u32x4v *>(base + o) = decode16_counts(wA[u] & (valid ? maskA : 0u));
            *reinterpret_cast<u32x4v *>(base + IMG + o) = decode16_counts(wB[u] & (valid ? maskB : 0u));
        }
    };

    i32x16 c[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[mi][ni][r] = 0;

    const int h = lane >> 5;
    const int lane_off = (16 * h + ((lane & 15) >> 1)) * PITCH + 16 * ((lane >> 4) & 1) + 8 * (lane & 1);
    auto mfma_ks = [&](const uint8_t *base, int ks) {
        const uint8_t *sA = base + lane_off + wm * WM + ks * 32 * PITCH, *sB = base + IMG + lane_off + wn * WN + ks * 32 * PITCH;
        i32x4 a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = tr8_frag<PITCH>(sA + mi * 32);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = tr8_frag<PITCH>(sB + ni * 32);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                c[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[mi], b[ni], c[mi][ni], 0, 0, 0);
    };

    constexpr int KS = BK / 32;
#pragma unroll
    for (int u = 0; u < NL; ++u) load_rec(u, k0);
#pragma unroll
    for (int u = 0; u < NL; ++u) load_payload(u);
#pragma unroll
    for (int u = 0; u < NL; ++u) load_rec(u, k0 + BK);
    if constexpr (!DBUF) {
        // 4 workgroups per CU: other workgroups' MFMAs cover this one's decode
        for (int64_t kbase = k0; kbase < k1; kbase += BK) {
#pragma unroll
            for (int u = 0; u < NL; ++u) decode_to(smem, u, kbase);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                load_payload(u);                   // step kbase + BK
                load_rec(u, kbase + 2 * BK);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mfma_ks(smem, ks);
            __syncthreads();
        }
    } else {
        static_assert(!DBUF || NL % KS == 0 || KS % NL == 0, "decode pieces per k-step");
#pragma unroll
        for (int u = 0; u < NL; ++u) decode_to(smem, u, k0);
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            load_payload(u);
            load_rec(u, k0 + 2 * BK);
        }
        __syncthreads();
        int cur = 0;
        for (int64_t kbase = k0; kbase < k1; kbase += BK) {
            // one barrier interval: the MFMAs of this step from image set `cur`, the decode of the next step into the other
            // set, piece by piece between the k-steps (VALU and LDS writes ride in the MFMA issue gaps), and behind every
            // decoded piece the loads that refill its registers for the step after
            const uint8_t *rd = smem + cur * SET;
            uint8_t *wr = smem + (cur ^ 1) * SET;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                mfma_ks(rd, ks);
#pragma unroll
                for (int u = ks * NL / KS; u < (ks + 1) * NL / KS; ++u) {
                    decode_to(wr, u, kbase + BK);
                    load_payload(u);               // step kbase + 2 BK (rec loaded during the previous step)
                    load_rec(u, kbase + 3 * BK);
                }
                // keep every piece's loads in its own k-step: sunk to the end of the iteration (what the scheduler does
                // otherwise) they have a quarter of a step instead of a whole one to land before their decode
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    // f64 merge.  C/D layout of the 32 x 32 shapes: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): tile
    // POSITIONS, mapped to samples by pos_to_sample.
    const bool add_corr = corr != nullptr && k0 == k_begin;
    const double corr_b = add_corr ? corr[ld] : 0.0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int64_t gj = (int64_t)tj * TN + pos_to_sample(wn * WN + ni * 32 + (lane & 31));
            const double corr_j = (add_corr && gj < ld) ? corr[gj] + corr_b : 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = (int64_t)ti * TM + pos_to_sample(wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h);
                if (gi < ld && gj < ld) {
                    double *dst = acc + gi * ld + gj;
                    double v = (double)c[mi][ni][r];
                    if constexpr (LUT) v *= gscale;
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

// k_grm_fp4.hip: the same Gram on the fp4 pipes from a nibble image of the counts
bool grm_fp4_enabled();
int launch_grm_fp4(hipStream_t st, const uint8_t *p32, int64_t m_total, const int32_t *rows, int64_t r0, int64_t r1, int nt128,
                   double *d_acc, int64_t ld, const double *corr, int64_t base256, int64_t ntl256);

// Launch over the SNP range [r0, r1) of the (reordered) list `rows`.  Lower-triangle tiles of the n_sel x n_sel accumulator
// (ld = 128 nt128); panel mode: tile rows [tile_row_begin, tile_row_end) of 128-row tiles only (d_acc already shifted so
// that global sample rows index it).  `corr`: r[0 .. ld) and B at [ld], added once.
int launch_grm_i8(hipStream_t st, const uint8_t *p32, int64_t m_total, const int32_t *rows, int64_t r0, int64_t r1, int nt128,
                  double *d_acc, int64_t ld, const double *corr, bool panel, int tile_row_begin, int tile_row_end) {
    const int64_t cnt = r1 - r0;
    if (cnt <= 0) return 0;
    if (!rows) return fail("launch_grm_i8: the SNP row list is required");
    const int tile_env = getenv("JXGPU_GRM_I8_TILE") ? atoi(getenv("JXGPU_GRM_I8_TILE")) : 0;   // 256 / 128 force a tile shape
    const int nt256 = (nt128 + 1) / 2;
    // 256 x 256 tiles (one workgroup per CU) once they fill the chip a few times over; row panels (tile rows counted in
    // 128-row units) take them when the panel starts on a 256-row boundary: rows [begin / 2, ceil(end / 2)) of the 256-tile
    // triangle (a trailing half tile is masked by the kernel's row guards)
    const int pb256 = panel ? tile_row_begin / 2 : 0, pe256 = panel ? (tile_row_end + 1) / 2 : nt256;
    const int64_t base256 = (int64_t)pb256 * (pb256 + 1) / 2;
    const int64_t ntl256 = (int64_t)pe256 * (pe256 + 1) / 2 - base256;
    const bool big = (!panel || (tile_row_begin % 2 == 0 && (tile_row_end % 2 == 0 || tile_row_end == nt128))) &&
                     (tile_env ? tile_env >= 256 : ntl256 >= 3 * 256);
    if (big) {
        if (base256 + ntl256 > 0x7fffffffLL) return fail("jxg_grm_accumulate: too many tiles");
        if (grm_fp4_enabled()) return launch_grm_fp4(st, p32, m_total, rows, r0, r1, nt128, d_acc, ld, corr, base256, ntl256);
        // i32 sums stay exact for 2^29 SNPs per chunk: one launch for the whole range
        for (int64_t kb = r0; kb < r1; kb += (int64_t)1 << 29) {
            const int64_t ke = (kb + ((int64_t)1 << 29) < r1) ? kb + ((int64_t)1 << 29) : r1;
            hipLaunchKernelGGL((grm_i8_kernel<256, 256, 128, 64, 128, true, 2>), dim3((unsigned)ntl256, 1), dim3(512), 0, st,
                               p32, m_total, rows, kb, ke, (int64_t)1 << 29, nt128, d_acc, ld, 0, kb == r0 ? corr : nullptr,
                               (int)base256, (const uint32_t *)nullptr, 1.0);
            JX_LAUNCH_CHECK();
        }
        return 0;
    }
    const int64_t tile_base = panel ? (int64_t)tile_row_begin * (tile_row_begin + 1) / 2 : 0;
    const int64_t ntiles = panel ? (int64_t)tile_row_end * (tile_row_end + 1) / 2 - tile_base : (int64_t)nt128 * (nt128 + 1) / 2;
    if (tile_base + ntiles > 0x7fffffffLL) return fail("jxg_grm_accumulate: too many tiles");
    const int64_t slots = 4 * 256;                 // resident workgroups on 256 CUs
    if (ntiles < 2 * slots && cnt > 4096) {
        // few tiles: SNP chunks over blockIdx.y, f64 atomics (the i32 partial sums are exact, the f64 adds commute up to
        // rounding of the affine terms only)
        int64_t want = (2 * slots + ntiles - 1) / ntiles;
        int64_t kc = (cnt + want - 1) / want;
        kc = ((kc + 63) / 64) * 64;
        if (kc < 2048) kc = 2048;
        const int64_t ny = (cnt + kc - 1) / kc;
        if (ny > 65535) return fail("jxg_grm_accumulate: too many chunks");
        hipLaunchKernelGGL((grm_i8_kernel<128, 128, 64, 64, 64, false, 4>), dim3((unsigned)ntiles, (unsigned)ny), dim3(256), 0,
                           st, p32, m_total, rows, r0, r1, kc, nt128, d_acc, ld, 1, corr, (int)tile_base, (const uint32_t *)nullptr, 1.0);
        JX_LAUNCH_CHECK();
        return 0;
    }
    for (int64_t kb = r0; kb < r1; kb += (int64_t)1 << 29) {
        const int64_t ke = (kb + ((int64_t)1 << 29) < r1) ? kb + ((int64_t)1 << 29) : r1;
        hipLaunchKernelGGL((grm_i8_kernel<128, 128, 64, 64, 64, false, 4>), dim3((unsigned)ntiles, 1), dim3(256), 0, st, p32,
                           m_total, rows, kb, ke, (int64_t)1 << 29, nt128, d_acc, ld, 0, kb == r0 ? corr : nullptr,
                           (int)tile_base, (const uint32_t *)nullptr, 1.0);
        JX_LAUNCH_CHECK();
    }
    return 0;
}

// The same Gram with the operand bytes taken from the per-SNP LUTs `luts` (positions r0 .. r1 of the list) and the i32 sums
// scaled by `gscale`: whole lower triangle only (no row panels).  |byte| <= 127, so the i32 sums are exact for 2^31 / 127^2 SNPs:
// chunks of 131072.
int launch_grm_i8_lut(hipStream_t st, const uint8_t *p32, int64_t m_total, const int32_t *rows, const uint32_t *luts, int64_t r0,
                      int64_t r1, int nt128, double *d_acc, int64_t ld, const double *corr, double gscale) {
    if (r1 <= r0) return 0;
    const int nt256 = (nt128 + 1) / 2;
    const int64_t ntl256 = (int64_t)nt256 * (nt256 + 1) / 2;
    const int64_t ntiles = (int64_t)nt128 * (nt128 + 1) / 2;
    if (ntiles > 0x7fffffffLL) return fail("jxg_grm_accumulate: too many tiles");
    const int64_t kc = 131072;
    const bool big = ntl256 >= 3 * 256;
    for (int64_t kb = r0; kb < r1; kb += kc) {
        const int64_t ke = std::min(r1, kb + kc);
        if (big)
            hipLaunchKernelGGL((grm_i8_kernel<256, 256, 128, 64, 128, true, 2, true>), dim3((unsigned)ntl256, 1), dim3(512), 0, st,
                               p32, m_total, rows, kb, ke, kc, nt128, d_acc, ld, 0, kb == r0 ? corr : nullptr, 0, luts, gscale);
        else
            hipLaunchKernelGGL((grm_i8_kernel<128, 128, 64, 64, 64, false, 4, true>), dim3((unsigned)ntiles, 1), dim3(256), 0, st,
                               p32, m_total, rows, kb, ke, kc, nt128, d_acc, ld, 0, kb == r0 ? corr : nullptr, 0, luts, gscale);
        JX_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace jx
