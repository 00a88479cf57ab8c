// The count Gram of the GRM (k_grm_i8.hip: acc(lower tiles) += C C^T, C = allele counts {0, 1, 2} of SNPs without missing calls among
// the selected samples) on the fp4 matrix pipes: v_mfma_scale_f32_32x32x64_f8f6f4 with both operands e2m1 (0 / 1 / 2 = 0x0 / 0x2 / 0x4,
// scales 2^0) issues twice the multiply-adds of v_mfma_i32_32x32x32_i8 per clock, and its f32 accumulation of these small integers
// is exact below 2^24 (scripts/probes/fp4_probe.hip: operand layout -- lane l holds row l % 32 and k = 32 (l / 32) + q in nibble q of
// registers 0 .. 3 --, 40 000 chained products exact, 8.78 against 4.79 POP/s).  Reference path being replaced: as k_grm_i8.hip
// (decode_additive_grm_block_f32 -> cblas_ssyrk -> f64 merge, src/stats/grm.rs:1638-1772).
//
// A 2-bit -> nibble decode in the Gram kernel would cost more VALU time than the products take (16 instructions per 16 genotypes
// against 512 cycles of products per 128-SNP step), so the counts are written ONCE per chunk of SNPs as a nibble image
//     nib[tile of 128 samples][SNP of the chunk][64 B],   nibble s of a record = fp4(count of sample 128 tile + s)
// (4 bits per genotype: grm_nib_kernel, an HBM-bound pass) and the Gram kernel only moves bytes: 8-byte loads, ds_write_b64 into an LDS
// image [SNP][position] of nibbles, operands by two ds_read_b64_tr_b4 (a 16 x 16 nibble transpose per 16-lane group: output lane i,
// nibble q <- supplier lane q, nibble i; scripts/probes/tr4_probe.hip).  256 x 256 output tile per 512-thread workgroup, 8 waves of
// 128 x 64, 128 SNPs per step in two image sets, as the int8 kernel.  Position = sample (no transposition inside 16-groups).
// Chunks of at most 2^22 SNPs per launch keep the f32 sums exact (4 x 2^22 = 2^24); the f64 merge and the affine terms are those of
// the int8 kernel: the accumulator is the same bits.
#include <stdlib.h>

#include <algorithm>

#include "jx_common.h"

namespace jx {

typedef int f4_i32x2 __attribute__((ext_vector_type(2)));
typedef int f4_i32x8 __attribute__((ext_vector_type(8)));
typedef float f4_f32x16 __attribute__((ext_vector_type(16)));

// 16 two-bit codes -> 16 e2m1 nibbles of the counts (code 00 -> 0, 10 -> 1, 11 -> 2, 01 = missing / pad -> 0), nibble s = sample s
__device__ __forceinline__ uint2 nib16_counts(uint32_t w) {
    const uint32_t hi = (w >> 1) & 0x55555555u;
    const uint32_t v = hi + (hi & w);                       // per 2-bit field: the count
    auto spread = [](uint32_t y) {                          // 8 two-bit fields (16 bits) -> 8 nibbles holding field << 1
        y = (y | (y << 8)) & 0x00ff00ffu;
        y = (y | (y << 4)) & 0x0f0f0f0fu;
        y = (y | (y << 2)) & 0x33333333u;
        return y << 1;
    };
    return make_uint2(spread(v & 0xffffu), spread(v >> 16));
}

// nib[(tile * cnt + j) * 64 + 8 d ..] <- samples 128 tile + 16 d .. + 15 of SNP rows[r0 + j]; thread = (tile, j, d)
__global__ __launch_bounds__(256) void grm_nib_kernel(const uint8_t *__restrict__ p32, int64_t m_total, const int32_t *__restrict__ rows,
                                                      int64_t r0, int64_t cnt, uint8_t *__restrict__ nib) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= cnt * 8) return;
    const int tile = blockIdx.y;
    const int64_t j = id >> 3;
    const int d = (int)(id & 7);
    const int64_t rec = rows[r0 + j];
    const uint32_t w = *reinterpret_cast<const uint32_t *>(p32 + ((int64_t)tile * m_total + rec) * 32 + 4 * d);
    *reinterpret_cast<uint2 *>(nib + ((int64_t)tile * cnt + j) * 64 + 8 * d) = nib16_counts(w);
}

// MFMA operand of this lane (32 consecutive k of one position, e2m1) from a [k][position] nibble image: two transposed reads.
// `lane_base` = image + (k0 + 32 (lane >> 5) + (lane & 15)) * PITCH + (pos0 + 16 ((lane >> 4) & 1)) / 2
template <int PITCH>
__device__ __forceinline__ f4_i32x8 tr4_frag(const uint8_t *lane_base) {
    typedef __attribute__((address_space(3))) f4_i32x2 lds_i32x2;
    const f4_i32x2 a = __builtin_amdgcn_ds_read_tr4_b64_v2i32((lds_i32x2 *)(lane_base));
    const f4_i32x2 b = __builtin_amdgcn_ds_read_tr4_b64_v2i32((lds_i32x2 *)(lane_base + 16 * PITCH));
    f4_i32x8 r = {a.x, a.y, b.x, b.y, 0, 0, 0, 0};
    return r;
}

// grid: lower-triangle tiles (ti >= tj) of 256 x 256 from `tile_base`; SNPs [0, cnt) of the nibble image
__global__ __launch_bounds__(512, 2) void grm_fp4_kernel(const uint8_t *__restrict__ nib, int64_t cnt, int nt128, double *__restrict__ acc,
                                                         int64_t ld, const double *__restrict__ corr, int tile_base) {
    constexpr int TM = 256, WM = 128, WN = 64, BK = 128;
    constexpr int NWN = 4, NTHREADS = 512;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int PITCH = TM / 2 + 8;              // bytes per SNP row of an image: 16 rows x 8 bytes of a transposed read on disjoint banks
    constexpr int IMG = BK * PITCH;
    constexpr int DW = TM / 16;                    // 8-byte groups (16 samples) per SNP row of a panel
    constexpr int NL = BK * DW / NTHREADS;         // groups per thread, step and panel (4)
    constexpr int SET = 2 * IMG;
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * SET];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int t = blockIdx.x + tile_base;
    int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((int64_t)ti * (ti + 1) / 2 > t) --ti;
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - (int)((int64_t)ti * (ti + 1) / 2);

    const int d_of = tid % DW, kk_of = tid / DW;   // group of the panel row, SNP of the step (+ u * KSTRIDE)
    constexpr int KSTRIDE = NTHREADS / DW;
    const int recA128 = ti * 2 + (d_of >> 3), recB128 = tj * 2 + (d_of >> 3);
    const uint32_t maskA = recA128 < nt128 ? 0xffffffffu : 0u, maskB = recB128 < nt128 ? 0xffffffffu : 0u;
    const uint8_t *const baseA = nib + (int64_t)(recA128 < nt128 ? recA128 : nt128 - 1) * cnt * 64 + 8 * (d_of & 7);
    const uint8_t *const baseB = nib + (int64_t)(recB128 < nt128 ? recB128 : nt128 - 1) * cnt * 64 + 8 * (d_of & 7);

    uint2 wA[NL], wB[NL];
    auto load_payload = [&](int u, int64_t kbase) {          // clamped: rows beyond the chunk are zeroed at the store
        int64_t k = kbase + kk_of + u * KSTRIDE;
        k = k < cnt ? k : cnt - 1;
        wA[u] = *reinterpret_cast<const uint2 *>(baseA + k * 64);
        wB[u] = *reinterpret_cast<const uint2 *>(baseB + k * 64);
    };
    auto store_to = [&](uint8_t *base, int u, int64_t kbase) {
        const bool valid = kbase + kk_of + u * KSTRIDE < cnt;
        const int o = (kk_of + u * KSTRIDE) * PITCH + d_of * 8;
        const uint32_t ma = valid ? maskA : 0u, mb = valid ? maskB : 0u;
        *reinterpret_cast<uint2 *>(base + o) = make_uint2(wA[u].x & ma, wA[u].y & ma);
        *reinterpret_cast<uint2 *>(base + IMG + o) = make_uint2(wB[u].x & mb, wB[u].y & mb);
    };

    f4_f32x16 c[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[mi][ni][r] = 0.0f;

    const int h = lane >> 5;
    const int lane_off = (32 * h + (lane & 15)) * PITCH + 8 * ((lane >> 4) & 1);
    auto mfma_ks = [&](const uint8_t *base, int ks) {
        const uint8_t *sA = base + lane_off + (wm * WM) / 2 + ks * 64 * PITCH, *sB = base + IMG + lane_off + (wn * WN) / 2 + ks * 64 * PITCH;
        f4_i32x8 a[MI], b[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = tr4_frag<PITCH>(sA + mi * 16);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = tr4_frag<PITCH>(sB + ni * 16);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                c[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[mi], b[ni], c[mi][ni], 4, 4, 0, 127, 0, 127);
    };

    constexpr int KS = BK / 64;
#pragma unroll
    for (int u = 0; u < NL; ++u) load_payload(u, 0);
#pragma unroll
    for (int u = 0; u < NL; ++u) store_to(smem, u, 0);
#pragma unroll
    for (int u = 0; u < NL; ++u) load_payload(u, BK);
    __syncthreads();
    int cur = 0;
    for (int64_t kbase = 0; kbase < cnt; kbase += BK) {
        const uint8_t *rd = smem + cur * SET;
        uint8_t *wr = smem + (cur ^ 1) * SET;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            mfma_ks(rd, ks);
#pragma unroll
            for (int u = ks * NL / KS; u < (ks + 1) * NL / KS; ++u) {
                store_to(wr, u, kbase + BK);
                load_payload(u, kbase + 2 * BK);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        cur ^= 1;
    }

    // f64 merge (C/D layout of the 32 x 32 shapes: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); position = sample
    const bool add_corr = corr != nullptr;
    const double corr_b = add_corr ? corr[ld] : 0.0;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int64_t gj = (int64_t)tj * TM + wn * WN + ni * 32 + (lane & 31);
            const double corr_j = (add_corr && gj < ld) ? corr[gj] + corr_b : 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = (int64_t)ti * TM + wm * WM + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gi < ld && gj < ld) {
                    double *dst = acc + gi * ld + gj;
                    double v = (double)c[mi][ni][r];
                    if (add_corr) v += corr[gi] + corr_j;
                    *dst += v;
                }
            }
        }
}

// JXGPU_GRM_FP4=1: the count Gram of the 256-tile form on the fp4 pipes.  Default OFF: 1.29 x the int8 kernel at configs[2] (31.7 ->
// 24.6 ms, nibble pass included), but the north star quotes the GRM against the peak of the pipes it runs on, and 3.25 POP/s is
// 0.33 of the fp4 peak where the int8 kernel's 2.5 POP/s are 0.5 of its own
bool grm_fp4_enabled() {
    const char *e = getenv("JXGPU_GRM_FP4");
    return e && atoi(e) != 0;
}

// the `big` form of launch_grm_i8 (256 x 256 tiles `base256` .. + `ntl256` of the lower triangle) over the SNPs rows[r0 .. r1)
int launch_grm_fp4(hipStream_t st, const uint8_t *p32, int64_t m_total, const int32_t *rows, int64_t r0, int64_t r1, int nt128,
                   double *d_acc, int64_t ld, const double *corr, int64_t base256, int64_t ntl256) {
    // SNP chunks: the nibble image of a chunk (64 nt128 bytes per SNP) lives in a kept scratch block (slot 7, <= 6 GB: handed back by
    // jxg_scratch_trim like the eigensolver's), and 2^22 SNPs keep the f32 sums exact
    int64_t kc = (int64_t)(kScratchKeepBytes / (64 * (size_t)nt128));
    kc = std::min<int64_t>(kc, (int64_t)1 << 22);
    kc = std::max<int64_t>((kc / 128) * 128, 128);
    const int64_t cnt_all = r1 - r0;
    if (kc > cnt_all) kc = cnt_all;
    ScratchLease img;
    if (img.take(7, (size_t)kc * 64 * (size_t)nt128)) return 1;
    for (int64_t kb = r0; kb < r1; kb += kc) {
        const int64_t cnt = std::min(kc, r1 - kb);
        const int64_t nbx = (cnt * 8 + 255) / 256;
        if (nbx > 0x7fffffffLL || nt128 > 65535) return fail("launch_grm_fp4: grid too large");
        hipLaunchKernelGGL(grm_nib_kernel, dim3((unsigned)nbx, (unsigned)nt128), dim3(256), 0, st, p32, m_total, rows, kb, cnt,
                           img.as<uint8_t>());
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(grm_fp4_kernel, dim3((unsigned)ntl256), dim3(512), 0, st, img.as<uint8_t>(), cnt, nt128, d_acc, ld,
                           kb == r0 ? corr : nullptr, (int)base256);
        JX_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace jx
