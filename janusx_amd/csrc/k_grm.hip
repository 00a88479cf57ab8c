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

#include "jx_common.h"

namespace jx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int G_BK = 32;        // SNPs per step
constexpr int G_PITCH = 320;    // bytes per SNP row of an LDS image (128 samples * 2 B + 64 B skew)
constexpr int G_IMG = G_BK * G_PITCH;

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
__device__ __forceinline__ void decode16_to_lds(uint32_t w, const uint4 L, const uint2 *__restrict__ seltab,
                                                uint8_t *dst_hi, uint8_t *dst_lo) {
    u32x4 h0, h1, l0, l1;
    {
        const uint2 s0 = seltab[w & 0xffu];
        const uint2 s1 = seltab[(w >> 8) & 0xffu];
        h0.x = __builtin_amdgcn_perm(L.y, L.x, s0.x);
        h0.y = __builtin_amdgcn_perm(L.y, L.x, s0.y);
        h0.z = __builtin_amdgcn_perm(L.y, L.x, s1.x);
        h0.w = __builtin_amdgcn_perm(L.y, L.x, s1.y);
        l0.x = __builtin_amdgcn_perm(L.w, L.z, s0.x);
        l0.y = __builtin_amdgcn_perm(L.w, L.z, s0.y);
        l0.z = __builtin_amdgcn_perm(L.w, L.z, s1.x);
        l0.w = __builtin_amdgcn_perm(L.w, L.z, s1.y);
    }
    {
        const uint2 s2 = seltab[(w >> 16) & 0xffu];
        const uint2 s3 = seltab[w >> 24];
        h1.x = __builtin_amdgcn_perm(L.y, L.x, s2.x);
        h1.y = __builtin_amdgcn_perm(L.y, L.x, s2.y);
        h1.z = __builtin_amdgcn_perm(L.y, L.x, s3.x);
        h1.w = __builtin_amdgcn_perm(L.y, L.x, s3.y);
        l1.x = __builtin_amdgcn_perm(L.w, L.z, s2.x);
        l1.y = __builtin_amdgcn_perm(L.w, L.z, s2.y);
        l1.z = __builtin_amdgcn_perm(L.w, L.z, s3.x);
        l1.w = __builtin_amdgcn_perm(L.w, L.z, s3.y);
    }
    *reinterpret_cast<u32x4 *>(dst_hi) = h0;
    *reinterpret_cast<u32x4 *>(dst_hi + 16) = h1;
    *reinterpret_cast<u32x4 *>(dst_lo) = l0;
    *reinterpret_cast<u32x4 *>(dst_lo + 16) = l1;
}

// MFMA operand (8 consecutive k for this lane's sample) from a [k][sample] image: two transposed reads.
__device__ __forceinline__ half8 tr_frag(const uint8_t *img_lane_base) {
    typedef __attribute__((address_space(3))) fp16x4 lds_fp16x4;
    const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4 *)(img_lane_base));
    const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4 *)(img_lane_base + 4 * G_PITCH));
    u32x4 r;
    const u32x2 ua = __builtin_bit_cast(u32x2, a);
    const u32x2 ub = __builtin_bit_cast(u32x2, b);
    r.x = ua.x;
    r.y = ua.y;
    r.z = ub.x;
    r.w = ub.y;
    return __builtin_bit_cast(half8, r);
}

__global__ __launch_bounds__(256, 2) void grm_f16x2_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                           const int32_t *__restrict__ rows,
                                                           const uint4 *__restrict__ lut16, int64_t k_begin,
                                                           int64_t k_end, int kchunk, double *__restrict__ acc,
                                                           int64_t ld, int use_atomic) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * G_IMG + 2048];
    uint8_t *sAh = smem;
    uint8_t *sAl = smem + G_IMG;
    uint8_t *sBh = smem + 2 * G_IMG;
    uint8_t *sBl = smem + 3 * G_IMG;
    uint2 *seltab = reinterpret_cast<uint2 *>(smem + 4 * G_IMG);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    seltab[tid] = make_selectors((uint32_t)tid);

    // lower-triangular tile pair (ti >= tj) from the linear block index
    const int t = blockIdx.x;
    int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((int64_t)ti * (ti + 1) / 2 > t) --ti;
    while ((int64_t)(ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - (int)((int64_t)ti * (ti + 1) / 2);

    const int64_t k0 = k_begin + (int64_t)blockIdx.y * kchunk;
    const int64_t k1 = (k0 + kchunk < k_end) ? (k0 + kchunk) : k_end;

    const int kk = tid >> 3;  // SNP within the step
    const int d = tid & 7;    // dword (16 samples) within the 128-sample tile
    const uint8_t *baseA = p32 + (int64_t)ti * m_total * 32 + 4 * d;
    const uint8_t *baseB = p32 + (int64_t)tj * m_total * 32 + 4 * d;

    uint32_t wA = 0, wB = 0;
    uint4 L = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](int64_t kbase) {
        const int64_t k = kbase + kk;
        if (k < k1) {
            const int64_t rec = rows ? (int64_t)rows[k] : k;
            wA = *reinterpret_cast<const uint32_t *>(baseA + rec * 32);
            wB = *reinterpret_cast<const uint32_t *>(baseB + rec * 32);
            L = lut16[k];
        } else {
            wA = 0;
            wB = 0;
            L = make_uint4(0, 0, 0, 0);
        }
    };

    floatx16 c[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[mi][ni][r] = 0.0f;

    // per-lane transposed-read geometry
    const int g = lane >> 4;          // 16-lane group
    const int h = lane >> 5;          // k half of the MFMA operand
    const int q = (lane & 15) >> 2;   // row of the 4x16 block this lane addresses
    const int pp = lane & 3;          // 4-column group this lane addresses
    const int lane_off = (8 * h + q) * G_PITCH + (16 * (g & 1) + 4 * pp) * 2;

    prefetch(k0);
    __syncthreads();  // selector table ready

    for (int64_t kbase = k0; kbase < k1; kbase += G_BK) {
        decode16_to_lds(wA, L, seltab, sAh + kk * G_PITCH + d * 32, sAl + kk * G_PITCH + d * 32);
        decode16_to_lds(wB, L, seltab, sBh + kk * G_PITCH + d * 32, sBl + kk * G_PITCH + d * 32);
        __syncthreads();
        prefetch(kbase + G_BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int off = ks * 16 * G_PITCH + lane_off + (wm * 64 + mi * 32) * 2;
                ah[mi] = tr_frag(sAh + off);
                al[mi] = tr_frag(sAl + off);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int off = ks * 16 * G_PITCH + lane_off + (wn * 64 + ni * 32) * 2;
                bh[ni] = tr_frag(sBh + off);
                bl[ni] = tr_frag(sBl + off);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    c[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], c[mi][ni], 0, 0, 0);
                    c[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], c[mi][ni], 0, 0, 0);
                    c[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], c[mi][ni], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // f64 merge: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int64_t gj = (int64_t)tj * JXG_TILE + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gi = (int64_t)ti * JXG_TILE + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                double *dst = acc + gi * ld + gj;
                const double v = (double)c[mi][ni][r];
                if (use_atomic) {
                    unsafeAtomicAdd(dst, v);
                } else {
                    *dst += v;
                }
            }
        }
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
float g_last_ms[4] = {0.f, 0.f, 0.f, 0.f};  // 0: GRM MFMA kernel(s), 1: rotation kernel, 2: reserved
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

extern "C" float jxg_last_kernel_ms(int which) {
    if (which < 0 || which >= 4) return 0.f;
    if (which == 1 && g_timer_pending[1] && g_rot_b) {
        if (hipEventSynchronize(g_rot_b) == hipSuccess) (void)hipEventElapsedTime(&g_last_ms[1], g_rot_a, g_rot_b);
        g_timer_pending[1] = 0;
    }
    return g_last_ms[which];
}

extern "C" int jxg_grm_accumulate(const uint8_t *d_p32, int64_t m_total, int n_sel, const int32_t *d_rows,
                                  const float *d_lut, int64_t mk, double *d_acc, int kchunk, int precision,
                                  void *stream) {
    if (mk <= 0) return 0;
    if (precision != 0) return fail("jxg_grm_accumulate: precision=1 (f32 MFMA) path not built yet");
    hipStream_t st = (hipStream_t)stream;
    const int nt = num_tiles(n_sel);
    const int64_t ld = (int64_t)nt * JXG_TILE;
    if (kchunk <= 0) kchunk = 8192;
    kchunk = ((kchunk + G_BK - 1) / G_BK) * G_BK;

    DevBuf lut16, flags;
    if (lut16.alloc(sizeof(uint4) * (size_t)mk)) return 1;
    if (flags.alloc(sizeof(int))) return 1;
    JX_HIP(hipMemsetAsync(flags.p, 0, sizeof(int), st));
    hipLaunchKernelGGL(lut_split_kernel, dim3((unsigned)((mk + 255) / 256)), dim3(256), 0, st, d_lut, mk,
                       lut16.as<uint4>(), 1.0f, flags.as<int>());
    JX_LAUNCH_CHECK();
    int hflag = 0;
    JX_HIP(hipMemcpyAsync(&hflag, flags.p, sizeof(int), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    if (hflag) return fail("jxg_grm_accumulate: design values exceed the fp16 split range (|z| > 3e4)");

    const int64_t ntiles = (int64_t)nt * (nt + 1) / 2;
    if (ntiles > 0x7fffffffLL) return fail("jxg_grm_accumulate: too many tiles");
    const int64_t nchunks = (mk + kchunk - 1) / kchunk;
    // Few tiles: spread SNP chunks over blockIdx.y with f64 atomics so the chip is filled.
    // Many tiles: one launch per chunk, the owning workgroup does a plain f64 read-modify-write.
    const bool atomic_mode = ntiles < 1024 && nchunks > 1;
    if (g_grm_ev.init()) return 1;
    JX_HIP(hipEventRecord(g_grm_ev.a, st));
    if (atomic_mode) {
        // shrink chunks if that is what it takes to reach ~4 workgroups per CU
        int64_t want = (4 * 256 + ntiles - 1) / ntiles;
        int64_t kc = kchunk;
        while ((mk + kc - 1) / kc < want && kc > 1024) kc /= 2;
        kc = ((kc + G_BK - 1) / G_BK) * G_BK;
        const int64_t ny = (mk + kc - 1) / kc;
        if (ny > 65535) return fail("jxg_grm_accumulate: too many chunks");
        hipLaunchKernelGGL(grm_f16x2_kernel, dim3((unsigned)ntiles, (unsigned)ny), dim3(256), 0, st, d_p32, m_total,
                           d_rows, lut16.as<uint4>(), (int64_t)0, mk, (int)kc, d_acc, ld, 1);
        JX_LAUNCH_CHECK();
    } else {
        for (int64_t c = 0; c < nchunks; ++c) {
            const int64_t kb = c * kchunk;
            const int64_t ke = (kb + kchunk < mk) ? kb + kchunk : mk;
            hipLaunchKernelGGL(grm_f16x2_kernel, dim3((unsigned)ntiles, 1), dim3(256), 0, st, d_p32, m_total,
                               d_rows, lut16.as<uint4>(), kb, ke, kchunk, d_acc, ld, 0);
            JX_LAUNCH_CHECK();
        }
    }
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
