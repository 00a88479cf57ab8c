// Eigenvector rotation G~ = G U on MFMA, 256 x 256 output tiles (the form for full-size blocks; k_rotate.hip keeps the
// 128 x 128 kernel for small problems, the block-diagonal route and the fused fixed-lambda epilogue).
//
// Reference: rotate_snp_block_with_ut_blas (src/stats/lmm.rs:728-784), design decode decode_centered_block_packed_f32
// (src/decode/decode.rs:192-271).   out[r, j] = sum_i g[r, i] * u_t[j, i]
//
// Same arithmetic as rotate_f16x2_kernel (fp16 hi / lo operands, f32 accumulation; rows that factor as beta + {0,1,2}
// without missing calls use their integer LUT and add beta * usum in the epilogue; a tile whose rows all qualify skips the
// A-lo plane), different shape: the 128 x 128 kernel spends as many LDS cycles (operand images written + fragments read) as
// MFMA cycles per k-step.  Here a 512-thread workgroup (8 waves, 2 x 4, 128 x 64 per wave) owns 256 SNP rows x 256
// eigenvector columns: LDS bytes per MFMA are halved, the images are double buffered (one barrier per k-step) and the
// staging of step k + 1 (payload decode by waves 0-3, U-plane chunks by all waves) is issued piecewise between the two
// halves of step k's MFMAs; its global loads have a whole step to land.  Images are [row][32 k] fp16 with a 64-byte pitch
// and the 16-byte chunk index XOR-swizzled by (row >> 2) & 3 (conflict-free ds_read_b128 fragments without padding: two
// image sets of four planes fit the 160 KB).
#include <hip/hip_fp16.h>

#include <stdlib.h>

#include <type_traits>

#include "jx_common.h"

namespace jx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int R2_T = 256;                 // tile rows = tile columns
constexpr int R2_BK = 32;                 // samples per k-step
constexpr int R2_IMG = R2_T * 64;         // one plane of one panel: 256 rows x 64 B
constexpr int R2_SET = 4 * R2_IMG;        // A hi | A lo | B hi | B lo

__device__ __forceinline__ int r2_chunk_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

__global__ __launch_bounds__(512, 2) void rotate256_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                           const int32_t *__restrict__ rows, int nrows,
                                                           const uint4 *__restrict__ lut16, const float *__restrict__ rowoff,
                                                           const float *__restrict__ usum, const __half *__restrict__ uhi,
                                                           const __half *__restrict__ ulo, int64_t npad, int n,
                                                           float out_scale, float *__restrict__ out, int64_t ldo,
                                                           const int32_t *__restrict__ sel) {
    // `sel` (optional): positions of this launch's rows inside the block (rows / lut16 / rowoff / out are indexed by position)
    extern __shared__ __attribute__((aligned(16))) uint8_t r2_smem[];       // 2 sets | seltab (64 B) | row offsets | positions
    uint32_t *seltab = reinterpret_cast<uint32_t *>(r2_smem + 2 * R2_SET);
    float *sOff = reinterpret_cast<float *>(r2_smem + 2 * R2_SET + 64);
    int *sPos = reinterpret_cast<int *>(sOff + R2_T);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // 1-D grid, XCD-aware (workgroup b runs on XCD b % 8 under round-robin dispatch): every XCD owns the column tiles
    // ct = 8 g + xcd and walks the row tiles fastest, so the workgroups resident on an XCD stream the same U planes
    const int nrt = (nrows + R2_T - 1) / R2_T;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int ct = (local / nrt) * 8 + xcd;
    if ((int64_t)ct * R2_T >= npad) return;
    const int j0 = ct * R2_T;
    const int r0 = (local % nrt) * R2_T;

    if (tid < 16) {
        const uint32_t c0 = tid & 3u, c1 = (tid >> 2) & 3u;
        seltab[tid] = (2u * c0) | ((2u * c0 + 1u) << 8) | ((2u * c1) << 16) | ((2u * c1 + 1u) << 24);
    }

    // ---- staging roles ------------------------------------------------------------------------------------------------
    // A panel: thread = (SNP row r0 + (tid >> 1), half tid & 1): one payload dword = 16 samples per k-step (every wave
    // decodes: no wave-dependent branch in the step loop)
    const int arow = tid >> 1, ahalf = tid & 1;
    const uint8_t *arec = p32;            // always a valid address (row 0 of the payload) for rows past the end
    uint4 L = make_uint4(0, 0, 0, 0);
    int row_exact = 1;
    {
        const int r = r0 + arow;
        float boff = 0.0f;
        int pos = -1;
        if (r < nrows) {
            pos = sel ? sel[r] : r;
            const int64_t rec = rows ? (int64_t)rows[pos] : (int64_t)pos;
            arec = p32 + rec * 32;
            L = lut16[pos];
            if (rowoff) {
                const float t = rowoff[pos];
                row_exact = (t == t) ? 1 : 0;
                boff = row_exact ? t : 0.0f;
            } else {
                row_exact = 0;
            }
        }
        if (ahalf == 0) {
            sOff[arow] = boff;
            sPos[arow] = pos;
        }
    }
    arec += 4 * ahalf;
    const bool tile_exact = __syncthreads_and(row_exact) != 0 && rowoff != nullptr;
    // all waves: U-plane chunks.  Chunk id = tid + 512 c (c = 0, 1): row = id >> 2 (eigenvector j0 + row), part = id & 3
    // (8 samples = 16 B); each id is loaded from the hi and from the lo plane.  Rows past npad never occur (npad is a
    // multiple of 128 and ct * 256 < npad, but the second half of the last tile may lie beyond: clamp + zero).
    const int brow0 = tid >> 2, bpart = tid & 3;
    const bool bok0 = j0 + brow0 < npad, bok1 = j0 + brow0 + 128 < npad;
    const __half *bsrc_h0 = uhi + (int64_t)(bok0 ? j0 + brow0 : 0) * npad + bpart * 8;
    const __half *bsrc_l0 = ulo + (int64_t)(bok0 ? j0 + brow0 : 0) * npad + bpart * 8;
    const __half *bsrc_h1 = uhi + (int64_t)(bok1 ? j0 + brow0 + 128 : 0) * npad + bpart * 8;
    const __half *bsrc_l1 = ulo + (int64_t)(bok1 ? j0 + brow0 + 128 : 0) * npad + bpart * 8;
    const uint32_t bm0 = bok0 ? 0xffffffffu : 0u, bm1 = bok1 ? 0xffffffffu : 0u;
    const int boff0 = r2_chunk_off(brow0, bpart), boff1 = r2_chunk_off(brow0 + 128, bpart);

    uint32_t wa = 0;
    u32x4 bh0, bl0, bh1, bl1;
    auto load_a = [&](int kstep) {        // payload dword of k-step `kstep`
        const int tile = kstep >> 2, sub = kstep & 3;
        wa = *reinterpret_cast<const uint32_t *>(arec + (int64_t)tile * m_total * 32 + sub * 8);
    };
    auto load_b0 = [&](int64_t kcol) {
        bh0 = *reinterpret_cast<const u32x4 *>(bsrc_h0 + kcol);
        bl0 = *reinterpret_cast<const u32x4 *>(bsrc_l0 + kcol);
    };
    auto load_b1 = [&](int64_t kcol) {
        bh1 = *reinterpret_cast<const u32x4 *>(bsrc_h1 + kcol);
        bl1 = *reinterpret_cast<const u32x4 *>(bsrc_l1 + kcol);
    };
    // the payload dword (16 samples = chunks 2 half, 2 half + 1 of row `arow`) -> hi (and lo) plane
    auto decode_half = [&](uint8_t *set, auto ex_tag) {
        constexpr bool EX = decltype(ex_tag)::value;
        const uint32_t w = wa;
        const int half = ahalf;
        uint32_t sl[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sl[i] = seltab[(w >> (4 * i)) & 15u];
        u32x4 h0, h1;
        h0.x = __builtin_amdgcn_perm(L.y, L.x, sl[0]);
        h0.y = __builtin_amdgcn_perm(L.y, L.x, sl[1]);
        h0.z = __builtin_amdgcn_perm(L.y, L.x, sl[2]);
        h0.w = __builtin_amdgcn_perm(L.y, L.x, sl[3]);
        h1.x = __builtin_amdgcn_perm(L.y, L.x, sl[4]);
        h1.y = __builtin_amdgcn_perm(L.y, L.x, sl[5]);
        h1.z = __builtin_amdgcn_perm(L.y, L.x, sl[6]);
        h1.w = __builtin_amdgcn_perm(L.y, L.x, sl[7]);
        *reinterpret_cast<u32x4 *>(set + r2_chunk_off(arow, 2 * half)) = h0;
        *reinterpret_cast<u32x4 *>(set + r2_chunk_off(arow, 2 * half + 1)) = h1;
        if constexpr (!EX) {
            u32x4 l0, l1;
            l0.x = __builtin_amdgcn_perm(L.w, L.z, sl[0]);
            l0.y = __builtin_amdgcn_perm(L.w, L.z, sl[1]);
            l0.z = __builtin_amdgcn_perm(L.w, L.z, sl[2]);
            l0.w = __builtin_amdgcn_perm(L.w, L.z, sl[3]);
            l1.x = __builtin_amdgcn_perm(L.w, L.z, sl[4]);
            l1.y = __builtin_amdgcn_perm(L.w, L.z, sl[5]);
            l1.z = __builtin_amdgcn_perm(L.w, L.z, sl[6]);
            l1.w = __builtin_amdgcn_perm(L.w, L.z, sl[7]);
            *reinterpret_cast<u32x4 *>(set + R2_IMG + r2_chunk_off(arow, 2 * half)) = l0;
            *reinterpret_cast<u32x4 *>(set + R2_IMG + r2_chunk_off(arow, 2 * half + 1)) = l1;
        }
    };
    auto store_b0 = [&](uint8_t *set) {
        u32x4 m = {bm0, bm0, bm0, bm0};
        *reinterpret_cast<u32x4 *>(set + 2 * R2_IMG + boff0) = bh0 & m;
        *reinterpret_cast<u32x4 *>(set + 3 * R2_IMG + boff0) = bl0 & m;
    };
    auto store_b1 = [&](uint8_t *set) {
        u32x4 m = {bm1, bm1, bm1, bm1};
        *reinterpret_cast<u32x4 *>(set + 2 * R2_IMG + boff1) = bh1 & m;
        *reinterpret_cast<u32x4 *>(set + 3 * R2_IMG + boff1) = bl1 & m;
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const int nk = (int)(npad / R2_BK);
    const int h = lane >> 5;
    const int frow = lane & 31;
    // fragment of k sub-step kk (16 samples): chunk 2 kk + h of row (tile row base + frow)
    auto mfma_half = [&](const uint8_t *set, int kk, auto ex_tag) {
        constexpr bool EX = decltype(ex_tag)::value;
        half8 ah[4], al[4], bhf[2], blf[2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int off = r2_chunk_off(wm * 128 + mi * 32 + frow, 2 * kk + h);
            ah[mi] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(set + off));
            if constexpr (!EX) al[mi] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(set + R2_IMG + off));
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int off = r2_chunk_off(wn * 64 + ni * 32 + frow, 2 * kk + h);
            bhf[ni] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(set + 2 * R2_IMG + off));
            blf[ni] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(set + 3 * R2_IMG + off));
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bhf[ni], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], blf[ni], acc[mi][ni], 0, 0, 0);
                if constexpr (!EX)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bhf[ni], acc[mi][ni], 0, 0, 0);
            }
    };

    // the whole pipeline in two instantiations (all rows exact: no A-lo plane, two products; else three), chosen once per
    // workgroup: a test of `tile_exact` around every third MFMA would cut the step loop into basic blocks
    auto run = [&](auto ex_tag) {
        // prologue: step 0 staged, step 1 in registers
        load_a(0);
        load_b0(0);
        load_b1(0);
        decode_half(r2_smem, ex_tag);
        store_b0(r2_smem);
        store_b1(r2_smem);
        {
            const int k1 = nk > 1 ? 1 : 0;
            load_a(k1);
            load_b0((int64_t)k1 * R2_BK);
            load_b1((int64_t)k1 * R2_BK);
        }
        __syncthreads();
        int cur = 0;
        for (int ks = 0; ks < nk; ++ks) {
            const uint8_t *rd = r2_smem + cur * R2_SET;
            uint8_t *wr = r2_smem + (cur ^ 1) * R2_SET;
            // step ks + 2 for the loads (clamped: the last two iterations re-load the last step, nothing reads it)
            const int kn = ks + 2 < nk ? ks + 2 : nk - 1;
            mfma_half(rd, 0, ex_tag);
            decode_half(wr, ex_tag);
            load_a(kn);
            store_b0(wr);
            load_b0((int64_t)kn * R2_BK);
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(rd, 1, ex_tag);
            store_b1(wr);
            load_b1((int64_t)kn * R2_BK);
            __syncthreads();
            cur ^= 1;
        }
    };
    if (tile_exact)
        run(std::true_type{});
    else
        run(std::false_type{});

#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int gj = j0 + wn * 64 + ni * 32 + frow;
            const float us = (usum && gj < n) ? usum[gj] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int gr = sPos[lr];
                if (gr >= 0 && gj < n) out[(int64_t)gr * ldo + gj] = fmaf(sOff[lr], us, acc[mi][ni][r] * out_scale);
            }
        }
}

// nonzero when the 256-tile kernel took the call
int launch_rotate256(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                     const void *d_lut16, const float *d_rowoff, const float *d_usum, const uint16_t *d_uhi,
                     const uint16_t *d_ulo, float out_scale, float *d_out, int64_t ld_out, const int32_t *d_sel, int *took) {
    // d_sel != NULL: `nrows` selected positions of a block (always this kernel: it gives the same bits as the 128-tile one)
    *took = 0;
    static const int env = getenv("JXGPU_ROT256") ? atoi(getenv("JXGPU_ROT256")) : 1;
    const int nt = num_tiles(n);
    const int64_t npad = (int64_t)nt * JXG_TILE;
    // full-size blocks only: at least a few rounds of 256 workgroups
    const int nct = (int)((npad + R2_T - 1) / R2_T), nrt = (nrows + R2_T - 1) / R2_T;
    if (!d_sel && (!env || (int64_t)nct * nrt < 2 * 256)) return 0;
    if (nrows <= 0) {
        *took = 1;
        return 0;
    }
    static bool attr = false;
    const int lds = 2 * R2_SET + 64 + 2048;
    if (!attr) {
        JX_HIP(hipFuncSetAttribute((const void *)rotate256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    dim3 grid((unsigned)(((nct + 7) / 8) * 8 * nrt));
    hipLaunchKernelGGL(rotate256_kernel, grid, dim3(512), lds, st, d_p32, m_total, d_rows, nrows, (const uint4 *)d_lut16,
                       d_rowoff, d_usum, (const __half *)d_uhi, (const __half *)d_ulo, npad, n, out_scale, d_out, ld_out,
                       d_sel);
    JX_LAUNCH_CHECK();
    *took = 1;
    return 0;
}

}  // namespace jx
