// Eigenvector rotation G~ = G U on MFMA.
//
// Reference: rotate_snp_block_with_ut_blas (src/stats/lmm.rs:728-784, cblas_sgemm RowMajor NoTrans x Trans),
// pure-Rust rotate_snp_block_with_ut (src/stats/lmm.rs:520-552), design decode
// decode_centered_block_packed_f32 (src/decode/decode.rs:192-271).
//     out[r, j] = sum_i g[r, i] * u_t[j, i]
//
// Packed route (jxg_rotate_packed): A = design rows decoded on the fly from the 2-bit P32 payload through a
// per-SNP fp16 hi/lo LUT; B = U^T pre-split into two fp16 planes scaled by 2^scale_exp.  Three
// v_mfma_f32_32x32x16_f16 products (hi*hi + hi*lo + lo*hi) with f32 accumulation reproduce an f32 GEMM to
// ~2^-22 per operand.  Dense route (jxg_rotate_dense_f32): exact f32 MFMA (v_mfma_f32_32x32x2_f32).
#include <hip/hip_fp16.h>

#include <stdlib.h>

#include <type_traits>

#include "jx_common.h"

namespace jx {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int R_BK = 32;      // samples (k) per step
constexpr int R_PITCH = 80;   // bytes per row of a [row][k] fp16 image: 64 B + 16 B skew (conflict-free b128)
constexpr int R_IMG = 128 * R_PITCH;

__device__ __forceinline__ uint2 make_selectors_r(uint32_t byte) {
    const uint32_t c0 = byte & 3u, c1 = (byte >> 2) & 3u, c2 = (byte >> 4) & 3u, c3 = (byte >> 6) & 3u;
    uint2 s;
    s.x = (2u * c0) | ((2u * c0 + 1u) << 8) | ((2u * c1) << 16) | ((2u * c1 + 1u) << 24);
    s.y = (2u * c2) | ((2u * c2 + 1u) << 8) | ((2u * c3) << 16) | ((2u * c3 + 1u) << 24);
    return s;
}

// f32 U^T (n,n) -> fp16 planes (n_pad, n_pad): hi = f16(u * 2^e), lo = f16(u * 2^e - hi); zero padding.
__global__ __launch_bounds__(256) void ut_split_kernel(const float *__restrict__ ut, int n, int64_t npad,
                                                       __half *__restrict__ hi, __half *__restrict__ lo,
                                                       float scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npad * npad) return;
    const int64_t r = idx / npad, c = idx - r * npad;
    float v = 0.0f;
    if (r < n && c < n) v = ut[r * (int64_t)n + c] * scale;
    const __half h = __float2half_rn(v);
    const __half l = __float2half_rn(v - __half2float(h));
    hi[idx] = h;
    lo[idx] = l;
}

__global__ __launch_bounds__(256) void lut_split_r_kernel(const float *__restrict__ lut, int64_t mk,
                                                          uint4 *__restrict__ out, int *__restrict__ flags) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= mk) return;
    uint16_t hi[4], lo[4];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float v = lut[k * 4 + c];
        if (!(fabsf(v) <= 30000.0f)) bad = true;
        const __half h = __float2half_rn(v);
        const __half l = __float2half_rn(v - __half2float(h));
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


// Design rows that factor as beta + {0,1,2}: integer LUT (hi plane, lo = 0) and rowoff[k] = beta; every other row keeps
// the hi/lo split of its values and rowoff[k] = NaN.  A row qualifies when its three genotype values are
// beta + {0,1,2} in either allele orientation (the scan design: g - row mean, src/decode/decode.rs:192-271) and no
// selected sample carries the missing code (its imputed value is then never decoded).
__global__ __launch_bounds__(256) void lut_split_rows_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                             const int32_t *__restrict__ rows,
                                                             const float *__restrict__ lut, int64_t mk, int n_sel,
                                                             int nt128, uint4 *__restrict__ out,
                                                             float *__restrict__ rowoff, int *__restrict__ flags,
                                                             float *__restrict__ rowmiss, int miss_max) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= mk) return;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = lut[k * 4 + c];
    const float b = fminf(v[0], v[3]);
    const float c0 = v[0] - b, c2 = v[2] - b, c3 = v[3] - b;
    const float tol = 4e-6f;
    bool ok = fabsf(c2 - 1.0f) <= tol &&
              ((fabsf(c0) <= tol && fabsf(c3 - 2.0f) <= tol) || (fabsf(c0 - 2.0f) <= tol && fabsf(c3) <= tol)) &&
              fabsf(b) <= 4.0f;
    int nmiss = 0;
    if (ok) {
        const int64_t rec = rows ? (int64_t)rows[k] : k;
        for (int t = 0; t < nt128; ++t) {
            const uint4 *q = reinterpret_cast<const uint4 *>(p32 + ((int64_t)t * m_total + rec) * 32);
            const uint4 a = q[0], c = q[1];
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            const int valid = n_sel - t * 128;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                uint32_t miss = w[d] & ~(w[d] >> 1) & 0x55555555u;
                const int left = valid - d * 16;
                if (left <= 0) miss = 0;
                else if (left < 16) miss &= (1u << (2 * left)) - 1u;
                nmiss += __popc(miss);
            }
        }
        // a few missing calls: the row stays on the exact path (their code decodes to count 0 there) and
        // rot_miss_correct_kernel adds d * sum_{i missing} U[i, :] behind the rotation
        ok = (nmiss == 0) || (rowmiss != nullptr && nmiss <= miss_max && isfinite(v[1]));
    }
    float off = __builtin_nanf("");
    float dmiss = 0.0f;
    if (ok) {
        const float r0 = (c0 > 1.0f) ? 2.0f : 0.0f, r3 = 2.0f - r0;
        off = ((v[0] - r0) + (v[2] - 1.0f) + (v[3] - r3)) * (1.0f / 3.0f);
        // the int8 rotation (the only one this tolerance is used with) forms s (c U) + (offset + 2 [flipped]) usum with the
        // unflipped count c, which is 0 at a missing call: its value there is offset + r0
        if (nmiss > 0) dmiss = v[1] - (off + r0);
        v[0] = r0; v[1] = 0.0f; v[2] = 1.0f; v[3] = r3;
    }
    rowoff[k] = off;
    if (rowmiss) rowmiss[k] = dmiss;
    uint16_t hi[4], lo[4];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (!(fabsf(v[c]) <= 30000.0f)) bad = true;
        const __half h = __float2half_rn(v[c]);
        const __half l = __float2half_rn(v[c] - __half2float(h));
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

// usum[j] = sum_i u_t[j][i] (f64 accumulation), one wave per eigenvector row; zero beyond n
__global__ __launch_bounds__(256) void ut_rowsum_kernel(const float *__restrict__ ut, int n, int npad,
                                                        float *__restrict__ usum) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= npad) return;
    double a = 0.0;
    if (j < n)
        for (int i = lane; i < n; i += 64) a += (double)ut[(int64_t)j * n + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) usum[j] = (float)a;
}

// grid: x = column tile (eigenvector index j), y = row tile (SNP rows). 256 threads = 4 waves (2x2 of 64x64).
// waves 0-1 decode the A panel (128 SNP rows x 32 samples), waves 2-3 stage the two U planes.
// Rows whose design values are beta + {0,1,2} without missing calls (`jxg_lut_split_rows`: integer LUT, lo plane zero,
// beta in rowoff[r]; NaN marks a general row) contribute  out = (c U) + beta * usum,  usum[j] = sum_i u_t[j][i]:
// a tile whose 128 rows all qualify skips the A-lo plane (decode, LDS traffic and one of the three MFMA products).
// FUSE: instead of writing the rotated tile, reduce it against the fixed-lambda weights of its 128 eigenvector columns
// (fw = 1 / (s + lambda), fpy = W P y~, fwx = W X~, the state of jxg_fvlmm_prepare) and add the p + 2 partial sums of every
// SNP row into fsums[tile0 + column tile][row][0 .. p+1] = this tile's share of (sum w g~^2, g~.Py~, g~.WX~[k]); the finish
// kernel adds the column tiles in index order (no atomics: a fixed-lambda result is the same bits in every run).  The
// fixed-lambda scan (src/stats/fvlmm.rs:1691-1805) and the SparseLMM exact scan never see G~ in memory.  The tile goes through
// the LDS of the finished main loop, 64 rows at a time; four threads share a row (32 interleaved columns each, conflict-free
// reads).
constexpr int R_FP = 132;      // floats per row of the epilogue's LDS tile
constexpr int R_FMAXP = 8;     // covariates the fused epilogue supports
struct RotFuse {
    const float *fw, *fpy, *fwx;
    int fp;
    double *fsums;          // [column tiles of all calls][nrows][flds]
    int flds;
    int tile0;              // index of this call's first column tile in fsums
};

template <bool FUSE>
__global__ __launch_bounds__(256, 3) void rotate_f16x2_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                              const int32_t *__restrict__ rows, int nrows,
                                                              const uint4 *__restrict__ lut16,
                                                              const float *__restrict__ rowoff,
                                                              const float *__restrict__ usum,
                                                              const __half *__restrict__ uhi,
                                                              const __half *__restrict__ ulo, int64_t npad, int n,
                                                              float out_scale, float *__restrict__ out, int64_t ldo,
                                                              RotFuse F) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * R_IMG + 64 + 512];
    uint8_t *sAh = smem;
    uint8_t *sAl = smem + R_IMG;
    uint8_t *sBh = smem + 2 * R_IMG;
    uint8_t *sBl = smem + 3 * R_IMG;
    uint32_t *seltab = reinterpret_cast<uint32_t *>(smem + 4 * R_IMG);
    float *sOff = reinterpret_cast<float *>(smem + 4 * R_IMG + 64);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8 (round-robin dispatch).  Every XCD owns the eigenvector
    // column tiles ct = 8 g + xcd and walks the SNP row tiles of the block fastest, so the workgroups that are
    // co-resident on an XCD stream the same two U planes (16 KB per k-step, read from HBM once and then hit in
    // that XCD's L2) while each reads its own small payload panel (1 KB per k-step).
    const int nrt = (nrows + 127) >> 7;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int ct = (local / nrt) * 8 + xcd;
    if (ct * 128 >= ((n + 127) & ~127)) return;
    const int j0 = ct * 128;
    const int r0 = (local % nrt) * 128;

    if (tid < 16) seltab[tid] = make_selectors_r((uint32_t)tid).x;

    // ---- loader roles -------------------------------------------------------------------------
    const bool is_decoder = tid < 128;
    // decoder: thread = SNP row r0 + tid
    const uint8_t *arec = nullptr;
    uint4 L = make_uint4(0, 0, 0, 0);
    int row_exact = 1;
    if (is_decoder) {
        const int r = r0 + tid;
        float boff = 0.0f;
        if (r < nrows) {
            const int64_t rec = rows ? (int64_t)rows[r] : (int64_t)r;
            arec = p32 + rec * 32;
            L = lut16[r];
            if (rowoff) {
                const float t = rowoff[r];
                row_exact = (t == t) ? 1 : 0;
                boff = row_exact ? t : 0.0f;
            } else {
                row_exact = 0;
            }
        }
        sOff[tid] = boff;
    }
    const bool tile_exact = __syncthreads_and(row_exact) != 0 && rowoff != nullptr;
    // stager: u = tid - 128; chunk id = u + 128*c (c = 0..3): row = id >> 2, part = id & 3 (16 B each)
    const int u = tid - 128;

    uint2 wa = make_uint2(0, 0);
    u32x4 bh[4], bl[4];
    auto prefetch = [&](int kstep) {
        if (is_decoder) {
            if (arec) {
                const int tile = kstep >> 2, sub = kstep & 3;
                wa = *reinterpret_cast<const uint2 *>(arec + (int64_t)tile * m_total * 32 + sub * 8);
            }
        } else {
            const int64_t kcol = (int64_t)kstep * R_BK;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int id = u + 128 * c;
                const int row = id >> 2, part = id & 3;
                const int64_t off = (int64_t)(j0 + row) * npad + kcol + part * 8;
                bh[c] = *reinterpret_cast<const u32x4 *>(uhi + off);
                bl[c] = *reinterpret_cast<const u32x4 *>(ulo + off);
            }
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const int nk = (int)(npad / R_BK);
    const int h = lane >> 5;
    const int frag_off = (lane & 31) * R_PITCH + h * 16;

    prefetch(0);
    __syncthreads();

    for (int ks = 0; ks < nk; ++ks) {
        if (is_decoder) {
            // 8 payload bytes = 32 samples -> 64 B hi + 64 B lo in row `tid`
            uint8_t *dh = sAh + tid * R_PITCH;
            uint8_t *dl = sAl + tid * R_PITCH;
            const uint32_t ws[2] = {wa.x, wa.y};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const uint32_t w = ws[half];
                // nibble (two samples) -> v_perm selector: 16-entry table on 16 distinct banks, conflict-free
                uint32_t sl[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) sl[i] = seltab[(w >> (4 * i)) & 15u];
                u32x4 h0, h1, l0, l1;
                h0.x = __builtin_amdgcn_perm(L.y, L.x, sl[0]);
                h0.y = __builtin_amdgcn_perm(L.y, L.x, sl[1]);
                h0.z = __builtin_amdgcn_perm(L.y, L.x, sl[2]);
                h0.w = __builtin_amdgcn_perm(L.y, L.x, sl[3]);
                h1.x = __builtin_amdgcn_perm(L.y, L.x, sl[4]);
                h1.y = __builtin_amdgcn_perm(L.y, L.x, sl[5]);
                h1.z = __builtin_amdgcn_perm(L.y, L.x, sl[6]);
                h1.w = __builtin_amdgcn_perm(L.y, L.x, sl[7]);
                *reinterpret_cast<u32x4 *>(dh + half * 32) = h0;
                *reinterpret_cast<u32x4 *>(dh + half * 32 + 16) = h1;
                if (!tile_exact) {
                    l0.x = __builtin_amdgcn_perm(L.w, L.z, sl[0]);
                    l0.y = __builtin_amdgcn_perm(L.w, L.z, sl[1]);
                    l0.z = __builtin_amdgcn_perm(L.w, L.z, sl[2]);
                    l0.w = __builtin_amdgcn_perm(L.w, L.z, sl[3]);
                    l1.x = __builtin_amdgcn_perm(L.w, L.z, sl[4]);
                    l1.y = __builtin_amdgcn_perm(L.w, L.z, sl[5]);
                    l1.z = __builtin_amdgcn_perm(L.w, L.z, sl[6]);
                    l1.w = __builtin_amdgcn_perm(L.w, L.z, sl[7]);
                    *reinterpret_cast<u32x4 *>(dl + half * 32) = l0;
                    *reinterpret_cast<u32x4 *>(dl + half * 32 + 16) = l1;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int id = u + 128 * c;
                const int row = id >> 2, part = id & 3;
                *reinterpret_cast<u32x4 *>(sBh + row * R_PITCH + part * 16) = bh[c];
                *reinterpret_cast<u32x4 *>(sBl + row * R_PITCH + part * 16) = bl[c];
            }
        }
        __syncthreads();
        if (ks + 1 < nk) prefetch(ks + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 ah[2], al[2], bhf[2], blf[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int off = (wm * 64 + mi * 32) * R_PITCH + frag_off + kk * 32;
                ah[mi] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(sAh + off));
                if (!tile_exact) al[mi] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(sAl + off));
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int off = (wn * 64 + ni * 32) * R_PITCH + frag_off + kk * 32;
                bhf[ni] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(sBh + off));
                blf[ni] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4 *>(sBl + off));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bhf[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], blf[ni], acc[mi][ni], 0, 0, 0);
                    if (!tile_exact)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bhf[ni], acc[mi][ni], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    if constexpr (!FUSE) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int gj = j0 + wn * 64 + ni * 32 + (lane & 31);
                const float us = (usum && gj < n) ? usum[gj] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int gr = r0 + lr;
                    if (gr < nrows && gj < n) out[(int64_t)gr * ldo + gj] = fmaf(sOff[lr], us, acc[mi][ni][r] * out_scale);
                }
            }
    } else {
        float *tile = reinterpret_cast<float *>(smem);                   // [64][R_FP]
        float *wt = tile + 64 * R_FP;                                    // [128] w, [128] py, [128][fp] wx
        float *pyt = wt + 128;
        float *wxt = pyt + 128;
        const int fp = F.fp;
        if (tid < 128) {
            const int gj = j0 + tid;
            const bool okc = gj < n;
            wt[tid] = okc ? F.fw[gj] : 0.0f;
            pyt[tid] = okc ? F.fpy[gj] : 0.0f;
            for (int k = 0; k < fp; ++k) wxt[tid * fp + k] = okc ? F.fwx[(int64_t)gj * fp + k] : 0.0f;
        }
#pragma unroll 1
        for (int phase = 0; phase < 2; ++phase) {
            if (wm == phase) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        const int cj = wn * 64 + ni * 32 + (lane & 31);
                        const float us = (usum && j0 + cj < n) ? usum[j0 + cj] : 0.0f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int lr = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                            tile[lr * R_FP + cj] = fmaf(sOff[phase * 64 + lr], us, acc[mi][ni][r] * out_scale);
                        }
                    }
            }
            __syncthreads();
            {
                const int row = tid >> 2, q = tid & 3;
                double sv[2 + R_FMAXP];
#pragma unroll
                for (int k = 0; k < 2 + R_FMAXP; ++k) sv[k] = 0.0;
#pragma unroll 4
                for (int c = 0; c < 32; ++c) {
                    const int cj = q + 4 * c;
                    const double v = (double)tile[row * R_FP + cj];
                    sv[0] += (double)wt[cj] * v * v;
                    sv[1] += v * (double)pyt[cj];
#pragma unroll
                    for (int k = 0; k < R_FMAXP; ++k)
                        if (k < fp) sv[2 + k] += v * (double)wxt[cj * fp + k];
                }
                const int gr = r0 + phase * 64 + row;
#pragma unroll
                for (int k = 0; k < 2 + R_FMAXP; ++k) {
                    if (k < 2 + fp) {
                        double t = sv[k];
                        t += __shfl_xor(t, 1, 64);
                        t += __shfl_xor(t, 2, 64);
                        if (q == 0 && gr < nrows) F.fsums[((int64_t)(F.tile0 + ct) * nrows + gr) * F.flds + k] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Exact f32 NT GEMM on v_mfma_f32_32x32x2_f32:  C[M][N] = A[M][K] * B[N][K]^T  (row-major, any sizes).
// 128x128 tile, 4 waves (2x2 of 64x64), BK = 16.  LDS images [row][16 k] f32 with pitch 80 B.
// ---------------------------------------------------------------------------------------------------
constexpr int S_BK = 16;
constexpr int S_PITCH = 80;  // 16 floats + 16 B skew

__global__ __launch_bounds__(256, 2) void sgemm_nt_f32_kernel(const float *__restrict__ A, int M, int64_t lda,
                                                              const float *__restrict__ B, int N, int64_t ldb, int K,
                                                              float *__restrict__ C, int64_t ldc) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * 128 * S_PITCH];
    uint8_t *sA = smem;
    uint8_t *sB = smem + 128 * S_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;

    // loader: 128 rows x 16 k per operand = 512 float4 chunks -> 2 per thread per operand
    float4 ra[2], rb[2];
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int id = tid + 256 * c;
            const int row = id >> 2, part = id & 3;
            const int kc = k0 + part * 4;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            const int gm = m0 + row, gn = n0 + row;
            if (gm < M) {
                const float *p = A + (int64_t)gm * lda + kc;
                if (kc + 3 < K && ((((uintptr_t)p) & 15) == 0)) {
                    va = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (kc + 0 < K) va.x = p[0];
                    if (kc + 1 < K) va.y = p[1];
                    if (kc + 2 < K) va.z = p[2];
                    if (kc + 3 < K) va.w = p[3];
                }
            }
            if (gn < N) {
                const float *p = B + (int64_t)gn * ldb + kc;
                if (kc + 3 < K && ((((uintptr_t)p) & 15) == 0)) {
                    vb = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (kc + 0 < K) vb.x = p[0];
                    if (kc + 1 < K) vb.y = p[1];
                    if (kc + 2 < K) vb.z = p[2];
                    if (kc + 3 < K) vb.w = p[3];
                }
            }
            ra[c] = va;
            rb[c] = vb;
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const int h = lane >> 5;
    prefetch(0);
    for (int k0 = 0; k0 < K; k0 += S_BK) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int id = tid + 256 * c;
            const int row = id >> 2, part = id & 3;
            *reinterpret_cast<float4 *>(sA + row * S_PITCH + part * 16) = ra[c];
            *reinterpret_cast<float4 *>(sB + row * S_PITCH + part * 16) = rb[c];
        }
        __syncthreads();
        if (k0 + S_BK < K) prefetch(k0 + S_BK);
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            // lane (row, h) reads 4 consecutive k at kq*8 + 4*h; MFMA t pairs k = {kq*8+t, kq*8+4+t}
            float4 fa[2], fb[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                fa[mi] = *reinterpret_cast<const float4 *>(sA + (wm * 64 + mi * 32 + (lane & 31)) * S_PITCH +
                                                           kq * 32 + h * 16);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                fb[ni] = *reinterpret_cast<const float4 *>(sB + (wn * 64 + ni * 32 + (lane & 31)) * S_PITCH +
                                                           kq * 32 + h * 16);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].x, fb[ni].x, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].y, fb[ni].y, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].z, fb[ni].z, acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].w, fb[ni].w, acc[mi][ni], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int gj = n0 + wn * 64 + ni * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gr < M && gj < N) C[(int64_t)gr * ldc + gj] = acc[mi][ni][r];
            }
        }
}

}  // namespace jx

using namespace jx;

namespace jx {
int launch_rotate256(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                     const void *d_lut16, const float *d_rowoff, const float *d_usum, const uint16_t *d_uhi,
                     const uint16_t *d_ulo, float out_scale, float *d_out, int64_t ld_out, const int32_t *d_sel, int *took);
int launch_rotate_i8(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const int32_t *d_sel,
                     int nsel, const void *d_lut16, const float *d_rowoff, const float *d_usum, const int8_t *d_q,
                     const float *d_umax, float *d_out, int64_t ld_out);
extern float g_last_ms[24];
extern int g_timer_pending[4];
hipEvent_t g_rot_a = nullptr, g_rot_b = nullptr;
}  // namespace jx

extern "C" int jxg_ut_split(const float *d_ut, int n, uint16_t *d_hi, uint16_t *d_lo, int scale_exp,
                            void *stream) {
    const int64_t npad = (int64_t)num_tiles(n) * JXG_TILE;
    const int64_t total = npad * npad;
    const int64_t blocks = (total + 255) / 256;
    if (blocks * 256 > 0xffffffffLL) return fail("jxg_ut_split: grid too large (n beyond 65 408)");
    hipLaunchKernelGGL(ut_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_ut, n, npad,
                       (__half *)d_hi, (__half *)d_lo, ldexpf(1.0f, scale_exp));
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_lut_split(const float *d_lut, int64_t mk, void *d_lut16, void *stream) {
    if (mk <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int *flags = nullptr;
    async_pool_keep();
    JX_HIP(hipMallocAsync((void **)&flags, sizeof(int), st));
    JX_HIP(hipMemsetAsync(flags, 0, sizeof(int), st));
    hipLaunchKernelGGL(lut_split_r_kernel, dim3((unsigned)((mk + 255) / 256)), dim3(256), 0, st, d_lut, mk,
                       (uint4 *)d_lut16, flags);
    JX_LAUNCH_CHECK();
    int hflag = 0;
    JX_HIP(hipMemcpyAsync(&hflag, flags, sizeof(int), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    JX_HIP(hipFreeAsync(flags, st));
    if (hflag) return fail("design values exceed the fp16 split range (|z| > 3e4)");
    return 0;
}

// Largest number of missing calls with which a design row still takes the exact (int8) rotation + the gather correction
// (rot_miss_correct_kernel).  The correction reads one row of U (4 n bytes) per missing call -- 19 ns at n = 20 000 --, the
// fp16 kernel it avoids costs ~1 us per row there, but ALSO a fixed ~18 ms per launch, so a block should not be split between
// the two for a handful of rows: the decision is made for the whole scan from the mean number of missing calls per row.  Up to
// n / 300 of them (0.33 %) every row with <= 256 missing calls keeps the exact path with the gather correction (measured at
// n = 20 000, m = 200 000: rotation 288 / 305 / 358 / 397 ms at 0.15 / 0.2 / 0.3 / 0.4 % missing calls against 388 for the dense
// form below; n / 800 until the end of round 4, when the alternative was the fp16 kernel), beyond that the dense form takes over.
// JXGPU_ROT_MISS_MAX overrides the limit (0: off).
constexpr int JXG_ROT_MISS_DENSE = 1 << 30;      // "no limit": the dense form of the correction (any value > 256 means that)
extern "C" int jxg_rot_miss_max(int n, double mean_missing_per_row) {
    const char *e = getenv("JXGPU_ROT_MISS_MAX");
    if (e) {
        int v = atoi(e);
        return v < 0 ? 0 : (v > 256 ? 256 : v);
    }
    if (!(mean_missing_per_row > 0.0)) return 0;
    if (mean_missing_per_row <= (double)n / 300.0) return 256;
    // beyond: EVERY affine row keeps the exact path, and the missing-call term is one more int8 product with the indicator of the
    // missing calls (jxg_rotate_missing_dense) instead of a gather per call -- the cost of the fp16 kernel these rows took before
    // (two int8 passes against three fp16 products), but exact.  JXGPU_ROT_MISS_DENSE=0: the fp16 kernel as before.
    const char *dn = getenv("JXGPU_ROT_MISS_DENSE");
    if (dn && atoi(dn) == 0) return 0;
    return JXG_ROT_MISS_DENSE;
}

extern "C" int jxg_lut_split_rows_m(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const float *d_lut,
                                    int64_t mk, void *d_lut16, float *d_rowoff, float *d_rowmiss, int miss_max, void *stream);

extern "C" int jxg_lut_split_rows(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows,
                                  const float *d_lut, int64_t mk, void *d_lut16, float *d_rowoff, void *stream) {
    return jxg_lut_split_rows_m(d_p32, m_total, n, d_rows, d_lut, mk, d_lut16, d_rowoff, nullptr, 0, stream);
}

// d_rowmiss (mk) f32 or NULL: d of the rows that keep the exact path although they hold 1 .. miss_max missing calls (0 for
// every other row); see jxg_rotate_missing_correct.
extern "C" int jxg_lut_split_rows_m(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const float *d_lut,
                                    int64_t mk, void *d_lut16, float *d_rowoff, float *d_rowmiss, int miss_max, void *stream) {
    if (mk <= 0) return 0;
    // <= 256: the capacity of the gather form's per-row list (rot_miss_correct_kernel); > 256 = "no limit" (the dense form of the
    // missing-call term, jxg_rotate_missing_dense, needs no list): passed through
    hipStream_t st = (hipStream_t)stream;
    int *flags = nullptr;
    async_pool_keep();
    JX_HIP(hipMallocAsync((void **)&flags, sizeof(int), st));
    JX_HIP(hipMemsetAsync(flags, 0, sizeof(int), st));
    hipLaunchKernelGGL(lut_split_rows_kernel, dim3((unsigned)((mk + 255) / 256)), dim3(256), 0, st, d_p32, m_total,
                       d_rows, d_lut, mk, n, num_tiles(n), (uint4 *)d_lut16, d_rowoff, flags, d_rowmiss, miss_max);
    JX_LAUNCH_CHECK();
    int hflag = 0;
    JX_HIP(hipMemcpyAsync(&hflag, flags, sizeof(int), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    JX_HIP(hipFreeAsync(flags, st));
    if (hflag) return fail("design values exceed the fp16 split range (|z| > 3e4)");
    return 0;
}

// out[r][j] += d_r * sum_{i : sample i of row r is missing} usamp[i][j]   (usamp = U, sample-major: usamp[i][j] = u_t[j][i]).
// One workgroup per row with d_r != 0: the missing samples are listed in sample order (count, scan, fill: no atomics), then
// the threads walk the eigenvector index j and add the listed rows of U (coalesced), f64 sums.
constexpr int RM_MAXLIST = 256;
__global__ __launch_bounds__(256) void rot_miss_correct_kernel(const uint32_t *__restrict__ p32, int64_t m_total, int n,
                                                               int nt, const int32_t *__restrict__ rows,
                                                               const float *__restrict__ rowmiss,
                                                               const float *__restrict__ usamp, float *__restrict__ out,
                                                               int64_t ld) {
    const int r = blockIdx.x;
    const float d = rowmiss[r];
    if (d == 0.0f) return;                                  // uniform
    __shared__ int cnt[257];
    __shared__ int list[RM_MAXLIST];
    const int tid = threadIdx.x;
    const int64_t rec = rows ? (int64_t)rows[r] : (int64_t)r;
    const int ndw = nt * 8, per = (ndw + 255) / 256;
    const int q0 = tid * per, q1 = min(ndw, q0 + per);
    int c = 0;
    for (int q = q0; q < q1; ++q) {
        const uint32_t w = p32[((int64_t)(q >> 3) * m_total + rec) * 8 + (q & 7)];
        uint32_t miss = w & ~(w >> 1) & 0x55555555u;
        const int left = n - q * 16;
        if (left < 16) miss &= (left <= 0) ? 0u : ((1u << (2 * left)) - 1u);
        c += __popc(miss);
    }
    cnt[tid + 1] = c;
    if (tid == 0) cnt[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int t = 1; t <= 256; ++t) cnt[t] += cnt[t - 1];
    __syncthreads();
    int at = cnt[tid];
    const int total = min(cnt[256], RM_MAXLIST);
    for (int q = q0; q < q1; ++q) {
        const uint32_t w = p32[((int64_t)(q >> 3) * m_total + rec) * 8 + (q & 7)];
        uint32_t miss = w & ~(w >> 1) & 0x55555555u;
        const int left = n - q * 16;
        if (left < 16) miss &= (left <= 0) ? 0u : ((1u << (2 * left)) - 1u);
        while (miss) {
            const int b = __ffs(miss) - 1;
            miss &= miss - 1;
            if (at < RM_MAXLIST) list[at] = q * 16 + (b >> 1);
            ++at;
        }
    }
    __syncthreads();
    for (int j = tid; j < n; j += 256) {
        double acc = 0.0;
        int e = 0;
        for (; e + 4 <= total; e += 4) {
            const float a0 = usamp[(int64_t)list[e] * n + j], a1 = usamp[(int64_t)list[e + 1] * n + j];
            const float a2 = usamp[(int64_t)list[e + 2] * n + j], a3 = usamp[(int64_t)list[e + 3] * n + j];
            acc += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        }
        for (; e < total; ++e) acc += (double)usamp[(int64_t)list[e] * n + j];
        out[(int64_t)r * ld + j] += (float)((double)d * acc);
    }
}

__global__ __launch_bounds__(256) void transpose_f32_kernel(const float *__restrict__ src, int n, float *__restrict__ dst) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (by + r < n && bx + tx < n) tile[r][tx] = src[(int64_t)(by + r) * n + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (bx + r < n && by + tx < n) dst[(int64_t)(bx + r) * n + by + tx] = tile[tx][r];
}

// dst = src' for an (n, n) f32 matrix: the sample-major copy of U the correction below gathers from
extern "C" int jxg_transpose_f32(const float *d_src, int n, float *d_dst, void *stream) {
    if (n <= 0) return 0;
    const unsigned nb = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL(transpose_f32_kernel, dim3(nb, nb), dim3(256), 0, (hipStream_t)stream, d_src, n, d_dst);
    JX_LAUNCH_CHECK();
    return 0;
}

// Behind the rotation of a block of rows: adds the missing-call term of the rows jxg_lut_split_rows_m kept on the exact path
// (d_rowmiss[r] != 0).  d_usamp (n, n) f32 = U with one ROW per sample (jxg_transpose_f32 of u_t); d_out (nrows, ld_out).
extern "C" int jxg_rotate_missing_correct(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                          const float *d_rowmiss, const float *d_usamp, float *d_out, int64_t ld_out,
                                          void *stream) {
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(rot_miss_correct_kernel, dim3((unsigned)nrows), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *)d_p32, m_total, n, num_tiles(n), d_rows, d_rowmiss, d_usamp, d_out, ld_out);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_ut_rowsum(const float *d_ut, int n, float *d_usum, void *stream) {
    const int npad = num_tiles(n) * JXG_TILE;
    hipLaunchKernelGGL(ut_rowsum_kernel, dim3((unsigned)((npad + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_ut, n,
                       npad, d_usum);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_rotate_packed16x(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                    const void *d_lut16, const float *d_rowoff, const float *d_usum,
                                    const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp, float *d_out,
                                    void *stream);

extern "C" int jxg_rotate_packed16(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                   const void *d_lut16, const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp,
                                   float *d_out, void *stream) {
    return jxg_rotate_packed16x(d_p32, m_total, n, d_rows, nrows, d_lut16, nullptr, nullptr, d_uhi, d_ulo, scale_exp,
                                d_out, stream);
}

extern "C" int jxg_rotate_packed16x_ld(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                       const void *d_lut16, const float *d_rowoff, const float *d_usum,
                                       const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp, float *d_out,
                                       int64_t ld_out, void *stream);

extern "C" int jxg_rotate_packed16x(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                    const void *d_lut16, const float *d_rowoff, const float *d_usum,
                                    const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp, float *d_out,
                                    void *stream) {
    return jxg_rotate_packed16x_ld(d_p32, m_total, n, d_rows, nrows, d_lut16, d_rowoff, d_usum, d_uhi, d_ulo, scale_exp,
                                   d_out, (int64_t)n, stream);
}

// ld_out >= n: row pitch of d_out in floats (a block of eigenvector columns of a wider rotated-row buffer: the
// block-diagonal rotation of the sparse-GRM routes)
static int rotate_core(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows, const void *d_lut16,
                       const float *d_rowoff, const float *d_usum, const uint16_t *d_uhi, const uint16_t *d_ulo,
                       int scale_exp, const int8_t *d_q, const float *d_umax, const int32_t *d_sel_exact, int n_exact,
                       const int32_t *d_sel_rest, int n_rest, float *d_out, int64_t ld_out, void *stream);

extern "C" int jxg_rotate_packed16x_ld(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                       const void *d_lut16, const float *d_rowoff, const float *d_usum,
                                       const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp, float *d_out,
                                       int64_t ld_out, void *stream) {
    return rotate_core(d_p32, m_total, n, d_rows, nrows, d_lut16, d_rowoff, d_usum, d_uhi, d_ulo, scale_exp, nullptr, nullptr,
                       nullptr, 0, nullptr, 0, d_out, ld_out, stream);
}

// The same with the int8 planes of the eigenvectors (jxg_ut_quant3) and the block's rows split by the caller into the
// positions d_sel_exact (rows that factor as beta + {0,1,2} without missing calls: rowoff finite; int8 matrix pipes,
// k_rotate_i8.hip) and d_sel_rest (all other rows: fp16 kernel).  A row's path does not depend on its neighbours.
extern "C" int jxg_rotate_packed16x_q(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                      const void *d_lut16, const float *d_rowoff, const float *d_usum,
                                      const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp, const int8_t *d_q,
                                      const float *d_umax, const int32_t *d_sel_exact, int n_exact,
                                      const int32_t *d_sel_rest, int n_rest, float *d_out, void *stream) {
    if (n_exact < 0 || n_rest < 0 || n_exact + n_rest != nrows) return fail("jxg_rotate_packed16x_q: the two row lists must cover the block");
    return rotate_core(d_p32, m_total, n, d_rows, nrows, d_lut16, d_rowoff, d_usum, d_uhi, d_ulo, scale_exp, d_q, d_umax,
                       d_sel_exact, n_exact, d_sel_rest, n_rest, d_out, (int64_t)n, stream);
}

static int rotate_core(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows, const void *d_lut16,
                       const float *d_rowoff, const float *d_usum, const uint16_t *d_uhi, const uint16_t *d_ulo,
                       int scale_exp, const int8_t *d_q, const float *d_umax, const int32_t *d_sel_exact, int n_exact,
                       const int32_t *d_sel_rest, int n_rest, float *d_out, int64_t ld_out, void *stream) {
    if (ld_out < n) return fail("jxg_rotate_packed16x_ld: ld_out < n");
    if (nrows <= 0) return 0;
    static const int exact_env = getenv("JXGPU_ROT_EXACT") ? atoi(getenv("JXGPU_ROT_EXACT")) : 1;
    if (!d_usum) d_rowoff = nullptr;
    if (!exact_env && d_rowoff) return fail("JXGPU_ROT_EXACT=0 must be set before the design LUTs are split");
    hipStream_t st = (hipStream_t)stream;
    const int nt = num_tiles(n);
    const int64_t npad = (int64_t)nt * JXG_TILE;
    if (!g_rot_a) {
        JX_HIP(hipEventCreate(&g_rot_a));
        JX_HIP(hipEventCreate(&g_rot_b));
    }
    const int nrt = (nrows + 127) / 128;
    dim3 grid((unsigned)(((nt + 7) / 8) * 8 * nrt));
    JX_HIP(hipEventRecord(g_rot_a, st));
    int took = 0;
    if (d_q) {
        // exact rows on the int8 planes, the rest on the fp16 kernel, each as a compacted list of positions
        if (!d_rowoff || !exact_env) return fail("jxg_rotate_packed16x_q needs the exact-row LUTs (jxg_lut_split_rows)");
        if (launch_rotate_i8(st, d_p32, m_total, n, d_rows, d_sel_exact, n_exact, d_lut16, d_rowoff, d_usum, d_q, d_umax, d_out,
                             ld_out))
            return 1;
        if (n_rest > 0) {
            if (!d_sel_rest) return fail("jxg_rotate_packed16x_q: the list of the non-exact rows is missing");
            if (launch_rotate256(st, d_p32, m_total, n, d_rows, n_rest, d_lut16, d_rowoff, d_usum, d_uhi, d_ulo,
                                 ldexpf(1.0f, -scale_exp), d_out, ld_out, d_sel_rest, &took))
                return 1;
        }
        took = 1;   // every row of the block has been written by one of the two kernels: no 128-tile fallback behind them
        g_last_ms[14] = (float)((n_exact > 0) + (n_rest > 0));      // rotation kernels launched for this block
        g_last_ms[13] = (float)((double)n_exact / (double)nrows);   // share of the rows on the int8 kernel
    } else if (launch_rotate256(st, d_p32, m_total, n, d_rows, nrows, d_lut16, d_rowoff, d_usum, d_uhi, d_ulo,
                                ldexpf(1.0f, -scale_exp), d_out, ld_out, nullptr, &took)) {   // full-size blocks: 256 x 256 tiles
        return 1;
    }
    if (!took) {
        hipLaunchKernelGGL(rotate_f16x2_kernel<false>, grid, dim3(256), 0, st, d_p32, m_total, d_rows, nrows,
                           (const uint4 *)d_lut16, d_rowoff, d_usum, (const __half *)d_uhi, (const __half *)d_ulo, npad, n,
                           ldexpf(1.0f, -scale_exp), d_out, ld_out, RotFuse{});
        JX_LAUNCH_CHECK();
    }
    JX_HIP(hipEventRecord(g_rot_b, st));
    g_timer_pending[1] = 1;
    return 0;
}

// Rotation with the fused fixed-lambda reduction (rotate_f16x2_kernel<true>): G~ is not written; column tile t of this call
// (128 eigenvector columns, t < jxg_num_tiles(n)) writes its share of (sum_j w_j g~_rj^2, sum_j g~_rj py_j, sum_j g~_rj wx_jk)
// to d_part[tile0 + t][r][0 .. p+1] (d_w, d_py (n) f32, d_wx (n, p) f32: the state of jxg_fvlmm_prepare restricted to the
// columns of d_uhi / d_ulo).  d_part: (tiles of all calls, nrows, lds_sums >= p + 2) f64; jxg_fvlmm_finish_dev adds the tiles in
// index order and computes the statistics; the diagonal blocks of the block route are successive calls with tile0 advancing.
// p <= 8.
extern "C" int jxg_rotate_packed16x_fused(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                          const void *d_lut16, const float *d_rowoff, const float *d_usum,
                                          const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp, const float *d_w,
                                          const float *d_py, const float *d_wx, int p, double *d_sums, int lds_sums,
                                          int tile0, void *stream) {
    if (p < 1 || p > R_FMAXP || lds_sums < p + 2) return fail("jxg_rotate_packed16x_fused: p out of range (1..8)");
    if (nrows <= 0) return 0;
    if (!d_usum) d_rowoff = nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int nt = num_tiles(n);
    const int64_t npad = (int64_t)nt * JXG_TILE;
    if (!g_rot_a) {
        JX_HIP(hipEventCreate(&g_rot_a));
        JX_HIP(hipEventCreate(&g_rot_b));
    }
    const int nrt = (nrows + 127) / 128;
    dim3 grid((unsigned)(((nt + 7) / 8) * 8 * nrt));
    JX_HIP(hipEventRecord(g_rot_a, st));
    hipLaunchKernelGGL(rotate_f16x2_kernel<true>, grid, dim3(256), 0, st, d_p32, m_total, d_rows, nrows,
                       (const uint4 *)d_lut16, d_rowoff, d_usum, (const __half *)d_uhi, (const __half *)d_ulo, npad, n,
                       ldexpf(1.0f, -scale_exp), nullptr, 0, RotFuse{d_w, d_py, d_wx, p, d_sums, lds_sums, tile0});
    JX_LAUNCH_CHECK();
    JX_HIP(hipEventRecord(g_rot_b, st));
    g_timer_pending[1] = 1;
    return 0;
}

extern "C" int jxg_rotate_packed(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                 const float *d_lut, const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp,
                                 float *d_out, void *stream) {
    if (nrows <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    void *lut16 = nullptr;
    async_pool_keep();
    JX_HIP(hipMallocAsync(&lut16, sizeof(uint4) * (size_t)nrows, st));
    int rc = jxg_lut_split(d_lut, nrows, lut16, stream);
    if (!rc) rc = jxg_rotate_packed16(d_p32, m_total, n, d_rows, nrows, lut16, d_uhi, d_ulo, scale_exp, d_out, stream);
    (void)hipFreeAsync(lut16, st);
    return rc;
}

extern "C" int jxg_rotate_dense_f32(const float *d_g, int nrows, int n, const float *d_ut, float *d_out,
                                    void *stream) {
    if (nrows <= 0) return 0;
    dim3 grid((n + 127) / 128, (nrows + 127) / 128);
    hipLaunchKernelGGL(sgemm_nt_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_g, nrows, (int64_t)n, d_ut, n,
                       (int64_t)n, n, d_out, (int64_t)n);
    JX_LAUNCH_CHECK();
    return 0;
}
