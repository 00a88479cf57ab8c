// 2-bit genotype payload kernels: re-tiling into the P32 layout and per-SNP popcounts.
// Reference semantics: src/math/bedmath.rs:20-27 (codes), src/io/gfreader.rs:1378-1395 (counts).
#include <algorithm>

#include <stdlib.h>

#include "jx_common.h"

namespace jx {

// One thread produces one dword (16 samples) of the P32 image.
//   dst[(tile * m_out + j) * 32 + 4*d .. +3]  <- samples tile*128 + 16*d .. +15 of source row row_idx[j]
// Identity sample order + 4-byte-aligned source rows take the word-copy path; everything else gathers.
__global__ __launch_bounds__(256) void repack_p32_kernel(const uint8_t *__restrict__ src, int64_t bps, int n_src,
                                                         const int32_t *__restrict__ sample_idx, int n_sel,
                                                         const int64_t *__restrict__ row_idx, int64_t m_out,
                                                         uint32_t *__restrict__ dst, int nt) {
    // grid: x over the dwords of one tile, y = tile (a dispatch holds at most 2^32 work-items per dimension: the flat form
    // silently lost the tail of a 50 GB payload)
    const int64_t per_tile = m_out * 8;  // dwords per tile
    const int64_t rem = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rem >= per_tile) return;
    const int tile = (int)blockIdx.y;
    const int64_t gid = (int64_t)tile * per_tile + rem;
    const int64_t j = rem >> 3;
    const int d = (int)(rem & 7);
    const int64_t srow = row_idx ? row_idx[j] : j;
    const uint8_t *row = src + srow * bps;
    const int s0 = tile * JXG_TILE + d * 16;
    uint32_t w = 0;
    if (sample_idx == nullptr) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int s = s0 + 4 * b;  // first sample of this byte
            uint32_t byte;
            if (s + 3 < n_sel) {
                byte = row[s >> 2];
            } else {
                byte = 0x55u;  // all missing
                if (s < n_sel) {
                    const uint32_t raw = row[s >> 2];
                    const int valid = n_sel - s;  // 1..3 valid samples
                    const uint32_t mask = (1u << (2 * valid)) - 1u;
                    byte = (raw & mask) | (0x55u & ~mask);
                }
            }
            w |= byte << (8 * b);
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int s = s0 + i;
            uint32_t code = 1u;  // missing
            if (s < n_sel) {
                const int sid = sample_idx[s];
                code = (row[sid >> 2] >> (2 * (sid & 3))) & 3u;
            }
            w |= code << (2 * i);
        }
    }
    dst[gid] = w;
}

// ---- sample subsets: window form ------------------------------------------------------------------------------------------
// With a sample list the kernel above gathers every code on its own: 16 index loads + 16 byte loads per output dword, 0.34 TB/s
// (151 ms for the 40 GB training image of BASELINE configs[4]).  The 16 samples of an output dword are the same for every SNP, and
// for any dense subset (CV folds, a phenotyped subset) they sit inside a few source bytes: one DESCRIPTOR per (tile, dword) -- first
// source byte and the 16 positions relative to it, built once per call -- turns the gather into three dword loads of the row and 16
// shifts of a 64-bit window.  Dwords whose samples span more than 8 source bytes keep the gather (descriptor b0 = -1).
struct RpDesc {
    int32_t b0;          // first source byte of the window, -1: gather
    uint32_t valid;      // bit i: sample s0 + i < n_sel
    uint32_t rel[4];     // byte i: 2-bit position of sample i inside the window (sid - 4 b0)
    uint32_t pad[2];
};
static_assert(sizeof(RpDesc) == 32, "RpDesc is two 16-byte loads");

__global__ __launch_bounds__(256) void repack_desc_kernel(const int32_t *__restrict__ sample_idx, int n_sel, int nt, RpDesc *__restrict__ desc) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= nt * 8) return;
    const int s0 = (id >> 3) * JXG_TILE + (id & 7) * 16;
    int lo = 0x7fffffff, hi = -1;
    uint32_t valid = 0;
    int sid[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        sid[i] = 0;
        if (s0 + i < n_sel) {
            sid[i] = sample_idx[s0 + i];
            valid |= 1u << i;
            lo = min(lo, sid[i] >> 2);
            hi = max(hi, sid[i] >> 2);
        }
    }
    RpDesc d;
    d.b0 = (valid && hi - lo <= 7) ? lo : -1;
    d.valid = valid;
    d.rel[0] = d.rel[1] = d.rel[2] = d.rel[3] = 0;
    d.pad[0] = d.pad[1] = 0;
    if (d.b0 >= 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if ((valid >> i) & 1u) d.rel[i >> 2] |= (uint32_t)(sid[i] - 4 * lo) << (8 * (i & 3));
    }
    desc[id] = d;
}

__global__ __launch_bounds__(256) void repack_p32_window_kernel(const uint8_t *__restrict__ src, int64_t bps,
                                                                const int32_t *__restrict__ sample_idx, int n_sel,
                                                                const int64_t *__restrict__ row_idx, int64_t m_out,
                                                                const RpDesc *__restrict__ desc, uint32_t *__restrict__ dst) {
    const int64_t per_tile = m_out * 8;
    const int64_t rem = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rem >= per_tile) return;
    const int tile = (int)blockIdx.y;
    const int64_t gid = (int64_t)tile * per_tile + rem;
    const int64_t j = rem >> 3;
    const int d = (int)(rem & 7);
    const int64_t srow = row_idx ? row_idx[j] : j;
    const uint8_t *row = src + srow * bps;
    const uint4 *dp = reinterpret_cast<const uint4 *>(desc + (tile * 8 + d));
    const uint4 d0 = dp[0], d1 = dp[1];       // d0 = (b0, valid, rel0, rel1), d1 = (rel2, rel3, -, -)
    const int b0 = (int)d0.x;
    const uint32_t valid = d0.y;
    uint32_t w = 0;
    if (b0 >= 0) {
        const int a0 = b0 & ~3;
        uint32_t ww[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int off = a0 + 4 * k;
            uint32_t v = 0;
            if (off + 4 <= bps) {
                __builtin_memcpy(&v, row + off, 4);          // rows need not be 4-byte aligned (bps = ceil(n / 4)): one unaligned dword load
            } else {
                for (int b = 0; b < 4; ++b)
                    if (off + b < bps) v |= (uint32_t)row[off + b] << (8 * b);
            }
            ww[k] = v;
        }
        const int sh = 8 * (b0 & 3);
        unsigned long long val = (((unsigned long long)ww[1] << 32) | ww[0]) >> sh;
        if (sh) val |= (unsigned long long)ww[2] << (64 - sh);
        const uint32_t rel[4] = {d0.z, d0.w, d1.x, d1.y};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t r = (rel[i >> 2] >> (8 * (i & 3))) & 0xffu;
            const uint32_t code = ((valid >> i) & 1u) ? (uint32_t)(val >> (2 * r)) & 3u : 1u;
            w |= code << (2 * i);
        }
    } else {
        const int s0 = tile * JXG_TILE + d * 16;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int s = s0 + i;
            uint32_t code = 1u;  // missing
            if (s < n_sel) {
                const int sid = sample_idx[s];
                code = (row[sid >> 2] >> (2 * (sid & 3))) & 3u;
            }
            w |= code << (2 * i);
        }
    }
    dst[gid] = w;
}

// counts[j] = (missing, het, hom_alt) over the real samples; padding samples are stored as 01 and removed
// from `missing` afterwards by the host (n_pad - n_sel).  One thread per (SNP, tile) record of 32 bytes.
__global__ __launch_bounds__(256) void row_counts_p32_kernel(const uint4 *__restrict__ p32, int64_t m, int nt,
                                                             int32_t *__restrict__ counts) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // grid: x over SNPs, y = tile
    if (j >= m) return;
    const int tile = (int)blockIdx.y;
    const uint4 *rec = p32 + ((int64_t)tile * m + j) * 2;
    int mis = 0, het = 0, hom = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint4 v = rec[h];
        const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = ws[k] & 0x55555555u;
            const uint32_t hi = (ws[k] >> 1) & 0x55555555u;
            mis += __popc(lo & ~hi);
            het += __popc(hi & ~lo);
            hom += __popc(hi & lo);
        }
    }
    if (mis) atomicAdd(&counts[j * 3 + 0], mis);
    if (het) atomicAdd(&counts[j * 3 + 1], het);
    if (hom) atomicAdd(&counts[j * 3 + 2], hom);
}

// counts over a SAMPLE SUBSET straight from the PLINK payload (no P32 image): one wave per SNP row, lanes stride over the row's
// bytes four at a time where alignment allows; `mask` (bps bytes) holds 11 at every selected sample's two bits.  The selection
// must be duplicate-free (a mask cannot count a sample twice); order does not matter for counts.
__global__ __launch_bounds__(256) void row_counts_raw_masked_kernel(const uint8_t *__restrict__ src, int64_t bps, int64_t m,
                                                                    const uint8_t *__restrict__ mask,
                                                                    int32_t *__restrict__ counts) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= m) return;
    const int lane = threadIdx.x & 63;
    const uint8_t *row = src + j * bps;
    int mis = 0, het = 0, hom = 0;
    auto add = [&](uint32_t w, uint32_t mk) {
        const uint32_t sel = mk & 0x55555555u;                 // one bit per selected sample
        const uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
        mis += __popc(lo & ~hi & sel);
        het += __popc(hi & ~lo & sel);
        hom += __popc(hi & lo & sel);
    };
    // head bytes up to the first 4-byte boundary of this row, then dwords, then the tail
    const int64_t mis_al = (4 - ((uintptr_t)row & 3)) & 3;
    const int64_t head = mis_al < bps ? mis_al : bps;
    for (int64_t b = lane; b < head; b += 64) add(row[b], mask[b]);
    const int64_t nd = (bps - head) >> 2;
    const uint32_t *rw = reinterpret_cast<const uint32_t *>(row + head);
    for (int64_t d = lane; d < nd; d += 64) {
        const uint8_t *mp = mask + head + 4 * d;
        const uint32_t mk = (uint32_t)mp[0] | ((uint32_t)mp[1] << 8) | ((uint32_t)mp[2] << 16) | ((uint32_t)mp[3] << 24);
        add(rw[d], mk);
    }
    for (int64_t b = head + 4 * nd + lane; b < bps; b += 64) add(row[b], mask[b]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mis += __shfl_xor(mis, off, 64);
        het += __shfl_xor(het, off, 64);
        hom += __shfl_xor(hom, off, 64);
    }
    if (lane == 0) {
        counts[j * 3 + 0] = mis;
        counts[j * 3 + 1] = het;
        counts[j * 3 + 2] = hom;
    }
}

__global__ void fix_pad_missing_kernel(int32_t *counts, int64_t m, int pad) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) counts[j * 3] -= pad;
}

__global__ void cast_f64_f32_kernel(const double *__restrict__ src, float *__restrict__ dst, int64_t count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) dst[i] = (float)src[i];
}

}  // namespace jx

using namespace jx;

extern "C" int jxg_num_tiles(int n) { return num_tiles(n); }

extern "C" int jxg_repack_p32(const uint8_t *d_packed, int64_t bps, int n_src, int64_t m_src,
                              const int32_t *d_sample_idx, int n_sel, const int64_t *d_row_idx, int64_t m_out,
                              uint8_t *d_p32, void *stream) {
    (void)m_src;
    if (n_sel <= 0 || m_out <= 0) return fail("jxg_repack_p32: empty input");
    const int nt = num_tiles(n_sel);
    const int64_t blocks = (m_out * 8 + 255) / 256;
    if (blocks * 256 > 0xffffffffLL || nt > 65535) return fail("jxg_repack_p32: grid too large");
    hipStream_t st = (hipStream_t)stream;
    // sample subsets: the window form (descriptors per output dword position, built here); JXGPU_REPACK_WINDOW=0: the gather
    const char *we = getenv("JXGPU_REPACK_WINDOW");      // read per call: the two forms are compared inside one process by the tests
    const bool window = !(we && atoi(we) == 0);
    if (d_sample_idx && window && m_out >= 64) {
        AsyncBlock ab;
        if (ab.alloc(sizeof(RpDesc) * (size_t)nt * 8, st)) return 1;
        RpDesc *desc = (RpDesc *)ab.p;
        hipLaunchKernelGGL(repack_desc_kernel, dim3((unsigned)((nt * 8 + 255) / 256)), dim3(256), 0, st, d_sample_idx, n_sel, nt, desc);
        JX_LAUNCH_CHECK();
        hipLaunchKernelGGL(repack_p32_window_kernel, dim3((unsigned)blocks, (unsigned)nt), dim3(256), 0, st, d_packed, bps, d_sample_idx,
                           n_sel, d_row_idx, m_out, desc, (uint32_t *)d_p32);
        JX_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(repack_p32_kernel, dim3((unsigned)blocks, (unsigned)nt), dim3(256), 0, st, d_packed, bps,
                       n_src, d_sample_idx, n_sel, d_row_idx, m_out, (uint32_t *)d_p32, nt);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_row_counts_p32(const uint8_t *d_p32, int64_t m, int n_sel, int32_t *d_counts, void *stream) {
    const int nt = num_tiles(n_sel);
    hipStream_t st = (hipStream_t)stream;
    JX_HIP(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * 3 * (size_t)m, st));
    const int64_t blocks = (m + 255) / 256;
    if (blocks * 256 > 0xffffffffLL || nt > 65535) return fail("jxg_row_counts_p32: grid too large");
    hipLaunchKernelGGL(row_counts_p32_kernel, dim3((unsigned)blocks, (unsigned)nt), dim3(256), 0, st, (const uint4 *)d_p32, m, nt,
                       d_counts);
    JX_LAUNCH_CHECK();
    const int pad = nt * JXG_TILE - n_sel;
    if (pad > 0) {
        hipLaunchKernelGGL(fix_pad_missing_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, d_counts, m,
                           pad);
        JX_LAUNCH_CHECK();
    }
    return 0;
}

// Decoded rows of a P32 image: out[r][i] = lut[r][code(rows[r], i)], i < n (`bed_packed_decode_rows_f32`,
// src/stats/packed.rs:577-672).  One workgroup = one (sample tile, row): the 32-byte record is read once (uniform), the 128
// values of the tile are stored side by side.
__global__ __launch_bounds__(128) void p32_decode_rows_kernel(const uint32_t *__restrict__ p32, int64_t m_total,
                                                             const int32_t *__restrict__ rows, int nrows,
                                                             const float *__restrict__ lut, int n, float *__restrict__ out,
                                                             int64_t ld) {
    const int tile = blockIdx.x, t = threadIdx.x;
    const int i = tile * JXG_TILE + t;
    for (int r = blockIdx.y; r < nrows; r += gridDim.y) {
        const int64_t rec = rows ? (int64_t)rows[r] : (int64_t)r;
        const uint32_t w = p32[((int64_t)tile * m_total + rec) * 8 + (t >> 4)];
        const uint32_t code = (w >> (2 * (t & 15))) & 3u;
        if (i < n) out[(int64_t)r * ld + i] = lut[(int64_t)r * 4 + code];
    }
}

extern "C" int jxg_decode_rows_p32(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                                   const float *d_lut, float *d_out, int64_t ld, void *stream) {
    if (nrows <= 0 || n <= 0) return 0;
    if (ld < n) return fail("jxg_decode_rows_p32: ld must be >= n");
    const dim3 grid((unsigned)num_tiles(n), (unsigned)std::min(nrows, 65535));
    hipLaunchKernelGGL(p32_decode_rows_kernel, grid, dim3(128), 0, (hipStream_t)stream, (const uint32_t *)d_p32, m_total,
                       d_rows, nrows, d_lut, n, d_out, ld);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_cast_f64_to_f32(const double *d_src, float *d_dst, int64_t count, void *stream) {
    int64_t blocks = (count + 255) / 256;
    if (blocks > 65535 * 16) blocks = 65535 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_f64_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_src, d_dst,
                       count);
    JX_LAUNCH_CHECK();
    return 0;
}

// Counts over a duplicate-free sample subset straight from a DEVICE-resident PLINK payload (m x bps bytes): d_mask (bps bytes) holds
// 11 at the selected samples.  No P32 image: a 50 GB payload is read once (the staged form allocates, fills and frees a 40 GB image).
extern "C" int jxg_row_counts_raw_masked(const uint8_t *d_packed, int64_t bps, int64_t m, const uint8_t *d_mask, int32_t *d_counts,
                                         void *stream) {
    if (m <= 0) return 0;
    const int64_t blocks = (m + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail("jxg_row_counts_raw_masked: grid too large");
    hipLaunchKernelGGL(row_counts_raw_masked_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_packed, bps, m,
                       d_mask, d_counts);
    JX_LAUNCH_CHECK();
    return 0;
}

