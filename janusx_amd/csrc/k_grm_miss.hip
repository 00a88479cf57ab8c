// GRM rows that are affine in the allele count but hold missing calls (src/stats/grm.rs:1638-1772 with the decode of
// src/decode/decode.rs:813-839: a missing call takes the centred value of the mean, i.e. its own LUT entry).
//
// Such a row is  z = (b + s c) + d e :  c = count of the payload's second allele with 0 at a missing call, s = +-1, e = the
// 0/1 indicator of the missing calls, d = (LUT value of a missing call) - b.  The first part is what the int8 Gram kernel
// (k_grm_i8.hip) and its affine terms compute for a row WITHOUT missing calls, so the row goes there as it is, and
//   Z Z' = [clean form] + (W + W'),   W[i, i'] = sum_{j : e_j[i'] = 1} d_j ( b_j + s_j c_ij + d_j e_ij / 2 )
// is added here: a sparse-times-dense product -- per sample i' the list of the SNPs it misses (1 % of them at a 1 % missing
// rate), per (i, i') one table value wl_j[code(j, i)] per list entry, f64 sums.  nnz(e) n table lookups instead of the two
// extra fp16 products over all SNPs the split kernel (k_grm.hip) spends on these rows; exact up to the f64 sums.
// Lists are built without atomics (thread = sample, counts and fills in SNP order), W' is written transposed (coalesced) into
// an n_pad^2 f64 buffer and merged into the lower triangle of the accumulator by a last kernel: the result does not depend on
// scheduling.
#include <algorithm>

#include "jx_common.h"

namespace jx {

constexpr int GM_CHUNKS = 32;      // SNP chunks of the list build (parallelism of the count / fill passes)

// cnt[i' * GM_CHUNKS + c] = missing calls of sample i' among the flagged SNPs of chunk c (positions of the reordered list)
template <bool FILL>
__global__ __launch_bounds__(128) void gm_lists_kernel(const uint32_t *__restrict__ p32, int64_t m_total,
                                                       const int32_t *__restrict__ rows2, const uint8_t *__restrict__ miss2,
                                                       int64_t nex, int64_t chunk, int n_sel,
                                                       int64_t *__restrict__ cnt_or_off, int32_t *__restrict__ ent) {
    const int t = blockIdx.x, c = blockIdx.y, s = threadIdx.x;
    const int i = t * JXG_TILE + s;
    const int64_t k0 = (int64_t)c * chunk, k1 = (k0 + chunk < nex) ? k0 + chunk : nex;
    const int dsel = s >> 4, sh = 2 * (s & 15);
    const uint32_t *base = p32 + (int64_t)t * m_total * 8 + dsel;
    int64_t at = FILL ? cnt_or_off[(int64_t)i * GM_CHUNKS + c] : 0;
    const bool real = i < n_sel;
    int64_t k = k0;
    for (; k + 8 <= k1; k += 8) {                      // eight records in flight (the loop is a chain of dependent loads)
        uint32_t w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = miss2[k + u] ? base[(int64_t)rows2[k + u] * 8] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (((w[u] >> sh) & 3u) == 1u && real) {
                if (FILL) ent[at] = (int32_t)(k + u);
                ++at;
            }
        }
    }
    for (; k < k1; ++k) {
        if (!miss2[k]) continue;                                   // uniform
        const uint32_t w = base[(int64_t)rows2[k] * 8];
        if (((w >> sh) & 3u) == 1u && real) {
            if (FILL) ent[at] = (int32_t)k;
            ++at;
        }
    }
    if (!FILL) cnt_or_off[(int64_t)i * GM_CHUNKS + c] = at;
}

// exclusive scan of `len` int64 counters in place (one workgroup); total -> *total_out
__global__ __launch_bounds__(1024) void gm_scan_kernel(int64_t *__restrict__ v, int64_t len, int64_t *__restrict__ total_out) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (len + 1023) / 1024;
    const int64_t b = tid * per, e = (b + per < len) ? b + per : len;
    int64_t sum = 0;
    for (int64_t k = b; k < e; ++k) sum += v[k];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t x = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += x;
        __syncthreads();
    }
    int64_t run = part[tid] - sum;
    for (int64_t k = b; k < e; ++k) {
        const int64_t x = v[k];
        v[k] = run;
        run += x;
    }
    if (tid == 1023) *total_out = part[1023];
}

// wt[i' * ld + i] = W[i, i'] for 4096 samples i per workgroup: a thread owns ONE payload dword position (16 samples, 16 f64
// sums in registers), so a list entry costs it one 4-byte load -- a wave reads eight whole 32-byte records -- and per sample a
// bit-field extract, an LDS table read and an add.  The entries of a batch (record offset and the four table values of the
// SNP) are staged in LDS by the workgroup.
constexpr int GM_BATCH = 64;
__global__ __launch_bounds__(256) void gm_spmm_kernel(const uint32_t *__restrict__ p32, int64_t m_total,
                                                      const int32_t *__restrict__ rows2, const double *__restrict__ wl,
                                                      const int64_t *__restrict__ off, const int32_t *__restrict__ ent,
                                                      int64_t nnz, int n_sel, int nt, int64_t ld, double *__restrict__ wt) {
    __shared__ double tab[GM_BATCH * 4];
    __shared__ int64_t recs[GM_BATCH];
    const int ip = blockIdx.x;                         // sample whose missing calls are listed
    const int tid = threadIdx.x;
    const int t = blockIdx.y * 32 + (tid >> 3), d = tid & 7;      // sample tile and dword of this thread
    const bool live = t < nt;
    const int64_t e0 = off[(int64_t)ip * GM_CHUNKS];
    const int64_t e1 = (ip + 1 < (int)gridDim.x) ? off[(int64_t)(ip + 1) * GM_CHUNKS] : nnz;
    const uint32_t *base = p32 + (int64_t)(live ? t : 0) * m_total * 8 + d;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int64_t eb = e0; eb < e1; eb += GM_BATCH) {
        const int cnt = (int)((e1 - eb < GM_BATCH) ? (e1 - eb) : GM_BATCH);
        __syncthreads();
        if (tid < cnt) {
            const int32_t k = ent[eb + tid];
            recs[tid] = (int64_t)rows2[k] * 8;
            const double4 v = *reinterpret_cast<const double4 *>(wl + (int64_t)k * 4);
            tab[tid * 4 + 0] = v.x;
            tab[tid * 4 + 1] = v.y;
            tab[tid * 4 + 2] = v.z;
            tab[tid * 4 + 3] = v.w;
        }
        __syncthreads();
        int u = 0;
        for (; u + 4 <= cnt; u += 4) {
            uint32_t w[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) w[x] = base[recs[u + x]];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const double *tb = tab + (u + x) * 4;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] += tb[(w[x] >> (2 * q)) & 3u];
            }
        }
        for (; u < cnt; ++u) {
            const uint32_t w = base[recs[u]];
            const double *tb = tab + u * 4;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] += tb[(w >> (2 * q)) & 3u];
        }
    }
    if (live) {
        double *o = wt + (int64_t)ip * ld + (int64_t)t * JXG_TILE + d * 16;
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] = (t * JXG_TILE + d * 16 + q < n_sel) ? acc[q] : 0.0;
    }
}

// acc[i][i'] += W[i, i'] + W[i', i] on the lower triangle (32 x 32 blocks, the transposed operand through LDS)
__global__ __launch_bounds__(256) void gm_merge_kernel(const double *__restrict__ wt, int64_t ld, int n_sel,
                                                       double *__restrict__ acc) {
    __shared__ double tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;       // column block, row block
    if (bx > by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    // tile[r][c] = wt[(by*32 + r) * ld + bx*32 + c] = W[bx*32 + c, by*32 + r]  (= W'[i, i'] transposed operand)
    for (int r = ty; r < 32; r += 8) tile[r][tx] = wt[(int64_t)(by * 32 + r) * ld + bx * 32 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = by * 32 + r, ip = bx * 32 + tx;
        if (i < n_sel && ip <= i) {
            // W[i, i'] = wt[i' * ld + i]: strided over i' (tx) -- served from L2 in 32-byte sectors; W[i', i] = tile[r][tx]
            acc[(int64_t)i * ld + ip] += wt[(int64_t)ip * ld + i] + tile[r][tx];
        }
    }
}

// Adds (W + W') of the flagged rows (miss2[k] != 0, k < nex of the reordered list) to the lower triangle of d_acc.
// wl: (nex, 4) f64 table per SNP and 2-bit code.  Returns 0 on success, 1 on a failed launch / allocation, 2 when the lists
// would not fit (nothing has been added then).  The caller (k_grm.hip) checks the memory BEFORE it commits rows to this path
// -- n_pad^2 doubles + 4 bytes per missing call, known right after the classification -- and sends them to the general kernel
// instead; a 2 here (another process took the memory in between) is a hard failure, the clean form has been accumulated.
int grm_missing_correction(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n_sel, int nt, const int32_t *rows2,
                           const uint8_t *miss2, const double *wl, int64_t nex, double *d_acc, int64_t ld) {
    if (nex <= 0) return 0;
    const int64_t ncnt = ld * GM_CHUNKS;
    DevBuf offb, totb, entb, wtb;
    if (offb.alloc(sizeof(int64_t) * (size_t)(ncnt + 1)) || totb.alloc(sizeof(int64_t))) return 1;
    const int64_t chunk = (nex + GM_CHUNKS - 1) / GM_CHUNKS;
    const uint32_t *p32w = reinterpret_cast<const uint32_t *>(d_p32);
    hipLaunchKernelGGL(gm_lists_kernel<false>, dim3(nt, GM_CHUNKS), dim3(128), 0, st, p32w, m_total, rows2, miss2, nex, chunk,
                       n_sel, offb.as<int64_t>(), (int32_t *)nullptr);
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(gm_scan_kernel, dim3(1), dim3(1024), 0, st, offb.as<int64_t>(), ncnt, totb.as<int64_t>());
    JX_LAUNCH_CHECK();
    int64_t nnz = 0;
    JX_HIP(hipMemcpyAsync(&nnz, totb.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    if (nnz == 0) return 0;
    size_t fr = 0, tot = 0;
    const size_t need = sizeof(double) * (size_t)ld * (size_t)ld + sizeof(int32_t) * (size_t)nnz;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < need + ((size_t)2 << 30)) return 2;
    if (entb.alloc(sizeof(int32_t) * (size_t)nnz) || wtb.alloc(sizeof(double) * (size_t)ld * (size_t)ld)) return 1;
    hipLaunchKernelGGL(gm_lists_kernel<true>, dim3(nt, GM_CHUNKS), dim3(128), 0, st, p32w, m_total, rows2, miss2, nex, chunk,
                       n_sel, offb.as<int64_t>(), entb.as<int32_t>());
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(gm_spmm_kernel, dim3((unsigned)ld, (unsigned)((nt + 31) / 32)), dim3(256), 0, st, p32w, m_total,
                       rows2, wl, offb.as<int64_t>(), entb.as<int32_t>(), nnz, n_sel, nt, ld, wtb.as<double>());
    JX_LAUNCH_CHECK();
    const unsigned nb = (unsigned)(ld / 32);
    hipLaunchKernelGGL(gm_merge_kernel, dim3(nb, nb), dim3(256), 0, st, wtb.as<double>(), ld, n_sel, d_acc);
    JX_LAUNCH_CHECK();
    JX_HIP(hipStreamSynchronize(st));      // the buffers are released on return
    return 0;
}

}  // namespace jx
