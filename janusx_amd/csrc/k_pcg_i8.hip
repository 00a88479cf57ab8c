// The two halves of the matrix-free PCG operator Z (Z'p) of rrBLUP (`rrblup_pcg_bed`, src/stats/rrblup.rs:1220-1372; operator
// src/math/pcg.rs:578-640) on the int8 matrix pipes.
//
// The table forms of k_gblup.hip (packed_tdot_f32_kernel / packed_dot_t32_kernel) spend three LDS lookups and three f32 additions
// per four genotypes plus a table build per 128-sample tile: 40.6 ms per application at BASELINE configs[4] (80 GB of payload =
// 0.25 of the HBM peak), instruction-bound.  Both halves are sums of a VECTOR over the three bit planes of the 2-bit codes
//     S_lo = sum v [b0],   S_hi = sum v [b1],   S_both = sum v [b0 & b1]
// (per SNP over samples for Z u, per sample over SNPs -- with three per-SNP weight vectors -- for Z'p), i.e. products of a 0/1
// matrix with a few vectors.  Here the vector is written ONCE per application as FOUR signed base-254 digit planes against its
// largest magnitude (v = vmax (q1/127 + q2/(127 254) + q3/(127 254^2) + q4/(127 254^3)), |q| <= 127: 2^-31 of vmax, finer than the
// f32 rounding the reference's vectors carry) and the planes are the ROWS of the A operand of v_mfma_i32_16x16x64_i8; the B operand
// is the bit plane of 16 SNPs (resp. samples) x 64 samples (SNPs), decoded in registers from the lane's own payload dword (21 VALU
// instructions per 16 genotypes for the three planes).  The i32 sums are exact; a wave keeps ONE unit (16 SNPs / 16 samples) for the
// whole K range, so there is no table, no per-tile barrier and no atomic -- what is left is the payload stream.
//   Z u   (markers <- samples): P32 image p32[tile][snp][32 B]; A rows 0..3 = planes of u (f32-rounded like the table form).
//         out[r] = l0 c0 + l1 c1 + l2 c2 + l3 c3 from the class sums c3 = S_both, c2 = S_hi - S_both, c1 = S_lo - S_both,
//         c0 = (sum of the quantised u) - c1 - c2 - c3.
//   Z'p   (samples <- markers): T32 image t32[snp tile][sample][32 B]; A rows 0..3 / 4..7 / 8..11 = planes of the per-SNP weights
//         d_lo = w1 - w0, d_hi = w2 - w0, d_both = w3 - w2 - w1 + w0 (w_c = f32(lut[r][c] f32(p_r)), as pcg_plane_weights_kernel);
//         out[i] = sum w0 + S_lo(d_lo) + S_hi(d_hi) + S_both(d_both).  The SNP tiles are cut into slices whose partial sums a small
//         kernel adds in a fixed order.
// k slots: MFMA h (0, 1) of a 128-element record takes for lane quarter kq the payload dword 2 kq + h (a lane's two dwords are one
// 8-byte load); byte 4 q + b of a decoded dword is element 4 b + q, and the digit images are stored in that order.
#include <stdlib.h>

#include "jx_common.h"

namespace jx {

typedef int pi_i32x4 __attribute__((ext_vector_type(4)));

constexpr int PI_STAGE = 8;          // records (128 elements of K) per LDS stage of the digit image
constexpr int PI_WAVES = 8;          // waves per workgroup, 16 units each

struct PiScalars {                   // device scalars of one quantised vector set
    unsigned long long maxbits;      // bit pattern of the largest magnitude (non-negative doubles order like integers)
    long long tot[4];                // sum of every digit plane (Z u: the sum of the quantised vector, exact)
    double w0sum;                    // Z'p: sum of w0
};

__device__ __forceinline__ void pi_digits(double x, double inv, int q[4]) {
    double a = x * inv * 127.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double d = rint(a);
        a = (a - d) * 254.0;
        d = fmin(fmax(d, -127.0), 127.0);
        q[k] = (int)d;
    }
}

// three bit planes of 16 two-bit codes as MFMA operand bytes (byte 4 q + b = element 4 b + q; 0 / 1)
__device__ __forceinline__ void pi_planes(uint32_t w, pi_i32x4 &lo, pi_i32x4 &hi, pi_i32x4 &both) {
    const uint32_t l = w & 0x55555555u, h = (w >> 1) & 0x55555555u;
    lo.x = (int)(l & 0x01010101u);
    lo.y = (int)((l >> 2) & 0x01010101u);
    lo.z = (int)((l >> 4) & 0x01010101u);
    lo.w = (int)((l >> 6) & 0x01010101u);
    hi.x = (int)(h & 0x01010101u);
    hi.y = (int)((h >> 2) & 0x01010101u);
    hi.z = (int)((h >> 4) & 0x01010101u);
    hi.w = (int)((h >> 6) & 0x01010101u);
    both = lo & hi;
}

__device__ __forceinline__ double pi_combine(const pi_i32x4 a) {
    constexpr double W1 = 1.0 / 127.0, R = 1.0 / 254.0;
    return (((double)a.w * R + (double)a.z) * R + (double)a.y) * R * W1 + (double)a.x * W1;
}

// ---- quantisation of the vector of Z u ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pi_absmax_f32_kernel(const double *__restrict__ u, int64_t n, PiScalars *__restrict__ sc) {
    double m = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = fabs((double)(float)u[i]);
        m = (v > m || v != v) ? v : m;
    }
    unsigned long long b = (unsigned long long)__double_as_longlong(m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(b, off, 64);
        b = o > b ? o : b;
    }
    if ((threadIdx.x & 63) == 0 && b) atomicMax(&sc->maxbits, b);
}

// image [tile][h][kq][plane][16 B]: thread = (tile, slot h * 4 + kq): the slot's 16 elements 16 (2 kq + h) .. + 15 of the tile
__global__ __launch_bounds__(256) void pi_quant_u_kernel(const double *__restrict__ u, int n, int ntiles, PiScalars *__restrict__ sc,
                                                         int8_t *__restrict__ img) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    long long tot[4] = {0, 0, 0, 0};
    if (idx < (int64_t)ntiles * 8) {
        const int tile = (int)(idx >> 3), slot = (int)(idx & 7), h = slot >> 2, kq = slot & 3;
        const double vmax = __longlong_as_double((long long)sc->maxbits);
        const double inv = (vmax > 0.0 && vmax < 1.0e300) ? 1.0 / vmax : 0.0;
        int8_t o[4][16];
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            const int e = ((pos & 3) << 2) | (pos >> 2);                 // byte 4 q + b holds element 4 b + q
            const int i = tile * 128 + 16 * (2 * kq + h) + e;
            int q[4] = {0, 0, 0, 0};
            if (i < n) pi_digits((double)(float)u[i], inv, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k][pos] = (int8_t)q[k];
                tot[k] += q[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<uint4 *>(img + (((int64_t)tile * 8 + slot) * 4 + k) * 16) = *reinterpret_cast<const uint4 *>(o[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        long long t = tot[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if ((threadIdx.x & 63) == 0 && t) atomicAdd((unsigned long long *)&sc->tot[k], (unsigned long long)t);
    }
}

// ---- Z u: out[r] = sum_i lut[r][code(r, i)] u_i --------------------------------------------------------------------------------
__global__ __launch_bounds__(PI_WAVES * 64) void pi_tdot_kernel(const uint8_t *__restrict__ p32, int64_t m_total,
                                                                const int32_t *__restrict__ rows, int nrows, int ntiles,
                                                                const int8_t *__restrict__ img, const PiScalars *__restrict__ sc,
                                                                const float *__restrict__ lut, double *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) int8_t a_sh[2][PI_STAGE * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 15, kq = lane >> 4;
    const int r = (blockIdx.x * PI_WAVES + wave) * 16 + s;
    const int64_t rec = (r < nrows) ? (rows ? (int64_t)rows[r] : (int64_t)r) : 0;
    const uint8_t *gsrc = p32 + rec * 32 + 8 * kq;
    const int64_t tstride = m_total * 32;
    pi_i32x4 acc_lo = {0, 0, 0, 0}, acc_hi = {0, 0, 0, 0}, acc_b = {0, 0, 0, 0};
    const int nstage = (ntiles + PI_STAGE - 1) / PI_STAGE;
    auto stage_load = [&](int sg, int buf) {
        // 512 threads x 8 B = one stage of the digit image (tiles past the end: zeros)
        const int64_t off = (int64_t)sg * PI_STAGE * 512 + tid * 8;
        uint2 v = make_uint2(0u, 0u);
        if (off < (int64_t)ntiles * 512) v = *reinterpret_cast<const uint2 *>(img + off);
        *reinterpret_cast<uint2 *>(&a_sh[buf][tid * 8]) = v;
    };
    stage_load(0, 0);
    for (int sg = 0; sg < nstage; ++sg) {
        const int buf = sg & 1;
        __syncthreads();                                   // stage sg is in LDS, nobody reads the other buffer any more
        if (sg + 1 < nstage) stage_load(sg + 1, buf ^ 1);
        const int t0 = sg * PI_STAGE;
        uint2 g[PI_STAGE];
#pragma unroll
        for (int j = 0; j < PI_STAGE; ++j) {
            const int t = (t0 + j < ntiles) ? t0 + j : ntiles - 1;           // past the end: a record whose digits are zero
            g[j] = *reinterpret_cast<const uint2 *>(gsrc + (int64_t)t * tstride);
        }
#pragma unroll
        for (int j = 0; j < PI_STAGE; ++j) {
            pi_i32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            if (s < 4) {
                a0 = *reinterpret_cast<const pi_i32x4 *>(&a_sh[buf][((j * 8 + kq) * 4 + s) * 16]);
                a1 = *reinterpret_cast<const pi_i32x4 *>(&a_sh[buf][((j * 8 + 4 + kq) * 4 + s) * 16]);
            }
            pi_i32x4 lo, hi, bo;
            pi_planes(g[j].x, lo, hi, bo);
            acc_lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, lo, acc_lo, 0, 0, 0);
            acc_hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, hi, acc_hi, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bo, acc_b, 0, 0, 0);
            pi_planes(g[j].y, lo, hi, bo);
            acc_lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, lo, acc_lo, 0, 0, 0);
            acc_hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, hi, acc_hi, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bo, acc_b, 0, 0, 0);
        }
    }
    // D rows 4 g + r of column s: the lanes of quarter 0 hold the four plane sums of their SNP
    if (kq == 0 && r < nrows) {
        const double vmax = __longlong_as_double((long long)sc->maxbits);
        const double s_lo = vmax * pi_combine(acc_lo), s_hi = vmax * pi_combine(acc_hi), s_b = vmax * pi_combine(acc_b);
        const double utot = vmax * ((((double)sc->tot[3] / 254.0 + (double)sc->tot[2]) / 254.0 + (double)sc->tot[1]) / 254.0 / 127.0 +
                                    (double)sc->tot[0] / 127.0);
        const double c3 = s_b, c2 = s_hi - s_b, c1 = s_lo - s_b, c0 = utot - c1 - c2 - c3;
        const float *l = lut + (int64_t)r * 4;
        out[r] = (double)l[0] * c0 + (double)l[1] * c1 + (double)l[2] * c2 + (double)l[3] * c3;
    }
}

// ---- weights of Z'p: wq (d_lo, d_hi, d_both, w0) as pcg_plane_weights_kernel, their largest magnitude, sum w0 ------------------
__global__ __launch_bounds__(256) void pi_weights_kernel(const float *__restrict__ lut, const double *__restrict__ p, int nrows,
                                                         float4 *__restrict__ wq, PiScalars *__restrict__ sc) {
    __shared__ double sh[4];
    const int r = blockIdx.x * 256 + threadIdx.x;
    double w0d = 0.0, m = 0.0;
    if (r < nrows) {
        const float pr = (float)p[r];
        const float w0 = lut[(int64_t)r * 4 + 0] * pr, w1 = lut[(int64_t)r * 4 + 1] * pr;
        const float w2 = lut[(int64_t)r * 4 + 2] * pr, w3 = lut[(int64_t)r * 4 + 3] * pr;
        const double d3 = ((double)w3 - (double)w2) - (double)w1 + (double)w0;
        const float4 q = make_float4(w1 - w0, w2 - w0, (float)d3, w0);
        wq[r] = q;
        w0d = (double)w0;
        const double a = fabs((double)q.x), b = fabs((double)q.y), c = fabs((double)q.z);
        m = fmax(a, fmax(b, c));
        if (a != a || b != b || c != c) m = a + b + c;       // NaN goes through to the scale
    }
    unsigned long long bits = (unsigned long long)__double_as_longlong(m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(bits, off, 64);
        bits = o > bits ? o : bits;
        w0d += __shfl_xor(w0d, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = w0d;
        if (bits) atomicMax(&sc->maxbits, bits);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = sh[0] + sh[1] + sh[2] + sh[3];
        if (t != 0.0) unsafeAtomicAdd(&sc->w0sum, t);
    }
}

// image [snp tile][h][kq][row = 4 vector + plane][16 B] (12 rows): thread = (tile, slot, vector)
__global__ __launch_bounds__(256) void pi_quant_w_kernel(const float4 *__restrict__ wq, int nrows, int nst, const PiScalars *__restrict__ sc,
                                                         int8_t *__restrict__ img) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)nst * 24) return;
    const int vec = (int)(idx % 3), slot = (int)((idx / 3) & 7), tile = (int)(idx / 24), h = slot >> 2, kq = slot & 3;
    const double vmax = __longlong_as_double((long long)sc->maxbits);
    const double inv = (vmax > 0.0 && vmax < 1.0e300) ? 1.0 / vmax : 0.0;
    int8_t o[4][16];
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        const int e = ((pos & 3) << 2) | (pos >> 2);
        const int r = tile * 128 + 16 * (2 * kq + h) + e;
        int q[4] = {0, 0, 0, 0};
        if (r < nrows) {
            const float4 w = wq[r];
            pi_digits((double)(vec == 0 ? w.x : (vec == 1 ? w.y : w.z)), inv, q);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k][pos] = (int8_t)q[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<uint4 *>(img + (((int64_t)tile * 8 + slot) * 12 + vec * 4 + k) * 16) = *reinterpret_cast<const uint4 *>(o[k]);
}

// ---- Z'p: part[slice][i] = sum over the slice's SNP tiles of the three plane sums ---------------------------------------------
__global__ __launch_bounds__(PI_WAVES * 64) void pi_dot_kernel(const uint8_t *__restrict__ t32, int n, int nst, int tiles_per_slice,
                                                               const int8_t *__restrict__ img, const PiScalars *__restrict__ sc,
                                                               double *__restrict__ part) {
    __shared__ __attribute__((aligned(16))) int8_t a_sh[2][PI_STAGE * 1536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = lane & 15, kq = lane >> 4;
    const int i = (blockIdx.x * PI_WAVES + wave) * 16 + s;
    const int st0 = blockIdx.y * tiles_per_slice;
    const int st1 = (st0 + tiles_per_slice < nst) ? st0 + tiles_per_slice : nst;
    const uint8_t *gsrc = t32 + (int64_t)(i < n ? i : 0) * 32 + 8 * kq;
    const int64_t tstride = (int64_t)n * 32;
    pi_i32x4 acc_lo = {0, 0, 0, 0}, acc_hi = {0, 0, 0, 0}, acc_b = {0, 0, 0, 0};
    const int ntl = st1 - st0;
    const int nstage = (ntl + PI_STAGE - 1) / PI_STAGE;
    auto stage_load = [&](int sg, int buf) {
        // 512 threads x 24 B = one stage (8 tiles x 1536 B)
        const int64_t base = ((int64_t)st0 + (int64_t)sg * PI_STAGE) * 1536;
        const int64_t end = (int64_t)st1 * 1536;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int o = (q * 512 + tid) * 8;
            uint2 v = make_uint2(0u, 0u);
            if (base + o < end) v = *reinterpret_cast<const uint2 *>(img + base + o);
            *reinterpret_cast<uint2 *>(&a_sh[buf][o]) = v;
        }
    };
    if (nstage > 0) stage_load(0, 0);
    for (int sg = 0; sg < nstage; ++sg) {
        const int buf = sg & 1;
        __syncthreads();
        if (sg + 1 < nstage) stage_load(sg + 1, buf ^ 1);
        const int t0 = st0 + sg * PI_STAGE;
        uint2 g[PI_STAGE];
#pragma unroll
        for (int j = 0; j < PI_STAGE; ++j) {
            const int t = (t0 + j < st1) ? t0 + j : st1 - 1;
            g[j] = *reinterpret_cast<const uint2 *>(gsrc + (int64_t)t * tstride);
        }
#pragma unroll
        for (int j = 0; j < PI_STAGE; ++j) {
            pi_i32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            if (s < 12) {
                a0 = *reinterpret_cast<const pi_i32x4 *>(&a_sh[buf][((j * 8 + kq) * 12 + s) * 16]);
                a1 = *reinterpret_cast<const pi_i32x4 *>(&a_sh[buf][((j * 8 + 4 + kq) * 12 + s) * 16]);
            }
            pi_i32x4 lo, hi, bo;
            pi_planes(g[j].x, lo, hi, bo);
            acc_lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, lo, acc_lo, 0, 0, 0);
            acc_hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, hi, acc_hi, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bo, acc_b, 0, 0, 0);
            pi_planes(g[j].y, lo, hi, bo);
            acc_lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, lo, acc_lo, 0, 0, 0);
            acc_hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, hi, acc_hi, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bo, acc_b, 0, 0, 0);
        }
    }
    // D rows 4 kq + r of column s: quarter 0 holds the planes of d_lo (product with the lo plane), 1 d_hi, 2 d_both
    const double vmax = __longlong_as_double((long long)sc->maxbits);
    double v = 0.0;
    if (kq == 0) v = vmax * pi_combine(acc_lo);
    else if (kq == 1) v = vmax * pi_combine(acc_hi);
    else if (kq == 2) v = vmax * pi_combine(acc_b);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (kq == 0 && i < n) part[(int64_t)blockIdx.y * n + i] = v;
}

__global__ __launch_bounds__(256) void pi_dot_reduce_kernel(const double *__restrict__ part, int n, int slices, const PiScalars *__restrict__ sc,
                                                            double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = sc->w0sum;
    for (int q = 0; q < slices; ++q) v += part[(int64_t)q * n + i];
    out[i] = v;
}

// 1 (default): the operator halves on the int8 pipes from 4096 units on; JXGPU_PCG_I8=0: the table forms of k_gblup.hip
bool pcg_i8_enabled(int n, int nrows) {
    const char *e = getenv("JXGPU_PCG_I8");          // read per call: the forms are compared inside one process by the tests
    return !(e && atoi(e) == 0) && n >= 1024 && nrows >= 1024;
}

// d_out[r] = sum_i lut[r][code(r, i)] f32(u_i), r < nrows
int packed_tdot_i8(hipStream_t st, const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows, const float *d_lut,
                   const double *d_u, double *d_out) {
    const int ntiles = (n + 127) / 128;
    const size_t b_img = (size_t)ntiles * 512;
    AsyncBlock ab;
    if (ab.alloc(256 + b_img, st)) return 1;
    char *blk = (char *)ab.p;
    PiScalars *sc = (PiScalars *)blk;
    int8_t *img = (int8_t *)(blk + 256);
    JX_HIP(hipMemsetAsync(sc, 0, sizeof(PiScalars), st));
    int gb = (n + 255) / 256;
    if (gb > 1024) gb = 1024;
    hipLaunchKernelGGL(pi_absmax_f32_kernel, dim3(gb), dim3(256), 0, st, d_u, (int64_t)n, sc);
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(pi_quant_u_kernel, dim3((unsigned)(((int64_t)ntiles * 8 + 255) / 256)), dim3(256), 0, st, d_u, n, ntiles, sc, img);
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(pi_tdot_kernel, dim3((nrows + PI_WAVES * 16 - 1) / (PI_WAVES * 16)), dim3(PI_WAVES * 64), 0, st, d_p32, m_total,
                       d_rows, nrows, ntiles, img, sc, d_lut, d_out);
    JX_LAUNCH_CHECK();
    return 0;
}

// d_out[i] = sum_r f32(lut[r][code(r, i)] f32(beta_r)), i < n; d_work: nrows float4 (the caller's 16 nrows + 16 bytes)
int packed_dot_t32_i8(hipStream_t st, const uint8_t *d_t32, int n, int nrows, const float *d_lut, const double *d_beta, void *d_work,
                      double *d_out) {
    const int nst = (nrows + 127) / 128;
    const int gx = (n + PI_WAVES * 16 - 1) / (PI_WAVES * 16);
    int slices = (2048 + gx - 1) / gx;               // ~2048 workgroups
    if (slices > nst) slices = nst;
    if (slices < 1) slices = 1;
    const int tps = (nst + slices - 1) / slices;
    slices = (nst + tps - 1) / tps;
    const size_t b_img = (size_t)nst * 1536, b_part = sizeof(double) * (size_t)slices * (size_t)n;
    AsyncBlock ab;
    if (ab.alloc(256 + b_img + b_part, st)) return 1;
    char *blk = (char *)ab.p;
    PiScalars *sc = (PiScalars *)blk;
    int8_t *img = (int8_t *)(blk + 256);
    double *part = (double *)(blk + 256 + b_img);
    float4 *wq = (float4 *)d_work;
    JX_HIP(hipMemsetAsync(sc, 0, sizeof(PiScalars), st));
    hipLaunchKernelGGL(pi_weights_kernel, dim3((nrows + 255) / 256), dim3(256), 0, st, d_lut, d_beta, nrows, wq, sc);
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(pi_quant_w_kernel, dim3((unsigned)(((int64_t)nst * 24 + 255) / 256)), dim3(256), 0, st, wq, nrows, nst, sc, img);
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(pi_dot_kernel, dim3(gx, slices), dim3(PI_WAVES * 64), 0, st, d_t32, n, nst, tps, img, sc, part);
    JX_LAUNCH_CHECK();
    hipLaunchKernelGGL(pi_dot_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, part, n, slices, sc, d_out);
    JX_LAUNCH_CHECK();
    return 0;
}

}  // namespace jx
