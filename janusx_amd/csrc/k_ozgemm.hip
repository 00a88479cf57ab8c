// f64-accurate GEMM on the int8 matrix pipes (v_mfma_i32_32x32x32_i8): the GEMM-shaped stages of the eigensolver behind
// src/math/eigh.rs:1422-1528 (the reference calls LAPACK dsyevd: its back-transformation dormtr and the eigenvector products
// of the divide and conquer are dgemm there) without the 78.6 TFLOP/s f64 roof and without a vendor GEMM.
//
// C = alpha op(A) op(B) + beta C.  Every row of op(A) and every column of op(B) is scaled by its own largest magnitude and
// written as P signed base-254 digits (int8 "planes"):
//     x / max = d0 / 127 + d1 / (127 254) + d2 / (127 254^2) + ...,   |d| <= 127   (remainder after P digits <= 254^-(P-1) / 254)
// a'b' = sum over digit pairs (i, j) of (a_i . b_j) / (127^2 254^(i+j)); the pairs with i + j < P are kept (P (P + 1) / 2 int8
// products: 15 for P = 5), each an EXACT i32 sum (|a_i b_j| <= 127^2, K <= 2^31 / (127^2 P) per launch, longer K in several
// launches), the level sums S_l = sum_{i+j=l} are combined in f64 in the epilogue.  Error: ~254^-P relative to
// max|a_row| max|b_col| sqrt(K) (2e-12 for P = 5 at K = 20 000 on random digits), i.e. the product is as good as a dgemm whose
// operands carry 40 bits.  The int8 pipes issue 64 x the f64 MFMA rate, so 15 products stand at 4.3 x the f64 roof.
//
// Images.  An operand is sliced ONCE into the byte order the kernel's LDS-DMA copies verbatim:
//     image[row block of 128][k step of 32][plane][128 rows x 32 bytes],   byte of (row r, k) = r 32 + 16 ((k >> 4) ^ ((r >> 3) & 1)) + (k & 15)
// (the XOR makes every 16-lane group of a ds_read_b128 fragment read cover all 64 banks).  A-side and B-side images have the
// same format (rows = the C index, K contiguous), so one image of V serves V'V, V'C and (sliced the other way) V W.
//
// Kernel.  512 threads = 2 x 4 waves on a 128 x 128 tile of C, 64 x 32 per wave: P level accumulators of 2 x (32 x 32 i32) =
// 32 P accumulator registers (160 for P = 5; two waves per SIMD.  A 64 x 64 wave tile at one wave per SIMD needs 320: hipcc keeps
// MFMA accumulators in the 256 AGPRs and shuttled the rest through v_accvgpr moves, 160 per k step).  Per k step (32) a
// workgroup needs ALL planes of its 128 rows of A and B (2 P x 4 KB, one contiguous 4 P KB run per operand), copied by
// global_load_lds_dwordx4 into a 3-deep ring (raw s_barrier + counted vmcnt: two steps in flight), and issues
// P (P + 1) / 2 x 2 MFMAs per wave from it: the N-side fragments of all planes stay in registers, the M-side fragments are
// read per plane (P + 2 P ds_read_b128 per step and wave).  Per MFMA the workgroup moves (P + 1) / 2 times fewer bytes than a
// plain int8 GEMM on the concatenated planes.
// The MFMA A operand is the N-side image, so an accumulator's lane index is the memory-contiguous row index of C.
#include <stdlib.h>

#include "k_ozgemm.h"

namespace jx {

typedef int oz_i32x4 __attribute__((ext_vector_type(4)));
typedef int oz_i32x16 __attribute__((ext_vector_type(16)));

constexpr int OZ_TM = 128;                    // rows per image block = tile edge
constexpr int OZ_BK = 32;                     // k per step
constexpr int OZ_PLANE = OZ_TM * OZ_BK;       // bytes of one plane of one (row block, k step)
constexpr int OZ_NST = 3;                     // LDS ring depth

// run-time override of the plane count (0: none), set around a decomposition whose eigenvectors the caller keeps in f32.
// Per THREAD: a decomposition running on another thread of the process keeps its own plane count; a worker thread a
// decomposition starts (k_stedc.hip) inherits its parent's through oz_planes_override_set.
static thread_local int g_oz_planes_override = 0;
int oz_planes_override_get() { return g_oz_planes_override; }
void oz_planes_override_set(int planes) { g_oz_planes_override = (planes >= 4 && planes <= 6) ? planes : 0; }
int oz_planes() {
    static const int p = [] {
        const char *e = getenv("JXGPU_OZ_PLANES");
        const int v = e ? atoi(e) : 6;
        return (v >= 4 && v <= 6) ? v : 6;
    }();
    const int o = g_oz_planes_override;
    return (o >= 4 && o <= 6) ? o : p;
}

static inline size_t oz_align(size_t b) { return (b + 255) & ~(size_t)255; }

size_t oz_image_bytes(int rows, int k, int planes) {
    const size_t nrb = (size_t)ceil_div(rows, OZ_TM), nks = (size_t)ceil_div(k, OZ_BK);
    return oz_align(nrb * nks * (size_t)(planes ? planes : oz_planes()) * OZ_PLANE) + oz_align(nrb * OZ_TM * sizeof(double));
}

OzImage oz_image_at(void *mem, int rows, int k, int planes) {
    OzImage im;
    im.rows = rows;
    im.k = k;
    im.nrb = ceil_div(rows, OZ_TM);
    im.nks = ceil_div(k, OZ_BK);
    im.planes = planes ? planes : oz_planes();
    im.q = reinterpret_cast<int8_t *>(mem);
    im.scale = reinterpret_cast<double *>(reinterpret_cast<char *>(mem) +
                                          oz_align((size_t)im.nrb * im.nks * (size_t)im.planes * OZ_PLANE));
    return im;
}

// ---- row maxima ---------------------------------------------------------------------------------------------------------
// element (r, k) of the operand = x[r * rs + k * cs] (one of rs, cs is 1).  scale[r] = max_k |x| as the bit pattern of a
// non-negative double (integer order = value order; a NaN sorts above everything and so survives into the scale).
__global__ __launch_bounds__(256) void oz_rowmax_kernel(const double *__restrict__ x, int64_t rs, int64_t cs, int rows, int k,
                                                        int kchunk, unsigned long long *__restrict__ scale) {
    const int k0 = blockIdx.x * kchunk;
    const int k1 = min(k, k0 + kchunk);
    const int t = threadIdx.x;
    if (cs == 1) {
        // K contiguous: 16 rows per workgroup, 4 per wave, lanes along k (512-byte runs)
        const int wave = t >> 6, lane = t & 63;
        const int r0 = blockIdx.y * 16 + wave * 4;
        unsigned long long m[4] = {0, 0, 0, 0};
        for (int kk = k0 + lane; kk < k1; kk += 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (r0 + i < rows) {
                    const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(x[(int64_t)(r0 + i) * rs + kk]));
                    m[i] = b > m[i] ? b : m[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(m[i], off, 64);
                m[i] = o > m[i] ? o : m[i];
            }
            if (lane == 0 && m[i] && r0 + i < rows) atomicMax(scale + r0 + i, m[i]);
        }
    } else {
        // rows contiguous: 256 rows per workgroup, one per thread, 1 KB runs per k
        const int r = blockIdx.y * 256 + t;
        if (r >= rows) return;
        const double *p = x + (int64_t)r * rs;
        unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        int kk = k0;
        for (; kk + 3 < k1; kk += 4) {
            const unsigned long long b0 = (unsigned long long)__double_as_longlong(fabs(p[(int64_t)kk * cs]));
            const unsigned long long b1 = (unsigned long long)__double_as_longlong(fabs(p[(int64_t)(kk + 1) * cs]));
            const unsigned long long b2 = (unsigned long long)__double_as_longlong(fabs(p[(int64_t)(kk + 2) * cs]));
            const unsigned long long b3 = (unsigned long long)__double_as_longlong(fabs(p[(int64_t)(kk + 3) * cs]));
            m0 = b0 > m0 ? b0 : m0;
            m1 = b1 > m1 ? b1 : m1;
            m2 = b2 > m2 ? b2 : m2;
            m3 = b3 > m3 ? b3 : m3;
        }
        for (; kk < k1; ++kk) {
            const unsigned long long b0 = (unsigned long long)__double_as_longlong(fabs(p[(int64_t)kk * cs]));
            m0 = b0 > m0 ? b0 : m0;
        }
        m0 = m0 > m1 ? m0 : m1;
        m2 = m2 > m3 ? m2 : m3;
        m0 = m0 > m2 ? m0 : m2;
        if (m0) atomicMax(scale + r, m0);
    }
}

// ---- slicing ------------------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void oz_slice_kernel(const double *__restrict__ x, int64_t rs, int64_t cs, int rows, int k,
                                                       int nks, int steps_per_block, const double *__restrict__ scale,
                                                       int8_t *__restrict__ q) {
    const int rb = blockIdx.y;
    const int t = threadIdx.x;
    // K contiguous: thread = (row t >> 1, half t & 1) reads 128 contiguous bytes; rows contiguous: thread = (row t & 127,
    // half t >> 7), the 128 threads of a half read 1 KB runs
    const int rr = (cs == 1) ? (t >> 1) : (t & 127), h = (cs == 1) ? (t & 1) : (t >> 7);
    const int r = rb * OZ_TM + rr;
    const bool rok = r < rows;
    double inv = 0.0;
    if (rok) {
        const double s = scale[r];
        inv = (s > 0.0 && s < 1.0e300) ? 127.0 / s : 0.0;   // zero row, or NaN / inf in the row: digits 0 (the scale carries the NaN)
    }
    const double *p = x + (int64_t)(rok ? r : 0) * rs;
    const int ks_lo = blockIdx.x * steps_per_block, ks_hi = min(nks, ks_lo + steps_per_block);
    for (int ks = ks_lo; ks < ks_hi; ++ks) {
        const int kb = ks * OZ_BK + 16 * h;
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = kb + i;
            v[i] = (rok && kk < k) ? p[(int64_t)kk * cs] : 0.0;
        }
        int8_t o[P][16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double tq = v[i] * inv;
#pragma unroll
            for (int pl = 0; pl < P; ++pl) {
                double d = rint(tq);
                d = fmin(fmax(d, -127.0), 127.0);
                o[pl][i] = (int8_t)(int)d;
                tq = (tq - d) * 254.0;
            }
        }
        int8_t *dst = q + ((int64_t)rb * nks + ks) * (P * OZ_PLANE) + rr * 32 + 16 * (h ^ ((rr >> 3) & 1));
#pragma unroll
        for (int pl = 0; pl < P; ++pl) *reinterpret_cast<uint4 *>(dst + pl * OZ_PLANE) = *reinterpret_cast<const uint4 *>(o[pl]);
    }
}

// ---- product ------------------------------------------------------------------------------------------------------------
struct OzArgs {
    const int8_t *a, *b;        // images: a rows <-> C rows (M), b rows <-> C columns (N)
    const double *sa, *sb;
    double *c;
    int64_t ldc;
    int m, n, nks, ks0, ks1, tm, tn, srt;
    double alpha, beta;
    int mode;                   // 0 all tiles; 1 tiles with column tile >= row tile only; 2 A[m][k] = 0 for k < 128 (m / 128)
};

__device__ __forceinline__ void oz_glds(const void *src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}

template <int N> __device__ __forceinline__ void oz_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int P>
__global__ __launch_bounds__(512, 2) void oz_mm_kernel(OzArgs g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t oz_smem[];
    constexpr int STAGE = 2 * P * OZ_PLANE;       // A planes | B planes of one k step
    constexpr int PIECES = STAGE / 1024 / 8;      // 1 KiB DMA pieces per wave and stage (P)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;      // 2 (M) x 4 (N) waves, 64 x 32 per wave
    // XCD-aware tile order: the 32 workgroups an XCD runs side by side cover a super tile of 8 row tiles x 4 column tiles
    int rt, ct;
    {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        const int s = (local >> 5) * 8 + xcd, w = local & 31;
        rt = (s % g.srt) * 8 + (w & 7);
        ct = (s / g.srt) * 4 + (w >> 3);
    }
    if (rt >= g.tm || ct >= g.tn) return;
    if (g.mode == 1 && ct < rt) return;
    int ks0 = g.ks0;
    if (g.mode == 2) ks0 = max(ks0, rt * (OZ_TM / OZ_BK));
    const int nk = g.ks1 - ks0;

    const int8_t *asrc = g.a + ((int64_t)rt * g.nks + ks0) * (P * OZ_PLANE) + lane * 16;
    const int8_t *bsrc = g.b + ((int64_t)ct * g.nks + ks0) * (P * OZ_PLANE) + lane * 16 - 4 * P * 1024;
    const unsigned lds0 = (unsigned)(uintptr_t)oz_smem;
    // piece q of this wave (q = 0 .. P - 1) = piece wave + 8 q of the stage's 8 P; the first 4 P pieces are the A planes
    auto issue_piece = [&](int kstep, int buf, int q) {
        const int piece = wave + 8 * q;
        const int8_t *src = (piece < 4 * P ? asrc : bsrc) + (int64_t)kstep * (P * OZ_PLANE) + piece * 1024;
        oz_glds(src, lds0 + buf * STAGE + piece * 1024);
    };

    oz_i32x16 acc[P][2];
#pragma unroll
    for (int l = 0; l < P; ++l)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[l][bj][r] = 0;

    // fragment address of (row rho = 32 blk + (lane & 31), half lane >> 5) inside a plane
    const int frag = (lane & 31) * 32 + 16 * ((lane >> 5) ^ ((lane >> 3) & 1));

    if (nk > 0) {
#pragma unroll
        for (int q = 0; q < PIECES; ++q) issue_piece(0, 0, q);
#pragma unroll
        for (int q = 0; q < PIECES; ++q) issue_piece(min(1, nk - 1), 1, q);
    }
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        oz_wait_vm<PIECES>();                                      // stage t landed; stage t + 1 (real or dummy) stays in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // branch-free ring: past the end the last step is copied again into a buffer nothing reads any more
        const int tn2 = min(t + 2, nk - 1);
        const int nbuf = buf >= 1 ? buf - 1 : OZ_NST - 1;          // (t + 2) % 3
        const uint8_t *sa = oz_smem + buf * STAGE;
        const uint8_t *sb = sa + P * OZ_PLANE;
        oz_i32x4 bn[P];
#pragma unroll
        for (int j = 0; j < P; ++j) bn[j] = *reinterpret_cast<const oz_i32x4 *>(sb + j * OZ_PLANE + (wn * 32) * 32 + frag);
        // M-side fragments of plane i + 1 are read while plane i is multiplied
        oz_i32x4 am[2][2];
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
            am[0][bj] = *reinterpret_cast<const oz_i32x4 *>(sa + (wm * 64 + bj * 32) * 32 + frag);
        int q = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            if (i + 1 < P) {
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
                    am[(i + 1) & 1][bj] =
                        *reinterpret_cast<const oz_i32x4 *>(sa + (i + 1) * OZ_PLANE + (wm * 64 + bj * 32) * 32 + frag);
            }
#pragma unroll
            for (int j = 0; j < P - i; ++j) {
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
                    acc[i + j][bj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bn[j], am[i & 1][bj], acc[i + j][bj], 0, 0, 0);
                // the next-but-one step's DMA pieces go out between the MFMA groups
                if (q < PIECES) {
                    issue_piece(tn2, nbuf, q);
                    ++q;
                }
            }
        }
        buf = buf + 1 < OZ_NST ? buf + 1 : 0;
    }
    oz_wait_vm<0>();                                               // no DMA may outlive the workgroup's LDS allocation

    // epilogue: levels combined in f64 (Horner from the finest level), scales, alpha / beta
    const int m0 = rt * OZ_TM + wm * 64, n0 = ct * OZ_TM + wn * 32;
    const int h = lane >> 5;
    constexpr double W0 = 1.0 / (127.0 * 127.0), R254 = 1.0 / 254.0;
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int gm = m0 + bj * 32 + (lane & 31);
        const bool mok = gm < g.m;
        double sam = 0.0;
        if (mok) {
            sam = g.sa[gm];
            if (sam == 0.0) sam = 1.0;
        }
        sam *= W0 * g.alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gn = n0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mok && gn < g.n) {
                double sbn = g.sb[gn];
                if (sbn == 0.0) sbn = 1.0;
                double v = (double)acc[P - 1][bj][r];
#pragma unroll
                for (int l = P - 2; l >= 0; --l) v = v * R254 + (double)acc[l][bj][r];
                v *= sam * sbn;
                double *cp = g.c + (int64_t)gn * g.ldc + gm;
                *cp = (g.beta == 0.0) ? v : fma(g.beta, *cp, v);
            }
        }
    }
}

template <int P> static int oz_slice_launch(hipStream_t st, const double *x, int64_t rs, int64_t cs, const OzImage &im) {
    int spb = 8;         // k steps per workgroup, fewer when that leaves the chip empty
    while (spb > 1 && (int64_t)im.nrb * ceil_div(im.nks, spb) < 2048) spb /= 2;
    dim3 grid((unsigned)ceil_div(im.nks, spb), (unsigned)im.nrb);
    hipLaunchKernelGGL((oz_slice_kernel<P>), grid, dim3(256), 0, st, x, rs, cs, im.rows, im.k, im.nks, spb, im.scale, im.q);
    JX_LAUNCH_CHECK();
    return 0;
}

// operand element (r, k) = x[r * rs + k * cs], rs == 1 or cs == 1
int oz_slice(hipStream_t st, const double *x, int64_t rs, int64_t cs, const OzImage &im) {
    if (im.rows <= 0 || im.k <= 0) return 0;
    if (rs != 1 && cs != 1) return fail("oz_slice: one stride has to be 1");
    if (im.nrb > 65535) return fail("oz_slice: more than 65535 row blocks");
    JX_HIP(hipMemsetAsync(im.scale, 0, sizeof(double) * (size_t)im.nrb * OZ_TM, st));
    {
        // enough workgroups to fill the chip whatever the operand's shape
        const bool kfast = cs == 1;
        const int rgroups = kfast ? ceil_div(im.rows, 16) : ceil_div(im.rows, 256);
        int kchunk = kfast ? 2048 : 256;
        while (kchunk > 64 && (int64_t)rgroups * ceil_div(im.k, kchunk) < 2048) kchunk /= 2;
        if (rgroups > 65535) return fail("oz_slice: too many row groups");
        hipLaunchKernelGGL(oz_rowmax_kernel, dim3((unsigned)ceil_div(im.k, kchunk), (unsigned)rgroups), dim3(256), 0, st, x, rs, cs,
                           im.rows, im.k, kchunk, reinterpret_cast<unsigned long long *>(im.scale));
        JX_LAUNCH_CHECK();
    }
    switch (im.planes) {
        case 4: return oz_slice_launch<4>(st, x, rs, cs, im);
        case 5: return oz_slice_launch<5>(st, x, rs, cs, im);
        case 6: return oz_slice_launch<6>(st, x, rs, cs, im);
    }
    return fail("oz_slice: unsupported plane count");
}

template <int P> static int oz_mm_launch(hipStream_t st, const OzArgs &g) {
    static bool attr = false;
    constexpr int lds = OZ_NST * 2 * P * OZ_PLANE;
    if (!attr) {
        JX_HIP(hipFuncSetAttribute((const void *)oz_mm_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    const int sct = ceil_div(g.tn, 4);
    const int64_t supers = (int64_t)g.srt * sct;
    const int64_t groups = (supers + 7) / 8;                 // per XCD
    const int64_t blocks = groups * 32 * 8;
    if (blocks > 0x7fffffffLL) return fail("oz_mm: grid too large");
    hipLaunchKernelGGL((oz_mm_kernel<P>), dim3((unsigned)blocks), dim3(512), lds, st, g);
    JX_LAUNCH_CHECK();
    return 0;
}

// C (m x n, column-major, ldc) = alpha A B' + beta C from images (a: m rows, b: n rows, same k)
int oz_mm(hipStream_t st, const OzImage &a, const OzImage &b, int m, int n, double alpha, double beta, double *c, int64_t ldc,
          int mode) {
    if (m <= 0 || n <= 0) return 0;
    if (a.nks != b.nks || a.planes != b.planes || a.rows < m || b.rows < n) return fail("oz_mm: images do not match");
    const int P = a.planes;
    // exact i32 level sums: at most P pairs of |a b| <= 127^2 per k
    const int kmax_steps = (int)((2147483647LL / (16129LL * P)) / OZ_BK);
    OzArgs g{a.q, b.q, a.scale, b.scale, c, ldc, m, n, a.nks, 0, 0, ceil_div(m, OZ_TM), ceil_div(n, OZ_TM), 0, alpha, beta, mode};
    g.srt = ceil_div(g.tm, 8);
    bool first = true;
    for (int ks0 = 0; ks0 < a.nks || first; ks0 += kmax_steps) {
        g.ks0 = ks0;
        g.ks1 = std::min(a.nks, ks0 + kmax_steps);
        g.beta = first ? beta : 1.0;
        int rc = 1;
        switch (P) {
            case 4: rc = oz_mm_launch<4>(st, g); break;
            case 5: rc = oz_mm_launch<5>(st, g); break;
            case 6: rc = oz_mm_launch<6>(st, g); break;
        }
        if (rc) return rc;
        first = false;
    }
    return 0;
}

// C = alpha op(A) op(B) + beta C with the operands sliced into `work` (>= oz_image_bytes(m, k) + oz_image_bytes(n, k))
int oz_dgemm(hipStream_t st, bool ta, bool tb, int m, int n, int k, double alpha, const double *a, int64_t lda, const double *b,
             int64_t ldb, double beta, double *c, int64_t ldc, void *work, size_t work_bytes) {
    if (m <= 0 || n <= 0) return 0;
    if (k <= 0) return fail("oz_dgemm: k must be > 0");
    const size_t ba = oz_image_bytes(m, k), bb = oz_image_bytes(n, k);
    if (work_bytes < ba + bb) return fail("oz_dgemm: workspace too small");
    OzImage ia = oz_image_at(work, m, k), ib = oz_image_at(reinterpret_cast<char *>(work) + ba, n, k);
    // op(A) (m x k): element (r, kk) = a[r + kk lda] (N) or a[kk + r lda] (T); op(B)' rows = columns of op(B):
    // element (j, kk) = b[kk + j ldb] (N) or b[j + kk ldb] (T)
    if (oz_slice(st, a, ta ? lda : 1, ta ? 1 : lda, ia)) return 1;
    if (oz_slice(st, b, tb ? 1 : ldb, tb ? ldb : 1, ib)) return 1;
    return oz_mm(st, ia, ib, m, n, alpha, beta, c, ldc, 0);
}

}  // namespace jx

using namespace jx;

// Diagnostic / test entry: C = alpha op(A) op(B) + beta C (column-major device matrices) through the sliced int8 path.
// h_ms (optional, 3 floats): slicing of A, slicing of B, product [ms].
extern "C" int jxg_oz_dgemm_f64(int ta, int tb, int m, int n, int k, double alpha, const double *d_a, int64_t lda,
                                const double *d_b, int64_t ldb, double beta, double *d_c, int64_t ldc, float *h_ms, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    if (m <= 0 || n <= 0 || k <= 0) return fail("jxg_oz_dgemm_f64: m, n, k must be > 0");
    DevBuf work;
    const size_t ba = oz_image_bytes(m, k), bb = oz_image_bytes(n, k);
    if (work.alloc(ba + bb)) return 1;
    OzImage ia = oz_image_at(work.p, m, k), ib = oz_image_at(work.as<char>() + ba, n, k);
    hipEvent_t ev[4];
    for (auto &e : ev) JX_HIP(hipEventCreate(&e));
    JX_HIP(hipEventRecord(ev[0], st));
    if (oz_slice(st, d_a, ta ? lda : 1, ta ? 1 : lda, ia)) return 1;
    JX_HIP(hipEventRecord(ev[1], st));
    if (oz_slice(st, d_b, tb ? 1 : ldb, tb ? ldb : 1, ib)) return 1;
    JX_HIP(hipEventRecord(ev[2], st));
    if (oz_mm(st, ia, ib, m, n, alpha, beta, d_c, ldc, 0)) return 1;
    JX_HIP(hipEventRecord(ev[3], st));
    JX_HIP(hipStreamSynchronize(st));
    if (h_ms)
        for (int i = 0; i < 3; ++i) JX_HIP(hipEventElapsedTime(&h_ms[i], ev[i], ev[i + 1]));
    for (auto &e : ev) (void)hipEventDestroy(e);
    return 0;
}

extern "C" int jxg_oz_planes(void) { return oz_planes(); }
extern "C" int jxg_oz_set_planes(int planes) {
    const int prev = g_oz_planes_override;
    g_oz_planes_override = (planes >= 4 && planes <= 6) ? planes : 0;
    return prev;
}
