// f64 GEMM family on the matrix pipes (v_mfma_f64_16x16x4_f64) for the eigensolver behind src/math/eigh.rs:1422-1528
// (the reference calls LAPACK dsyevd; every O(n^3) stage of the two-stage reduction here is one of these products):
//   dgemm      C = alpha op(A) op(B) + beta C          (NN / TN / NT / TT, optional split over K: per-slice partial
//              products summed in a fixed order by a second kernel, bit-reproducible)
//   dsymm_l    C = alpha A B + beta C, A symmetric with only its LOWER triangle stored (band reduction: Z = A22 V)
//   dsyr2k_l   lower tiles of C += alpha (A B' ...) as one NT product over concatenated panels (trailing update)
// All matrices are column-major.  One workgroup = 512 threads = 4 x 2 waves on a BM x BN tile, K in steps of 16 through a
// double-buffered LDS image [k][x] (row pitch BM + 17 doubles: an odd pitch makes the k-fast staging stores of a
// transposed operand conflict-free, and the two k-rows a 32-lane read group touches land 34 banks apart); global loads
// of step t + 1 are in flight in registers while step t is multiplied.  The MFMA "A" operand is fed with op(B) and the
// "B" operand with op(A), so a lane's accumulator column index is the memory-contiguous row index of C: each store
// instruction writes four 128-byte runs.  f64 MFMA issues one 16x16x4 block per 64 cycles per SIMD (the f64 vector
// rate, 78.6 TFLOP/s per chip): LDS and address arithmetic hide behind it, HBM does as long as an operand element is
// reused >= 16 times, which every call site here satisfies.
#include <stdlib.h>

#include "jx_common.h"

namespace jx {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int DG_BK = 16;
constexpr int DG_THREADS = 512;   // 8 waves = 4 (M) x 2 (N): <= 128 VGPRs per lane, two workgroups per CU

struct DgemmArgs {
    const double *a;
    const double *b;
    double *c;
    int64_t lda, ldb, ldc;
    int m, n, k;
    double alpha, beta;
    int ta, tb;        // operand stored transposed
    int symm_a;        // A is m x m symmetric, lower triangle stored (k == m, ta ignored)
    int lower_tiles;   // grid enumerates the tiles (ti >= tj) of a square C only (BM == BN)
    int ksplit;        // > 1: blockIdx.z owns a K range and writes its partial product to ws[z] (m x n, ld = m); a second
                       // kernel sums the slices in a fixed order (bit-reproducible, unlike f64 atomics)
    double *ws;
};

// Branch-free: every lane always issues its NL loads from a clamped (valid) address and zeroes the out-of-range elements
// afterwards, so the loads of a step are independent instructions behind ONE wait (the former per-element `if` made hipcc
// wait for each load before the next address was formed: eight exposed memory latencies per K step).
// Returns the bit mask of the in-range elements: the zeroing is left to dg_store_tile (behind the MFMAs of the step), so
// that nothing consumes a loaded value while the step is multiplied.
template <int BX, int NT = DG_THREADS>
__device__ __forceinline__ unsigned dg_load_tile(const double *__restrict__ p, int64_t ld, bool trans, bool symm, int x0,
                                                 int xmax, int k0, int kmax, double (&r)[BX * DG_BK / NT]) {
    // element (x, kk) of the BX x 16 operand tile; storage: !trans -> p[x + kk * ld], trans -> p[kk + x * ld];
    // lower-stored symmetric operand: (x, kk) = p[max + min * ld], tiles entirely above the diagonal take the k-fast
    // (coalesced along the stored columns) thread mapping, the others the x-fast one
    constexpr int NL = BX * DG_BK / NT;
    const int t = threadIdx.x;
    const bool kfast = symm ? (x0 + BX <= k0) : trans;
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = i * NT + t;
        const int xo = kfast ? idx / DG_BK : idx % BX;
        const int ko = kfast ? idx % DG_BK : idx / BX;
        const int x = x0 + xo, kk = k0 + ko;
        const bool ok = x < xmax && kk < kmax;
        const int xc = min(x, xmax - 1), kc = min(kk, kmax - 1);
        int64_t off;
        if (symm) off = (int64_t)max(xc, kc) + (int64_t)min(xc, kc) * ld;
        else off = trans ? (int64_t)kc + (int64_t)xc * ld : (int64_t)xc + (int64_t)kc * ld;
        r[i] = p[off];
        okmask |= ok ? (1u << i) : 0u;
    }
    return okmask;
}

template <int BX, int NT = DG_THREADS>
__device__ __forceinline__ void dg_store_tile(double *__restrict__ s, bool kfast, const double (&r)[BX * DG_BK / NT],
                                              unsigned okmask) {
    constexpr int NL = BX * DG_BK / NT;
    constexpr int PITCH = BX + 17;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = i * NT + t;
        const int pos = kfast ? (idx % DG_BK) * PITCH + idx / DG_BK : (idx / BX) * PITCH + idx % BX;
        s[pos] = ((okmask >> i) & 1u) ? r[i] : 0.0;
    }
}

// Steady-state loader: inside the K range (no K tail) and, for the lower-stored symmetric operand, away from the diagonal the
// address of element i of step t + 1 is the address of step t plus a constant, and the in-range mask of x does not change:
// one 64-bit add per element and step instead of the clamps and the 64-bit index product of dg_load_tile (by ablation --
// scripts/probes/dgemm_probe.hip -- the loop skeleton reaches 69 TFLOP/s, the general loader held the kernel at 52).
template <int BX, int NT>
struct DgStream {
    static constexpr int NL = BX * DG_BK / NT;
    const char *base;        // wave-uniform: origin of the current step's tile (an SGPR pair, advanced by `step` per K step)
    unsigned off[NL];        // byte offset of this lane's elements from it (< 128 ld doubles: 32 bits)
    int64_t step;            // bytes per K step (uniform)
    unsigned xmask;
    __device__ __forceinline__ void init(const double *__restrict__ mat, int64_t ld, bool kfast, int x0, int xmax, int k0) {
        const int t = threadIdx.x;
        xmask = 0;
        base = reinterpret_cast<const char *>(mat + (kfast ? (int64_t)k0 + (int64_t)x0 * ld : (int64_t)x0 + (int64_t)k0 * ld));
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = i * NT + t;
            const int xo = kfast ? idx / DG_BK : idx % BX;
            const int ko = kfast ? idx % DG_BK : idx / BX;
            const int xoc = min(x0 + xo, xmax - 1) - x0;                  // clamped: always a valid address
            off[i] = (unsigned)(8 * (kfast ? (int64_t)ko + (int64_t)xoc * ld : (int64_t)xoc + (int64_t)ko * ld));
            xmask |= (x0 + xo < xmax) ? (1u << i) : 0u;
        }
        step = 8 * (kfast ? (int64_t)DG_BK : (int64_t)DG_BK * ld);
    }
    __device__ __forceinline__ void load(double (&r)[NL]) {
#pragma unroll
        for (int i = 0; i < NL; ++i) r[i] = *reinterpret_cast<const double *>(base + off[i]);
        base += step;
    }
};

// STREAM: K loop split into runs of steady-state steps (DgStream) and general steps; it wants ~230 registers (two waves per
// SIMD = one workgroup per CU), which pays for long K loops (plain GEMM 52 -> 56 TFLOP/s, the symmetric product 38 -> 45) and
// costs the short-K rank-2k update its second resident workgroup (40 -> 34): that one keeps the single general loop at four
// waves per SIMD.
template <int BM, int BN, int NT = DG_THREADS, bool STREAM = true>
__global__ __launch_bounds__(NT, STREAM ? 2 : (NT == 512 ? 4 : 2)) void dgemm_kernel(DgemmArgs g) {
    constexpr int PA = BM + 17, PB = BN + 17;
    constexpr int WGM = NT == 512 ? 4 : 2;           // waves along M (x 2 along N)
    constexpr int WM = BM / WGM, WN = BN / 2;        // wave tile
    constexpr int MB = WM / 16, NB = WN / 16;        // 16x16 blocks per wave
    extern __shared__ __attribute__((aligned(16))) double dg_smem[];
    double *as = dg_smem;                            // [2][16][PA]
    double *bs = dg_smem + 2 * DG_BK * PA;           // [2][16][PB]

    int ti, tj;
    if (g.lower_tiles) {
        // linear index -> (ti >= tj): ti = floor((sqrt(8 b + 1) - 1) / 2)
        const int bid = blockIdx.x;
        int r = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
        while ((int64_t)(r + 1) * (r + 2) / 2 <= bid) ++r;
        while ((int64_t)r * (r + 1) / 2 > bid) --r;
        ti = r;
        tj = bid - (int)((int64_t)r * (r + 1) / 2);
    } else {
        ti = blockIdx.x;
        tj = blockIdx.y;
    }
    const int m0 = ti * BM, n0 = tj * BN;
    // K range of this workgroup
    int kbeg = 0, kend = g.k;
    if (g.ksplit > 1) {
        const int ktiles = (g.k + DG_BK - 1) / DG_BK;
        const int per = (ktiles + g.ksplit - 1) / g.ksplit;
        kbeg = (int)blockIdx.z * per * DG_BK;
        kend = min(g.k, kbeg + per * DG_BK);
        if (kbeg > kend) kbeg = kend;                 // an empty slice still writes its zeros
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave % WGM) * WM, wn = (wave / WGM) * WN;
    const int lx = lane & 15, lk = lane >> 4;

    d4 acc[NB][MB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[j][i] = (d4){0.0, 0.0, 0.0, 0.0};

    double ra[BM * DG_BK / NT], rb[BN * DG_BK / NT];
    const bool symm = g.symm_a != 0;
    const bool ta = g.ta != 0, tb_kfast = g.tb == 0;   // B stored (k, n): k contiguous -> k-fast mapping
    auto a_kfast = [&](int k0) -> bool {
        if (symm) return (m0 + BM <= k0);              // tile entirely above the diagonal: read transposed
        return ta;
    };
    unsigned oka = 0, okb = 0;
    DgStream<BM, NT> sa;
    DgStream<BN, NT> sb;
    static_assert(BM >= DG_BK, "tile rows");
    // regime of the K step that starts at k: 0 = A stored as (x, k) (plain, or symmetric left of the tile's rows), 1 = A read
    // transposed (ta, or symmetric right of the tile's rows), 2 = general loader (K tail, symmetric step crossing the diagonal)
    auto mode_of = [&](int k) -> int {
        if (k + DG_BK > kend) return 2;
        if (!symm) return ta ? 1 : 0;
        if (k + DG_BK <= m0) return 0;
        if (k >= m0 + BM) return 1;
        return 2;
    };
    int buf = 0;
    auto multiply = [&]() {
        const double *ap = as + buf * DG_BK * PA + wm + lx;
        const double *bp = bs + buf * DG_BK * PB + wn + lx;
#pragma unroll
        for (int ks = 0; ks < DG_BK; ks += 4) {
            double fa[MB], fb[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) fa[i] = ap[(ks + lk) * PA + i * 16];
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[j] = bp[(ks + lk) * PB + j * 16];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[j][i], 0, 0, 0);
        }
    };
    // one K step with the GENERAL loader for the next one (or none)
    auto step_general = [&](int k0) {
        const bool more = k0 + DG_BK < kend;
        if (more) {
            oka = dg_load_tile<BM, NT>(g.a, g.lda, ta, symm, m0, g.m, k0 + DG_BK, kend, ra);
            // op(B)(kk, x): !tb -> b[kk + x ldb] (k contiguous = "trans" mapping of the loader), tb -> b[x + kk ldb]
            okb = dg_load_tile<BN, NT>(g.b, g.ldb, g.tb == 0, false, n0, g.n, k0 + DG_BK, kend, rb);
        }
        multiply();
        if (more) {
            dg_store_tile<BM, NT>(as + (buf ^ 1) * DG_BK * PA, a_kfast(k0 + DG_BK), ra, oka);
            dg_store_tile<BN, NT>(bs + (buf ^ 1) * DG_BK * PB, tb_kfast, rb, okb);
        }
        __syncthreads();
        buf ^= 1;
    };
    oka = dg_load_tile<BM, NT>(g.a, g.lda, ta, symm, m0, g.m, kbeg, kend, ra);
    okb = dg_load_tile<BN, NT>(g.b, g.ldb, g.tb == 0, false, n0, g.n, kbeg, kend, rb);
    dg_store_tile<BM, NT>(as, a_kfast(kbeg), ra, oka);
    dg_store_tile<BN, NT>(bs, tb_kfast, rb, okb);
    __syncthreads();
    int k0 = kbeg;
    while (k0 < kend) {
        const int kn = k0 + DG_BK;
        const int md = (STREAM && kn < kend) ? mode_of(kn) : 2;
        if (!STREAM || md == 2) {
            step_general(k0);
            k0 = kn;
            continue;
        }
        // run of steps whose NEXT step is in regime md: the streams advance by a constant per step (nothing but the loads,
        // the MFMAs and the staging stores inside this loop)
        int k_hi = kbeg + ((kend - kbeg) / DG_BK - 1) * DG_BK;     // last step that lies inside the K range
        if (symm && md == 0) k_hi = min(k_hi, m0 - DG_BK);
        sa.init(g.a, g.lda, md == 1, m0, g.m, kn);
        sb.init(g.b, g.ldb, g.tb == 0, n0, g.n, kn);
        const bool akf = md == 1;
        for (; k0 + DG_BK <= k_hi; k0 += DG_BK) {
            sa.load(ra);
            sb.load(rb);
            multiply();
            dg_store_tile<BM, NT>(as + (buf ^ 1) * DG_BK * PA, akf, ra, sa.xmask);
            dg_store_tile<BN, NT>(bs + (buf ^ 1) * DG_BK * PB, tb_kfast, rb, sb.xmask);
            __syncthreads();
            buf ^= 1;
        }
    }
    // epilogue: acc[j][i][r] = C[m0 + wm + 16 i + lx][n0 + wn + 16 j + lk + 4 r]
    const bool split = g.ksplit > 1;
    double *slice = split ? g.ws + (int64_t)blockIdx.z * g.m * g.n : nullptr;
    const bool rmw = !split && g.beta != 0.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        // the C values of this 16-column strip first (clamped, always valid addresses: MB x 4 independent loads behind one
        // wait), then the stores -- the former element-by-element read-modify-write exposed one memory latency per element
        double cv[4][MB];
        if (rmw) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = min(n0 + wn + 16 * j + lk + 4 * r, g.n - 1);
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const int row = min(m0 + wm + 16 * i + lx, g.m - 1);
                    cv[r][i] = g.c[row + (int64_t)col * g.ldc];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = n0 + wn + 16 * j + lk + 4 * r;
            if (col >= g.n) continue;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int row = m0 + wm + 16 * i + lx;
                if (row >= g.m) continue;
                if (g.lower_tiles && row < col) continue;     // diagonal tiles: the strict upper part is not referenced
                const double v = g.alpha * acc[j][i][r];
                if (split) slice[row + (int64_t)col * g.m] = v;
                else g.c[row + (int64_t)col * g.ldc] = rmw ? (v + g.beta * cv[r][i]) : v;
            }
        }
    }
}

// ---- rank-2k update of the band reduction as ONE stream of K steps over a workgroup's tiles (round 6) -----------------------------
// C(lower tiles) = alpha A B' + beta C with A, B stored (x, k) (x contiguous), K a multiple of 16 (the trailing update: K = 128).
// At K = 128 a 128 x 128 tile is 13.8 us of products between a prologue (first operand step: one memory round trip) and an epilogue
// that reads and writes the 128 KB of C -- as many HBM cycles as the products take matrix-pipe cycles -- and in dgemm_kernel nothing
// overlaps the two (0.39 - 0.43 of the f64 peak at one workgroup per CU; a second resident workgroup starved the panel chain on
// the other stream, round 4).  Here a workgroup walks its tiles as one sequence of K steps: the operand loads of a step are always one
// step ahead, ACROSS tile boundaries (no prologue bubble), the C values of a tile are requested a strip per step over its first steps
// (the loads of a wave return in order: all of them in front of the first step put the next step's operands behind 128 KB of C) and
// consumed behind its last, the stores of the finished tile drain behind the next tile's products.  Same products in the same order,
// same alpha acc + beta c expression: the same bits as dgemm_kernel.
__global__ __launch_bounds__(512, 2) void dsyr2k_pipe_kernel(DgemmArgs g) {
    constexpr int BM = 128, BN = 128, NT = 512;
    constexpr int PA = BM + 17, PB = BN + 17;
    constexpr int WM = 32, WN = 64, MB = 2, NB = 4;
    constexpr int NL = BM * DG_BK / NT;              // 4 elements per thread, operand and step
    extern __shared__ __attribute__((aligned(16))) double dg_smem[];
    double *as = dg_smem;                            // [2][16][PA]
    double *bs = dg_smem + 2 * DG_BK * PA;           // [2][16][PB]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave % 4) * WM, wn = (wave / 4) * WN;
    const int lx = lane & 15, lk = lane >> 4;
    // Tiles by XCD: workgroup b runs on XCD b % 8 (round-robin dispatch).  XCD x owns the tile COLUMNS 8 q + (q odd ? 7 - x : x): its
    // ~T / 8 B panels (128 KB each: 2.5 MB at n = 20 000) stay in its 4 MB L2 for the whole launch and the A panel of the row in
    // progress is shared by the XCD's workgroups.  The XCD's tiles, row-major, are dealt round-robin to its workgroups
    // (slot = b / 8 of gridDim.x / 8).
    const int T = (g.m + BM - 1) / BM;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = max(1, (int)gridDim.x >> 3);
    auto col_of = [&](int j) { return 8 * j + ((j & 1) ? 7 - xcd : xcd); };
    auto count_of = [&](int row) {
        const int nb = (row + 1) >> 3;
        return nb + ((8 * nb <= row && col_of(nb) <= row) ? 1 : 0);
    };
    int it_row = 0, it_j = slot, it_p = 0;
    auto next_tile = [&](int &ti_o, int &tj_o) -> bool {
        for (;;) {
            if (it_row >= T) return false;
            const int cnt = count_of(it_row);
            if (it_j < cnt) {
                ti_o = it_row;
                tj_o = col_of(it_j);
                it_j += nslot;
                return true;
            }
            it_p += cnt;
            ++it_row;
            it_j = ((slot - it_p) % nslot + nslot) % nslot;
        }
    };
    int ti, tj;
    if (!next_tile(ti, tj)) return;
    const int nks = g.k / DG_BK;
    // operand element i of this thread at a step: x = t % 128 (clamped), k = 4 i + t / 128.  Addresses: a per-thread origin per
    // tile (row x of the panel), the same four 32-bit byte offsets for every step and tile (k rows 4 i + t / 128), and a
    // wave-uniform byte offset of the step
    const int xo = t & 127, ko = t >> 7;
    unsigned offa[NL], offb[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        offa[i] = (unsigned)(8 * (int64_t)(4 * i + ko) * g.lda);
        offb[i] = (unsigned)(8 * (int64_t)(4 * i + ko) * g.ldb);
    }
    const int64_t stepa = 8 * (int64_t)DG_BK * g.lda, stepb = 8 * (int64_t)DG_BK * g.ldb;
    auto load_step = [&](int m0, int n0, int k0, double (&ra)[NL], double (&rb)[NL]) {
        const int xa = min(m0 + xo, g.m - 1), xb = min(n0 + xo, g.n - 1);
        const char *pa = reinterpret_cast<const char *>(g.a + xa) + (int64_t)(k0 / DG_BK) * stepa;
        const char *pb = reinterpret_cast<const char *>(g.b + xb) + (int64_t)(k0 / DG_BK) * stepb;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            ra[i] = *reinterpret_cast<const double *>(pa + offa[i]);
            rb[i] = *reinterpret_cast<const double *>(pb + offb[i]);
        }
    };
    auto store_step = [&](int m0, int n0, int buf, const double (&ra)[NL], const double (&rb)[NL]) {
        const bool oka = m0 + xo < g.m, okb = n0 + xo < g.n;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            as[buf * DG_BK * PA + (4 * i + ko) * PA + xo] = oka ? ra[i] : 0.0;
            bs[buf * DG_BK * PB + (4 * i + ko) * PB + xo] = okb ? rb[i] : 0.0;
        }
    };
    double ra[NL], rb[NL];
    int buf = 0;
    load_step(ti * BM, tj * BN, 0, ra, rb);
    store_step(ti * BM, tj * BN, 0, ra, rb);
    __syncthreads();
    for (;;) {
        const int m0 = ti * BM, n0 = tj * BN;
        int ti2 = ti, tj2 = tj;
        const bool has_next = next_tile(ti2, tj2);
        d4 acc[NB][MB];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < MB; ++i) acc[j][i] = (d4){0.0, 0.0, 0.0, 0.0};
        double cv[NB][4][MB];
        const bool rmw = g.beta != 0.0;
        auto c_load = [&](int j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = min(n0 + wn + 16 * j + lk + 4 * r, g.n - 1);
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const int row = min(m0 + wm + 16 * i + lx, g.m - 1);
                    cv[j][r][i] = g.c[row + (int64_t)col * g.ldc];
                }
            }
        };
        for (int ks = 0; ks < nks; ++ks) {
            const bool last = ks + 1 == nks;
            const bool more = !last || has_next;
            const int lm0 = last ? ti2 * BM : m0, ln0 = last ? tj2 * BN : n0, lk0 = last ? 0 : (ks + 1) * DG_BK;
            if (more) load_step(lm0, ln0, lk0, ra, rb);
            if (rmw) {
                // NB strips over the first steps (K = 128: steps 0 .. 3 of 8; fewer steps than strips: the rest with the last step)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (ks == j || (last && j > ks)) c_load(j);
            }
            {
                const double *ap = as + buf * DG_BK * PA + wm + lx;
                const double *bp = bs + buf * DG_BK * PB + wn + lx;
#pragma unroll
                for (int k4 = 0; k4 < DG_BK; k4 += 4) {
                    double fa[MB], fb[NB];
#pragma unroll
                    for (int i = 0; i < MB; ++i) fa[i] = ap[(k4 + lk) * PA + i * 16];
#pragma unroll
                    for (int j = 0; j < NB; ++j) fb[j] = bp[(k4 + lk) * PB + j * 16];
#pragma unroll
                    for (int j = 0; j < NB; ++j)
#pragma unroll
                        for (int i = 0; i < MB; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[j][i], 0, 0, 0);
                }
            }
            if (more) store_step(lm0, ln0, buf ^ 1, ra, rb);
            __syncthreads();
            buf ^= 1;
        }
        // epilogue of the tile (the next tile's first step is already in LDS)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = n0 + wn + 16 * j + lk + 4 * r;
                if (col >= g.n) continue;
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const int row = m0 + wm + 16 * i + lx;
                    if (row >= g.m) continue;
                    if (row < col) continue;                  // diagonal tiles: the strict upper part is not referenced
                    const double v = g.alpha * acc[j][i][r];
                    g.c[row + (int64_t)col * g.ldc] = rmw ? (v + g.beta * cv[j][r][i]) : v;
                }
            }
        if (!has_next) break;
        ti = ti2;
        tj = tj2;
    }
}

template <int BM, int BN, int NT = DG_THREADS, bool STREAM = true>
static int dg_launch(const DgemmArgs &g, dim3 grid, hipStream_t st) {
    constexpr size_t smem = sizeof(double) * 2 * DG_BK * ((BM + 17) + (BN + 17));
    static bool attr_set = false;
    if (!attr_set) {
        JX_HIP(hipFuncSetAttribute((const void *)dgemm_kernel<BM, BN, NT, STREAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    hipLaunchKernelGGL((dgemm_kernel<BM, BN, NT, STREAM>), grid, dim3(NT), smem, st, g);
    JX_LAUNCH_CHECK();
    return 0;
}

// C = beta C + sum_z ws[z], bit-reproducible: ZL partial sums per element (lane zl adds the slices z = zl, zl + ZL, ... in index
// order) combined in lane order.  The loop is unrolled so that eight slice loads are in flight per thread (one load per
// dependent add took 21 us for a 64 x 64 Gram block in 136 slices and 58 us for the n_t x 64 symmetric product: the
// latency of `ksplit` loads in a row, 1540 launches per band reduction at n = 20 000); small results get eight lanes per
// element so that they spread over 8 x as many workgroups.
template <int ZL>
__global__ __launch_bounds__(256) void dg_reduce_kernel(double *c, int64_t ldc, int m, int n, double beta, const double *ws, int ksplit) {
    constexpr int EPB = 256 / ZL;                              // elements per workgroup
    __shared__ double part[ZL > 1 ? 256 : 1];
    const int el = threadIdx.x % EPB, zl = threadIdx.x / EPB;
    const int64_t i = (int64_t)blockIdx.x * EPB + el, mn = (int64_t)m * n;
    const bool in = i < mn;
    double acc = 0.0;
    if (in) {
        const double *p = ws + i + (int64_t)zl * mn;
        const int64_t stride = (int64_t)ZL * mn;
#pragma unroll 8
        for (int z = zl; z < ksplit; z += ZL, p += stride) acc += *p;
    }
    if (ZL > 1) {
        part[threadIdx.x] = acc;
        __syncthreads();
        if (zl != 0) return;
#pragma unroll
        for (int q = 1; q < ZL; ++q) acc += part[q * EPB + el];
    }
    if (!in) return;
    const int r = (int)(i % m), col = (int)(i / m);
    double *p = c + r + (int64_t)col * ldc;
    *p = (beta == 0.0) ? acc : (acc + beta * *p);
}

__global__ void dg_scale_kernel(double *c, int64_t ldc, int m, int n, double beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)m * n) return;
    const int r = (int)(i % m), col = (int)(i / m);
    double *p = c + r + (int64_t)col * ldc;
    *p = (beta == 0.0) ? 0.0 : beta * *p;
}

static int dg_fit_split(int ksplit, int m, int n, size_t ws_doubles) {
    const size_t per = (size_t)m * (size_t)n;
    if (ksplit > 1 && per * (size_t)ksplit > ws_doubles) ksplit = (int)(ws_doubles / per);
    return ksplit < 2 ? 1 : ksplit;
}

static int dg_reduce(hipStream_t st, const DgemmArgs &g) {
    if ((int64_t)g.m * g.n > 0xffffff00LL) return fail("dgemm: result beyond 2^32 elements (one dispatch dimension)");
    const int64_t mn = (int64_t)g.m * g.n;
    if (mn <= 16384 && g.ksplit >= 16)
        hipLaunchKernelGGL(dg_reduce_kernel<8>, dim3((unsigned)((mn + 31) / 32)), dim3(256), 0, st, g.c, g.ldc, g.m, g.n, g.beta, g.ws,
                           g.ksplit);
    else
        hipLaunchKernelGGL(dg_reduce_kernel<1>, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st, g.c, g.ldc, g.m, g.n, g.beta, g.ws,
                           g.ksplit);
    JX_LAUNCH_CHECK();
    return 0;
}

// C (m x n) = alpha op(A) op(B) + beta C.  ksplit <= 0: chosen so that the launch fills the chip; a split over K needs
// `ws` (ksplit m n doubles; the split is reduced to what fits, 1 without a workspace).
int dgemm(hipStream_t st, bool ta, bool tb, int m, int n, int k, double alpha, const double *a, int64_t lda,
          const double *b, int64_t ldb, double beta, double *c, int64_t ldc, int ksplit, double *ws, size_t ws_doubles) {
    if (m <= 0 || n <= 0) return 0;
    DgemmArgs g{a, b, c, lda, ldb, ldc, m, n, k, alpha, beta, ta ? 1 : 0, tb ? 1 : 0, 0, 0, 1, ws};
    if (k <= 0) {
        hipLaunchKernelGGL(dg_scale_kernel, dim3((unsigned)(((int64_t)m * n + 255) / 256)), dim3(256), 0, st, c, ldc, m, n, beta);
        JX_LAUNCH_CHECK();
        return 0;
    }
    const bool small_n = n <= 64, small_m = m <= 64;
    const int bm = small_m ? 64 : 128, bn = (small_n || small_m) ? 64 : 128;
    const int tm = ceil_div(m, bm), tn = ceil_div(n, bn);
    if (ksplit <= 0) {
        const int64_t tiles = (int64_t)tm * tn;
        const int ktiles = ceil_div(k, DG_BK);
        ksplit = 1;
        if (tiles < 384) {
            ksplit = (int)((768 + tiles - 1) / tiles);
            const int maxsplit = ktiles / 8 > 0 ? ktiles / 8 : 1;    // at least 128 of K per slice
            if (ksplit > maxsplit) ksplit = maxsplit;
            if (ksplit < 1) ksplit = 1;
        }
    }
    g.ksplit = dg_fit_split(ksplit, m, n, ws ? ws_doubles : 0);
    dim3 grid(tm, tn, g.ksplit);
    int rc;
    if (bm == 128 && bn == 128) rc = dg_launch<128, 128>(g, grid, st);
    else if (bm == 128) rc = dg_launch<128, 64>(g, grid, st);
    else rc = dg_launch<64, 64>(g, grid, st);
    if (rc) return rc;
    return g.ksplit > 1 ? dg_reduce(st, g) : 0;
}

// C (m x n) = alpha A B + beta C with A (m x m) symmetric, lower triangle stored
int dsymm_lower(hipStream_t st, int m, int n, double alpha, const double *a, int64_t lda, const double *b, int64_t ldb,
                double beta, double *c, int64_t ldc, double *ws, size_t ws_doubles) {
    if (m <= 0 || n <= 0) return 0;
    DgemmArgs g{a, b, c, lda, ldb, ldc, m, n, m, alpha, beta, 0, 0, 1, 0, 1, ws};
    const bool wide = n > 64;
    const int bn = wide ? 128 : 64;
    const int tm = ceil_div(m, 128), tn = ceil_div(n, bn);
    const int64_t tiles = (int64_t)tm * tn;
    int ksplit = 1;
    if (tiles < 384) {
        // two workgroups fit a CU: with `slots` resident workgroups the product takes ceil(tiles ks / slots) rounds of a
        // 1 / ks share of K each -- take the smallest ks within 3 % of the best (fewer slices to reduce);
        // JXGPU_DSYMM_SPLIT=768 restores the former "fill 768 slots" rule
        static const int rule = getenv("JXGPU_DSYMM_SPLIT") ? atoi(getenv("JXGPU_DSYMM_SPLIT")) : 0;
        const int maxsplit = ceil_div(m, DG_BK) / 8 > 0 ? ceil_div(m, DG_BK) / 8 : 1;
        if (rule > 0) {
            ksplit = (int)((rule + tiles - 1) / tiles);
        } else {
            static const int64_t slots = getenv("JXGPU_DSYMM_SLOTS") ? atoll(getenv("JXGPU_DSYMM_SLOTS")) : 512;
            double best = 1e300;
            for (int ks = 1; ks <= 16 && ks <= maxsplit; ++ks) {
                const double cost = (double)((tiles * ks + slots - 1) / slots) / (double)ks;
                if (cost < 0.97 * best) {
                    best = cost;
                    ksplit = ks;
                }
            }
        }
        if (ksplit > maxsplit) ksplit = maxsplit;
    }
    g.ksplit = dg_fit_split(ksplit, m, n, ws ? ws_doubles : 0);
    dim3 grid(tm, tn, g.ksplit);
    const int rc = wide ? dg_launch<128, 128>(g, grid, st) : dg_launch<128, 64>(g, grid, st);
    if (rc) return rc;
    return g.ksplit > 1 ? dg_reduce(st, g) : 0;
}

// lower tiles of C (m x m) = alpha A B' + beta C, A and B (m x k): with A = [V | W], B = [W | V] this is the symmetric
// rank-2k update C + alpha (V W' + W V'); the strict upper triangle of C is not referenced
int dsyr2k_lower_nt(hipStream_t st, int m, int k, double alpha, const double *a, int64_t lda, const double *b, int64_t ldb,
                    double beta, double *c, int64_t ldc) {
    if (m <= 0) return 0;
    DgemmArgs g{a, b, c, lda, ldb, ldc, m, m, k, alpha, beta, 0, 1, 0, 1, 1, nullptr};
    const int t = ceil_div(m, 128);
    // one stream of K steps over a workgroup's tiles (dsyr2k_pipe_kernel) from two tiles per workgroup on; JXGPU_SYR2K_PIPE=0: one
    // tile per workgroup (dgemm_kernel)
    const char *pe = getenv("JXGPU_SYR2K_PIPE");         // read per call: the two forms are compared inside one process by the tests
    const bool pipe = !(pe && atoi(pe) == 0);
    // its workgroups stay for the whole launch (one per CU: 216 registers), so the panel chain of the NEXT panel -- on the other
    // stream, the critical path -- only finds the CUs this kernel leaves alone: 7 / 8 of them measured best at n = 20 000 (band
    // reduction, pipe on 256 / 248 / 240 / 232 CUs: 372 - 382 ms, on 224: 336, 216: 339, 208: 342, 192: 352; dgemm_kernel: 355)
    static const int pipe_wgs = [] {
        const char *e = getenv("JXGPU_SYR2K_PIPE_WGS");
        int v = e ? atoi(e) : 0;
        if (v < 8) {
            int dev = 0;
            hipDeviceProp_t prop;
            v = 224;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 16)
                v = prop.multiProcessorCount * 7 / 8;
        }
        return v >= 8 ? (v / 8) * 8 : 224;
    }();
    const int64_t ntl = (int64_t)t * (t + 1) / 2;
    if (pipe && k > 0 && k % DG_BK == 0 && ntl >= 2 * (int64_t)pipe_wgs) {
        constexpr size_t smem = sizeof(double) * 2 * DG_BK * ((128 + 17) + (128 + 17));
        static bool attr_set = false;
        if (!attr_set) {
            JX_HIP(hipFuncSetAttribute((const void *)dsyr2k_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_set = true;
        }
        hipLaunchKernelGGL(dsyr2k_pipe_kernel, dim3((unsigned)pipe_wgs), dim3(512), smem, st, g);
        JX_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((unsigned)((int64_t)t * (t + 1) / 2), 1, 1);
    // the rank-2k update alone is faster in the general-loop form with two resident workgroups per CU (1.31 vs 1.50 ms at n_t =
    // 20000), the band reduction as a whole is faster with the one-workgroup stream form (397 vs 426 ms): the panel chain of
    // the NEXT panel runs on the second stream beside this kernel and is the critical path -- it gets the CU resources the
    // second workgroup would take.
    return dg_launch<128, 128>(g, grid, st);
}

}  // namespace jx

using namespace jx;

// C-ABI entry for tests and timing scripts: plain column-major dgemm on device pointers
namespace {
// workspace of the diagnostic entries (the eigensolver passes its own)
double *abi_workspace(size_t doubles) {
    static DevBuf buf;
    if (buf.bytes < doubles * sizeof(double) && buf.alloc(doubles * sizeof(double))) return nullptr;
    return buf.as<double>();
}
constexpr size_t kAbiWs = (size_t)8 << 20;   // 8 M doubles
}  // namespace

extern "C" int jxg_dgemm_f64(int ta, int tb, int m, int n, int k, double alpha, const double *d_a, int64_t lda,
                             const double *d_b, int64_t ldb, double beta, double *d_c, int64_t ldc, int ksplit,
                             void *stream) {
    if (m < 0 || n < 0 || k < 0) return fail("jxg_dgemm_f64: negative dimension");
    double *ws = abi_workspace(kAbiWs);
    if (!ws) return 1;
    return dgemm((hipStream_t)stream, ta != 0, tb != 0, m, n, k, alpha, d_a, lda, d_b, ldb, beta, d_c, ldc, ksplit, ws, kAbiWs);
}

extern "C" int jxg_dsymm_lower_f64(int m, int n, double alpha, const double *d_a, int64_t lda, const double *d_b,
                                   int64_t ldb, double beta, double *d_c, int64_t ldc, void *stream) {
    double *ws = abi_workspace(kAbiWs);
    if (!ws) return 1;
    return dsymm_lower((hipStream_t)stream, m, n, alpha, d_a, lda, d_b, ldb, beta, d_c, ldc, ws, kAbiWs);
}

extern "C" int jxg_dsyr2k_lower_nt_f64(int m, int k, double alpha, const double *d_a, int64_t lda, const double *d_b,
                                       int64_t ldb, double beta, double *d_c, int64_t ldc, void *stream) {
    return dsyr2k_lower_nt((hipStream_t)stream, m, k, alpha, d_a, lda, d_b, ldb, beta, d_c, ldc);
}
