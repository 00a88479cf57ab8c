// Householder tridiagonalisation (LAPACK dsytrd, lower, column-major output format) with TWO kernels per column.
//
// Replaces the panel factorisation of rocSOLVER's dsytrd, which issues five tiny latrd_* kernels per column
// (measured 155 ms of the 213 ms dsyevd at n = 5000; profiles/r01_bench_c2_lmm_kernel_stats.csv) behind
// src/math/eigh.rs:1422-1528 (reference: LAPACK dsyevd on the CPU).  Output is exactly LAPACK's: d, e, tau and the
// reflectors below the sub-diagonal of A, so rocSOLVER's dstedc + dormtr finish the eigendecomposition.
//
// Blocked algorithm (dlatrd): inside a panel of NB columns the trailing matrix is NOT updated; column j needs
//   (1) a_j  -= V W(j,:)' + W V(j,:)'          (2) larfg -> v, tau, beta
//   (3) y = T v (symv on the stale trailing matrix)   (4) w = tau (y - V (W'v) - W (V'v)) + alpha v
// and the panel ends with T -= V W' + W V' (rocBLAS dsyr2k).  Global dependencies per column are folded into
// two launches by moving every scalar reduction to where its inputs are already being streamed:
//   sytrd_symv_kernel  B(j): tiles of the lower trailing matrix -> y (f64 atomics), v'Tv; row chunks -> V'v, W'v,
//                            scaled reflector written to A / Vt; beta, tau derived in every block from the
//                            norm accumulated by the previous launch.
//   sytrd_update_kernel S(j): row-parallel w (alpha from v'Tv and (V'v).(W'v), no extra pass), then the
//                            update of column j+1 and its norm for the next reflector.
// Accumulators are double-buffered by column parity so no launch zeroes what a concurrent block still reads.
#include <hip/hip_ext.h>
#include <rocblas/rocblas.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "jx_common.h"

namespace jx {

int dsyr2k_lower_nt(hipStream_t st, int m, int k, double alpha, const double *a, int64_t lda, const double *b, int64_t ldb,
                    double beta, double *c, int64_t ldc);

constexpr int TD_NB = 64;     // panel width
constexpr int TD_TS = 64;     // symv tile
constexpr int TD_THREADS = 256;

struct TdAcc {            // one per column parity
    double *y;            // (TD_YC, n)  y = T v in TD_YC partial copies (copy = tile row block mod TD_YC)
    double *t1;           // (NB)  V'v
    double *t2;           // (NB)  W'v
    double *s0;           // v'Tv in TD_S0 partial sums, one per 128-byte line (slot = workgroup index mod TD_S0)
    double *s1;           // norm^2 of the next column below its first sub-diagonal entry, TD_S1 partial sums
};
// Scalars that every workgroup of a launch adds to are kept as a few partial sums on separate cache lines: atomics on
// one address retire one after the other (~50 ns each across XCDs), 660 strips adding to a single v'Tv word cost
// more than the 15 us launch they belong to.
constexpr int TD_S0 = 32, TD_S1 = 8, TD_SL = 16;   // slots, slots, doubles per slot (128 B)
constexpr int TD_YC = 4;                           // copies of y (the first column blocks take one atomic per tile row)

struct TdParams {
    double *a;            // (n,n) column-major, lower
    int64_t ld;
    int n;
    double *w;            // (n, NB) column-major panel W
    double *vt;           // (n, NB) row-major copy of the panel V (explicit unit entries)
    double *wt;           // (n, NB) row-major copy of W
    double *ubuf;         // (n) unscaled updated column j (rows j+1..)
    double *vbuf;         // (n) scaled reflector of column j
    double *d, *e, *tau;
    TdAcc acc[2];
    // node-level distribution (sytrd_set_dist): every rank keeps the whole matrix and runs every kernel, but a rank reads
    // only its 1 / world of the symv tiles.  The replicas must stay BIT-identical (a rank that updated its copy with a y
    // belonging to slightly different copies diverges by ~7x per column), so in this mode nothing replicated goes through
    // floating-point atomics: V'v / W'v are accumulated on rank 0 only and travel with the all-reduce, the norm of the next
    // column is summed in workgroup order by the last workgroup of S(j).
    int rank = 0, world = 1;
    double *s1part = nullptr;     // (chunks) per-workgroup partial norms of S(j)
    unsigned int *s1count = nullptr;
};

__device__ __forceinline__ double td_sum_s1(const double *s1) {
    double v[TD_S1];
#pragma unroll
    for (int k = 0; k < TD_S1; ++k) v[k] = s1[k * TD_SL];
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

__device__ __forceinline__ void larfg_scalars(double alpha0, double xnorm2, int nt, double &beta, double &tau,
                                              double &scale) {
    // LAPACK dlarfg without the safmin rescaling loop (inputs here are O(1) kinship entries)
    if (nt <= 1 || !(xnorm2 > 0.0)) {
        beta = alpha0;
        tau = 0.0;
        scale = 0.0;
    } else {
        const double nrm = sqrt(alpha0 * alpha0 + xnorm2);
        beta = (alpha0 >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha0) / beta;
        scale = 1.0 / (alpha0 - beta);
    }
}

// Panel start (column j = j0, no pending updates): d_j, unscaled column -> ubuf, norm^2 -> acc[par].sc[1].
template <bool DIST>
__global__ __launch_bounds__(TD_THREADS) void sytrd_panel_start_kernel(TdParams P, int j) {
    const int par = j & 1;
    const int r = j + 1 + blockIdx.x * TD_THREADS + threadIdx.x;
    double sq = 0.0;
    if (r < P.n) {
        const double u = P.a[r + (int64_t)j * P.ld];
        P.ubuf[r] = u;
        if (r >= j + 2) sq = u * u;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) P.d[j] = P.a[j + (int64_t)j * P.ld];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if (!DIST) {
        if ((threadIdx.x & 63) == 0 && sq != 0.0)
            unsafeAtomicAdd(&P.acc[par].s1[((blockIdx.x * 4 + (threadIdx.x >> 6)) & (TD_S1 - 1)) * TD_SL], sq);
        return;
    }
    // distributed form: bit-reproducible norm (per-wave partials in order, summed by the last workgroup)
    if ((threadIdx.x & 63) == 0) P.s1part[blockIdx.x * 4 + (threadIdx.x >> 6)] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int done = atomicAdd(P.s1count, 1u);
        if (done == gridDim.x - 1) {
            __threadfence();
            double acc_n = 0.0;
            for (unsigned int b = 0; b < gridDim.x * 4; ++b) acc_n += __builtin_nontemporal_load(&P.s1part[b]);
            P.acc[par].s1[0] = acc_n;
            *P.s1count = 0u;
        }
    }
}

// Row-chunk part of B(j) (64 rows per workgroup): scaled reflector out (A, vbuf, Vt), V'v and W'v partial sums,
// e_j / tau_j.  `scratch` >= 64 + 2 * 4 * 64 doubles of LDS.
template <bool DIST>
__device__ __forceinline__ void sytrd_chunk_path(const TdParams &P, int j, int j0, int chunk, double *scratch) {
    const int par = j & 1;
    const int n = P.n;
    const int nt = n - j - 1;
    const int base = j + 1;
    const int i = j - j0;
    const int tid = threadIdx.x;
    double beta, tau, scale;
    larfg_scalars(P.ubuf[base], td_sum_s1(P.acc[par].s1), nt, beta, tau, scale);

    // ---- row chunk (64 rows): scaled reflector out, V'v and W'v partial sums -----------------------------
    const int rrow0 = chunk * TD_TS;  // relative row of this block's first row
    // the row-chunk path never touches the tile image: reuse its storage (keeps the kernel at 4 workgroups per CU)
    double *vsh = scratch;
    double (*part)[4][TD_NB] = reinterpret_cast<double (*)[4][TD_NB]>(scratch + TD_TS);
    if (tid < TD_TS) {
        const int rr = rrow0 + tid;
        double v = 0.0;
        if (rr < nt) {
            v = (rr == 0) ? 1.0 : P.ubuf[base + rr] * scale;
            P.a[(base + rr) + (int64_t)j * P.ld] = v;  // explicit reflector (unit entry included) while in the panel
            P.vbuf[base + rr] = v;
            P.vt[(int64_t)(base + rr) * TD_NB + i] = v;
        }
        vsh[tid] = v;
    }
    if (chunk == 0 && tid == 0) {
        P.e[j] = beta;
        P.tau[j] = tau;
    }
    if (chunk == 0 && tid < TD_S1) P.acc[par ^ 1].s1[tid * TD_SL] = 0.0;   // S(j) accumulates the next norm there
    if (i > 0) {
        __syncthreads();
        // t1[k] += sum_r Vt[r][k] v_r ; t2[k] += sum_r Wt[r][k] v_r : thread (k = tid & 63, sub = tid >> 6), 16 rows
        const int k = tid & 63, sub = tid >> 6;
        double s1 = 0.0, s2 = 0.0;
        if (k < i) {
            // two batches of 8 rows: 16 loads in flight per batch, half the registers of one 16-row batch
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                double vv[8], a1[8], a2[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int rloc = sub * 16 + hb * 8 + q;
                    const bool okr = (rrow0 + rloc) < nt;
                    const int64_t row = (int64_t)(base + rrow0 + (okr ? rloc : 0)) * TD_NB;
                    vv[q] = okr ? vsh[rloc] : 0.0;
                    a1[q] = P.vt[row + k];
                    a2[q] = P.wt[row + k];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    s1 += a1[q] * vv[q];
                    s2 += a2[q] * vv[q];
                }
            }
        }
        part[0][sub][k] = s1;
        part[1][sub][k] = s2;
        __syncthreads();
        if (tid < i && (!DIST || P.rank == 0)) {   // distributed: rank 0's sums reach the others by the all-reduce
            const double a1 = part[0][0][tid] + part[0][1][tid] + part[0][2][tid] + part[0][3][tid];
            const double a2 = part[1][0][tid] + part[1][1][tid] + part[1][2][tid] + part[1][3][tid];
            if (a1 != 0.0) unsafeAtomicAdd(&P.acc[par].t1[tid], a1);
            if (a2 != 0.0) unsafeAtomicAdd(&P.acc[par].t2[tid], a2);
        }
    }
}

// B(j): see file header.  grid = nstrips + row chunks.
// Symv work unit ("strip"): one 64-row block x up to K consecutive 64-column tiles of the lower triangle.  The next
// tile's 8 x 16-byte loads per thread are issued before the current tile is reduced out of LDS, so every
// workgroup keeps 32 KB in flight; the row product accumulates in registers across the strip (one atomic per row
// per strip), the mirrored column product is reduced per tile.
template <bool DIST>
__global__ __launch_bounds__(TD_THREADS, 4) void sytrd_symv_kernel(TdParams P, int j, int j0, int ktiles,
                                                                int nstrips) {
    __shared__ __attribute__((aligned(16))) double tile[TD_TS][TD_TS + 2];  // tile[c][r], pitch 66 (16-B rows)
    __shared__ double vr[TD_TS], vc[TD_TS];
    __shared__ double red[4][TD_TS];
    __shared__ double red1[TD_THREADS / 64];
    const int par = j & 1;
    const int n = P.n;
    const int nt = n - j - 1;       // trailing dimension
    const int base = j + 1;         // first trailing row/col
    const int tid = threadIdx.x;

    if ((int)blockIdx.x < nstrips) {
        // distributed form: strips are dealt round-robin to the ranks; one all-reduce between this launch and S(j) sums
        // the ranks' partial y / v'Tv
        if (DIST && ((int)blockIdx.x % P.world) != P.rank) return;
        // ---- locate the strip: row block rb (rb+1 tiles), first tile cb0 ---------------------------------
        // uniform decomposition: nsx strips per row block, strips entirely above the diagonal exit at once
        const int side = (nt + TD_TS - 1) / TD_TS;
        const int nsx = (side + ktiles - 1) / ktiles;
        const int rb = blockIdx.x / nsx;
        const int cb0 = (blockIdx.x - rb * nsx) * ktiles;
        if (cb0 > rb) return;
        const int cb1 = (cb0 + ktiles < rb + 1) ? (cb0 + ktiles) : (rb + 1);
        const int r0 = rb * TD_TS;
        const int rp = tid & 31, cg = tid >> 5;
        const int rr0 = r0 + 2 * rp;
        double2 tv[8];
        auto load_tile = [&](int cb) {
            const int c0 = cb * TD_TS;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = c0 + cg + 8 * q;
                double2 val = make_double2(0.0, 0.0);
                if (c < nt) {
                    const double *src = P.a + (base + rr0) + (int64_t)(base + c) * P.ld;
                    if (rr0 + 1 < nt) {
                        val = *reinterpret_cast<const double2 *>(src);
                    } else if (rr0 < nt) {
                        val.x = src[0];
                    }
                }
                tv[q] = val;
            }
        };
        load_tile(cb0);
        // raw reflector pieces: rows of this strip (threads 0..63) and columns of the first tile (64..127)
        double raw_r = 0.0, raw_c = 0.0;
        if (tid < TD_TS) {
            const int vi = r0 + tid;
            if (vi < nt) raw_r = P.ubuf[base + vi];
        } else if (tid < 2 * TD_TS) {
            const int vi = cb0 * TD_TS + tid - TD_TS;
            if (vi < nt) raw_c = P.ubuf[base + vi];
        }
        double beta, tau, scale;
        larfg_scalars(P.ubuf[base], td_sum_s1(P.acc[par].s1), nt, beta, tau, scale);
        if (tid < TD_TS) {
            const int vi = r0 + tid;
            vr[tid] = (vi < nt) ? (vi == 0 ? 1.0 : raw_r * scale) : 0.0;
        }
        const int tr = tid & 63, tg = tid >> 6;
        const int qc = tid >> 2, qs = tid & 3;
        double p = 0.0, contrib = 0.0;
        for (int cb = cb0; cb < cb1; ++cb) {
            const int c0 = cb * TD_TS;
            const bool diag = (cb == rb);
            if (tid >= TD_TS && tid < 2 * TD_TS) {
                const int vi = c0 + tid - TD_TS;
                vc[tid - TD_TS] = (vi < nt) ? (vi == 0 ? 1.0 : raw_c * scale) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int cc = cg + 8 * q;
                double2 val = tv[q];
                if (diag) {  // keep the lower triangle (incl. diagonal) of a diagonal tile
                    if (2 * rp < cc) val.x = 0.0;
                    if (2 * rp + 1 < cc) val.y = 0.0;
                }
                *reinterpret_cast<double2 *>(&tile[cc][2 * rp]) = val;
            }
            __syncthreads();
            if (cb + 1 < cb1) {  // next tile and its reflector piece in flight while this tile is reduced
                load_tile(cb + 1);
                if (tid >= TD_TS && tid < 2 * TD_TS) {
                    const int vi = (cb + 1) * TD_TS + tid - TD_TS;
                    raw_c = (vi < nt) ? P.ubuf[base + vi] : 0.0;
                }
            }
#pragma unroll 4
            for (int q4 = 0; q4 < TD_TS / 4; ++q4) p += tile[tg + 4 * q4][tr] * vc[tg + 4 * q4];
            double q = 0.0;
#pragma unroll 4
            for (int q4 = 0; q4 < TD_TS / 4; ++q4) {
                const int rr = qs + 4 * q4;
                const double val = tile[qc][rr];
                if (!diag || rr != qc) q += val * vr[rr];
            }
            q += __shfl_xor(q, 1, 64);
            q += __shfl_xor(q, 2, 64);
            if (qs == 0) {
                if (c0 + qc < nt && q != 0.0)
                    unsafeAtomicAdd(&P.acc[par].y[(int64_t)(rb & (TD_YC - 1)) * n + base + c0 + qc], q);
                contrib += vc[qc] * q;
            }
            __syncthreads();
        }
        red[tg][tr] = p;
        __syncthreads();
        if (tid < TD_TS) {
            const double pr = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            if (r0 + tid < nt && pr != 0.0)
                unsafeAtomicAdd(&P.acc[par].y[(int64_t)(rb & (TD_YC - 1)) * n + base + r0 + tid], pr);
            contrib += vr[tid] * pr;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) contrib += __shfl_down(contrib, off, 64);
        if ((tid & 63) == 0) red1[tid >> 6] = contrib;
        __syncthreads();
        if (tid == 0) {
            const double tot = red1[0] + red1[1] + red1[2] + red1[3];
            if (tot != 0.0) unsafeAtomicAdd(&P.acc[par].s0[(blockIdx.x & (TD_S0 - 1)) * TD_SL], tot);
        }
        return;
    }
    sytrd_chunk_path<DIST>(P, j, j0, blockIdx.x - nstrips, &tile[0][0]);
}

// S(j): see file header. Four threads per trailing row (16 panel columns each), 64 rows per block.
template <bool DIST>
__global__ __launch_bounds__(TD_THREADS) void sytrd_update_kernel(TdParams P, int j, int j0, int do_next) {
    __shared__ double t1s[TD_NB], t2s[TD_NB], vj1[TD_NB], wj1k[TD_NB];
    __shared__ double scal[2];
    __shared__ double redn[TD_THREADS / 64];
    const int par = j & 1;
    const int n = P.n;
    const int base = j + 1;
    const int nt = n - j - 1;
    const int i = j - j0;
    const int tid = threadIdx.x;
    const double tau = P.tau[j];
    // every global read of this launch is issued up front (the scalar chain below is otherwise three dependent
    // round trips): v'Tv, y at the first trailing row, then the row-side operands
    const double yv_part = (tid < TD_S0) ? P.acc[par].s0[tid * TD_SL] : 0.0;
    double y_first = 0.0;
#pragma unroll
    for (int c = 0; c < TD_YC; ++c) y_first += P.acc[par].y[(int64_t)c * n + base];
    double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
    if (tid < i) {  // i <= 63
        a1 = P.acc[par].t1[tid];
        a2 = P.acc[par].t2[tid];
        b1 = P.vt[(int64_t)base * TD_NB + tid];
        b2 = P.wt[(int64_t)base * TD_NB + tid];
    }
    const int rloc = tid >> 2, q = tid & 3;
    const int rr = blockIdx.x * TD_TS + rloc;  // relative trailing row
    const bool okr = rr < nt;
    const int r = base + (okr ? rr : 0);
    double2 vv[8], ww[8];
    {
        const double2 *vrow = reinterpret_cast<const double2 *>(P.vt + (int64_t)r * TD_NB + q * 16);
        const double2 *wrow = reinterpret_cast<const double2 *>(P.wt + (int64_t)r * TD_NB + q * 16);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            vv[h] = vrow[h];
            ww[h] = wrow[h];
        }
    }
    const double v_r = P.vbuf[r];
    double y_r = 0.0;
#pragma unroll
    for (int c = 0; c < TD_YC; ++c) y_r += P.acc[par].y[(int64_t)c * n + r];
    const double a_next = do_next ? P.a[r + (int64_t)(j + 1) * P.ld] : 0.0;
    if (tid < TD_NB) {
        t1s[tid] = a1;
        t2s[tid] = a2;
        vj1[tid] = b1;
        wj1k[tid] = b2;
        double dot = a1 * a2;
        double s1 = b1 * a2 + b2 * a1;
        double yv_all = yv_part;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            dot += __shfl_xor(dot, off, 64);
            s1 += __shfl_xor(s1, off, 64);
            yv_all += __shfl_xor(yv_all, off, 64);
        }
        if (tid == 0) {
            const double alpha = -0.5 * tau * tau * (yv_all - 2.0 * dot);
            // w at the first trailing row (v = 1 there): needed by every row for the next-column update
            scal[0] = alpha;
            scal[1] = tau * (y_first - s1) + alpha;
        }
    }
    __syncthreads();
    const double alpha = scal[0], w_first = scal[1];
    double s1 = 0.0, s2 = 0.0;
    {
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int k0 = q * 16 + 2 * h;
            if (k0 < i) {
                s1 += vv[h].x * t2s[k0] + ww[h].x * t1s[k0];
                s2 += vv[h].x * wj1k[k0] + ww[h].x * vj1[k0];
            }
            if (k0 + 1 < i) {
                s1 += vv[h].y * t2s[k0 + 1] + ww[h].y * t1s[k0 + 1];
                s2 += vv[h].y * wj1k[k0 + 1] + ww[h].y * vj1[k0 + 1];
            }
        }
    }
    s1 += __shfl_xor(s1, 1, 64);
    s1 += __shfl_xor(s1, 2, 64);
    s2 += __shfl_xor(s2, 1, 64);
    s2 += __shfl_xor(s2, 2, 64);
    double sq = 0.0;
    if (okr && q == 0) {
        const double v = v_r;
        const double w = tau * (y_r - s1) + alpha * v;
        P.w[r + (int64_t)i * n] = w;
        P.wt[(int64_t)r * TD_NB + i] = w;
        if (do_next) {
            // column j+1 with the i+1 pending reflector pairs applied (rows >= j+1)
            const double unew = a_next - s2 - (v * w_first + w);
            if (rr == 0) {
                P.d[j + 1] = unew;
                P.a[r + (int64_t)(j + 1) * P.ld] = unew;
            } else {
                P.ubuf[r] = unew;
                if (rr >= 2) sq = unew * unew;
            }
        }
#pragma unroll
        for (int c = 0; c < TD_YC; ++c) P.acc[par ^ 1].y[(int64_t)c * n + r] = 0.0;
    }
    if (blockIdx.x == 0) {
        if (tid < TD_NB) {
            P.acc[par ^ 1].t1[tid] = 0.0;
            P.acc[par ^ 1].t2[tid] = 0.0;
        }
        if (tid < TD_S0) P.acc[par ^ 1].s0[tid * TD_SL] = 0.0;
    }
    if (do_next) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
        if ((tid & 63) == 0) redn[tid >> 6] = sq;
        __syncthreads();
        if (tid == 0) {
            const double tot = redn[0] + redn[1] + redn[2] + redn[3];
            if (!DIST) {
                if (tot != 0.0) unsafeAtomicAdd(&P.acc[par ^ 1].s1[(blockIdx.x & (TD_S1 - 1)) * TD_SL], tot);
            } else {
                // bit-reproducible form: partials in workgroup order, summed by whichever workgroup arrives last
                P.s1part[blockIdx.x] = tot;
                __threadfence();
                const unsigned int done = atomicAdd(P.s1count, 1u);
                if (done == gridDim.x - 1) {
                    __threadfence();
                    double acc_n = 0.0;
                    for (unsigned int b = 0; b < gridDim.x; ++b) acc_n += __builtin_nontemporal_load(&P.s1part[b]);
                    P.acc[par ^ 1].s1[0] = acc_n;   // the other slots stay zero (cleared by B(j))
                    *P.s1count = 0u;
                }
            }
        }
    }
}

// after the last panel: restore e on the sub-diagonal (LAPACK layout) and read the last diagonal entry
__global__ void sytrd_finish_kernel(TdParams P) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P.n - 1) P.a[(j + 1) + (int64_t)j * P.ld] = P.e[j];
    if (j == P.n - 1) P.d[j] = P.a[j + (int64_t)j * P.ld];
}

// LDS-resident tail: the last t <= TD_TAILN columns by ONE workgroup (a column of the two-kernel form costs two launch
// floors, ~17 us, however small the trailing matrix is; here it is ~1-2 us).  Unblocked dsytd2 (lower) on the packed
// lower triangle in LDS: per column larfg, y = tau S v, alpha = -tau/2 y'v, w = y + alpha v, S -= v w' + w v'.
// Element (r, c), r >= c, of the t x t block lives at tp_off(c) + r - c.  Writes d, e, tau and the reflectors like B(j).
constexpr int TD_TAILN = 192;
constexpr int TD_TAIL_THREADS = 1024;   // 16 waves: four per SIMD to cover the LDS latency of the column loops
constexpr int TD_TAIL_ROWS = 256;       // threads (r, h): row r = tid % 256 (< m <= 191 active), column class h = tid / 256

__global__ __launch_bounds__(TD_TAIL_THREADS) void sytrd_tail_kernel(TdParams P, int j0, long long *dbg) {
    extern __shared__ double tail_lds[];
    long long tk[6] = {0, 0, 0, 0, 0, 0};
    long long tc = dbg ? wall_clock64() : 0;
#define TAIL_MARK(i) if (dbg) { const long long now = wall_clock64(); tk[i] += now - tc; tc = now; }
    const int n = P.n, t = n - j0, tid = threadIdx.x;
    const int r = tid & (TD_TAIL_ROWS - 1), h = tid >> 8;   // h in 0..3
    constexpr int NW = TD_TAIL_THREADS / 64;
    double *Tp = tail_lds;                         // t (t + 1) / 2
    double *v = Tp + (t * (t + 1)) / 2;            // t
    double *w = v + t;                             // t
    double *red_a = w + t;                         // NW: partial v'Sv
    double *red_b = red_a + NW;                    // NW: partial norm of the next column
    double *e_s = red_b + NW;                      // t: e and tau stay in LDS until the end
    double *tau_s = e_s + t;                       // t
    double *part = tau_s + t;                      // 4 x t partial row products
    auto tp_off = [t](int c) { return c * t - (c * (c - 1)) / 2; };
    // 4 columns per pass, one element per thread and column: the loads of a pass are in flight together
    for (int c0 = 0; c0 < t; c0 += 16) {
        double val[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + 4 * q + h, rr = c + r;
            val[q] = (c < t && rr < t) ? P.a[j0 + rr + (int64_t)(j0 + c) * P.ld] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + 4 * q + h, rr = c + r;
            if (c < t && rr < t) Tp[tp_off(c) + r] = val[q];
        }
    }
    __syncthreads();
    // norm^2 of the first column below its sub-diagonal entry
    double xn2;
    {
        double sq = 0.0;
        if (h == 0 && r >= 2 && r < t) sq = Tp[r] * Tp[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        if ((tid & 63) == 0) red_b[tid >> 6] = sq;
        __syncthreads();
        xn2 = 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) xn2 += red_b[q];
    }
    TAIL_MARK(0)
    // Three barriers per column: (B) partial row products and v'Sv out, (C) w out, (D) trailing block updated and the
    // next column's norm out.  v is formed on the fly from the raw column (x * scale), every thread redoes the scalars.
    for (int k = 0; k + 1 < t; ++k) {
        const int m = t - k - 1;                   // trailing dimension, local rows / columns 0 .. m-1
        const int ok = tp_off(k);
        const double *xcol = Tp + ok + 1;          // raw column: x[0] = sub-diagonal entry
        double beta, tau, scale;
        larfg_scalars(xcol[0], xn2, m, beta, tau, scale);
        TAIL_MARK(1)
        double pd = 0.0;
        if (r < m) {
            // y = S v: thread (r, h) takes the columns c = h (mod 4).  Element (r, c), c <= r, sits at
            // tp_off(k+1) + r + c (m - 1) - c (c - 1) / 2 (lanes = consecutive r: conflict-free); stepping c by 4 moves it
            // by 4 (m - 1) - 4 c - 6 doubles and that step shrinks by 16 (no integer multiplies in the loops).
            // For c > r it is the mirrored entry in row r's own column.
            int addr = tp_off(k + 1) + r + h * (m - 1) - (h * (h - 1)) / 2;
            int step = 4 * (m - 1) - 4 * h - 6;
            double a0 = 0.0, a1 = 0.0;
            int c = h;
            if (c == 0 && c <= r) {                // v[0] = 1
                a0 = Tp[addr];
                addr += step;
                step -= 16;
                c = 4;
            }
            for (; c + 4 <= r; c += 8) {
                const int addr2 = addr + step;
                a0 += Tp[addr] * (xcol[c] * scale);
                a1 += Tp[addr2] * (xcol[c + 4] * scale);
                addr = addr2 + step - 16;
                step -= 32;
            }
            for (; c <= r; c += 4) {
                a0 += Tp[addr] * (xcol[c] * scale);
                addr += step;
                step -= 16;
            }
            const double *colp = Tp + tp_off(k + 1 + r) - r;    // element (c, r), c > r, at colp[c]
            for (; c + 4 < m; c += 8) {            // c continues in the same residue class past the diagonal
                a0 += colp[c] * (xcol[c] * scale);
                a1 += colp[c + 4] * (xcol[c + 4] * scale);
            }
            for (; c < m; c += 4) a0 += colp[c] * (xcol[c] * scale);
            const double a = a0 + a1;
            const double vr = (r == 0) ? 1.0 : xcol[r] * scale;
            part[h * t + r] = a;
            pd = a * vr;
            if (h == 0) v[r] = vr;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pd += __shfl_xor(pd, off, 64);
        if ((tid & 63) == 0) red_a[tid >> 6] = pd;
        if (tid == 0) {
            e_s[k] = beta;
            tau_s[k] = tau;
        }
        __syncthreads();                           // (B)
        TAIL_MARK(2)
        double vsv = 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) vsv += red_a[q];
        const double alpha = -0.5 * tau * (tau * vsv);          // -tau/2 y'v with y = tau S v
        if (h == 0 && r < m) {
            const double yr = tau * ((part[r] + part[t + r]) + (part[2 * t + r] + part[3 * t + r]));
            w[r] = yr + alpha * v[r];
        }
        __syncthreads();                           // (C)
        TAIL_MARK(3)
        double sq = 0.0;
        if (r < m) {
            const double vr = v[r], wr = w[r];
            int addr = tp_off(k + 1) + r + h * (m - 1) - (h * (h - 1)) / 2;
            int step = 4 * (m - 1) - 4 * h - 6;
            int c = h;
            if (h == 0) {                          // column 0 of the trailing block = the next column to reduce
                const double nv = Tp[addr] - (vr * w[0] + wr * v[0]);
                Tp[addr] = nv;
                if (r >= 2) sq = nv * nv;
                addr += step;
                step -= 16;
                c = 4;
            }
            for (; c + 4 <= r; c += 8) {
                const int addr2 = addr + step;
                const double t0 = Tp[addr], t1 = Tp[addr2];
                Tp[addr] = t0 - (vr * w[c] + wr * v[c]);
                Tp[addr2] = t1 - (vr * w[c + 4] + wr * v[c + 4]);
                addr = addr2 + step - 16;
                step -= 32;
            }
            for (; c <= r; c += 4) {
                Tp[addr] -= vr * w[c] + wr * v[c];
                addr += step;
                step -= 16;
            }
            if (h == 1) Tp[ok + 1 + r] = vr;       // reflector in place (unit entry included; e restored by the finish kernel)
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        if ((tid & 63) == 0) red_b[tid >> 6] = sq;
        __syncthreads();                           // (D)
        xn2 = 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) xn2 += red_b[q];
        TAIL_MARK(4)
    }
    // results out: reflectors (columns of the packed image, diagonal included: A(n-1, n-1) feeds the finish kernel), d, e, tau
    for (int c0 = 0; c0 < t; c0 += 4) {
        const int c = c0 + h, rr = c + r;
        if (c < t && rr < t) P.a[j0 + rr + (int64_t)(j0 + c) * P.ld] = Tp[tp_off(c) + r];
    }
    for (int k = tid; k < t; k += TD_TAIL_THREADS) {
        P.d[j0 + k] = Tp[tp_off(k)];
        if (k + 1 < t) {
            P.e[j0 + k] = e_s[k];
            P.tau[j0 + k] = tau_s[k];
        }
    }
    TAIL_MARK(5)
    if (dbg && tid == 0)
        for (int i = 0; i < 6; ++i) dbg[i] = tk[i];
#undef TAIL_MARK
}

// distributed form: the per-column collective carries [sum of the TD_YC y copies (n) | s0 | t1 | t2]; pack folds the
// copies in a fixed order, unpack puts the reduced vector into copy 0 and clears the others
constexpr int TD_TAIL = TD_S0 * TD_SL + 2 * TD_NB;
__global__ void sytrd_dist_pack_kernel(const double *__restrict__ reg, int n, double *__restrict__ staging) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double v = reg[i];
#pragma unroll
        for (int c = 1; c < TD_YC; ++c) v += reg[(int64_t)c * n + i];
        staging[i] = v;
    } else if (i < n + TD_TAIL) {
        staging[i] = reg[(int64_t)TD_YC * n + (i - n)];
    }
}
__global__ void sytrd_dist_unpack_kernel(double *__restrict__ reg, int n, const double *__restrict__ staging) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        reg[i] = staging[i];
#pragma unroll
        for (int c = 1; c < TD_YC; ++c) reg[(int64_t)c * n + i] = 0.0;
    } else if (i < n + TD_TAIL) {
        reg[(int64_t)TD_YC * n + (i - n)] = staging[i];
    }
}

struct SytrdDist {
    int rank = 0, world = 1;
    int (*allreduce)(void *user) = nullptr;   // sums `staging_doubles` doubles at `staging` over the ranks, on the stream
    void *user = nullptr;
    double *staging = nullptr;
    int64_t staging_doubles = 0;
    int min_n = 16384;
};
static SytrdDist g_dist;
static bool g_dist_paused = false;   // jxg_eigh_set_local: this rank decomposes a matrix of its own (no collective) for a while

extern float g_last_ms[24];   // [2] mean duration (ms) of the sampled symv launches, [3] their mean algorithmic MB

// d_a: (n,n) column-major symmetric (lower referenced), overwritten with the LAPACK dsytrd(lower) result.
int sytrd_lower(rocblas_handle h, hipStream_t st, double *d_a, int n, double *d_d, double *d_e, double *d_tau) {
    if (n < 2) {
        if (n == 1) JX_HIP(hipMemcpyAsync(d_d, d_a, sizeof(double), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    ScratchLease work;
    const size_t nn = (size_t)n;
    // w (n*NB) + vt (n*NB) + wt (n*NB) + ubuf (n) + vbuf (n) + 2 * (y (n) + t1 (NB) + t2 (NB) + sc (2))
    const size_t nchunk_max = (nn + TD_TS - 1) / TD_TS;
    const size_t doubles = 3 * nn * TD_NB + 2 * nn + 2 * (TD_YC * nn + 2 * TD_NB + (TD_S0 + TD_S1) * TD_SL) +
                           (nchunk_max + 4) + 8;
    if (work.take(2, sizeof(double) * doubles)) return 1;
    double *p = work.as<double>();
    TdParams P;
    P.a = d_a;
    P.ld = n;
    P.n = n;
    P.w = p; p += nn * TD_NB;
    P.vt = p; p += nn * TD_NB;
    P.wt = p; p += nn * TD_NB;
    P.ubuf = p; p += nn;
    P.vbuf = p; p += nn;
    double *acc_begin = p;
    for (int q = 0; q < 2; ++q) {
        P.acc[q].y = p; p += TD_YC * nn;           // y, s0, t1, t2 adjacent: one all-reduce in the distributed form
        P.acc[q].s0 = p; p += TD_S0 * TD_SL;
        P.acc[q].t1 = p; p += TD_NB;
        P.acc[q].t2 = p; p += TD_NB;
        P.acc[q].s1 = p; p += TD_S1 * TD_SL;
    }
    const size_t acc_bytes = sizeof(double) * (size_t)(p - acc_begin);
    P.s1part = p; p += nchunk_max + 4;   // panel start: one partial per wave of ceil(rows / 256) workgroups
    P.s1count = reinterpret_cast<unsigned int *>(p); p += 8;
    P.d = d_d;
    P.e = d_e;
    P.tau = d_tau;
    const size_t red_doubles = nn + (size_t)TD_TAIL;   // [folded y | s0 | t1 | t2]
    // JXGPU_DIST_EIGH_FORCE: run the distributed instantiation and the collective with a single rank too (how the RCCL
    // callback path is exercised on a one-GPU box)
    static const bool force_single = getenv("JXGPU_DIST_EIGH_FORCE") && atoi(getenv("JXGPU_DIST_EIGH_FORCE")) != 0;
    const bool dist_on = !g_dist_paused && (g_dist.world > 1 || force_single) && g_dist.allreduce && n >= g_dist.min_n &&
                         g_dist.staging_doubles >= (int64_t)red_doubles;
    if (dist_on) {
        JX_HIP(hipMemsetAsync(P.s1count, 0, 8 * sizeof(double), st));
        P.rank = g_dist.rank;
        P.world = g_dist.world;
    }
    // the single-rank instantiations carry none of the distributed form's branches
    const auto k_start = dist_on ? sytrd_panel_start_kernel<true> : sytrd_panel_start_kernel<false>;
    const auto k_symv = dist_on ? sytrd_symv_kernel<true> : sytrd_symv_kernel<false>;
    const auto k_update = dist_on ? sytrd_update_kernel<true> : sytrd_update_kernel<false>;
    const double minus1 = -1.0, one = 1.0;
    // live roofline sample: the symv launch in the middle of every panel is bracketed by HIP events
    // (one pair per panel = every 64th launch, uniformly spaced over the trailing sizes)
    static std::vector<hipEvent_t> ev;
    const int npanels = (n - 1 + TD_NB - 1) / TD_NB;
    // + 8 events for four empty back-to-back pairs (reported under JXGPU_EIGH_TRACE: what a hipEventRecord bracket
    // would add to a sample; the samples themselves use hipExtLaunchKernelGGL's dispatch-bound events)
    while ((int)ev.size() < 2 * npanels + 8) {
        hipEvent_t e;
        JX_HIP(hipEventCreate(&e));
        ev.push_back(e);
    }
    int nsamp = 0;
    double samp_bytes = 0.0;
    // the last tail_n columns go to the LDS-resident single-workgroup kernel; the first panel is narrower so that the
    // 64-column panels end exactly where the tail starts (JXGPU_SYTRD_TAIL=0: two-kernel form to the end)
    static const int tail_env = getenv("JXGPU_SYTRD_TAIL") ? atoi(getenv("JXGPU_SYTRD_TAIL")) : TD_TAILN;
    const int tail_cap = tail_env < 0 ? 0 : (tail_env > TD_TAILN ? TD_TAILN : tail_env);
    const int tail_n = (tail_cap >= 2) ? (n < tail_cap ? n : tail_cap) : 0;
    const int n_panel_cols = tail_n ? (n - tail_n) : (n - 1);   // columns reduced by the panel loop
    int first_pw = n_panel_cols % TD_NB;
    if (first_pw == 0) first_pw = TD_NB;
    for (int j0 = 0, pw = 0; j0 < n_panel_cols; j0 += pw) {
        pw = (j0 == 0) ? first_pw : TD_NB;
        if (pw > n_panel_cols - j0) pw = n_panel_cols - j0;
        JX_HIP(hipMemsetAsync(acc_begin, 0, acc_bytes, st));
        {
            const int rows = n - j0 - 1;
            hipLaunchKernelGGL(k_start, dim3((rows + TD_THREADS - 1) / TD_THREADS), dim3(TD_THREADS), 0, st, P, j0);
        }
        for (int i = 0; i < pw; ++i) {
            const int j = j0 + i;
            const int nt = n - j - 1;
            const int nchunks = (nt + TD_TS - 1) / TD_TS;
            static const int kt_env = getenv("JXGPU_SYTRD_KT") ? atoi(getenv("JXGPU_SYTRD_KT")) : 0;
            static const int tg_env = getenv("JXGPU_SYTRD_TARGET") ? atoi(getenv("JXGPU_SYTRD_TARGET")) : 768;
            // strips: one 64-row block x K consecutive tiles; K grows with the tile count so that ~768
            // workgroups stay in flight.  (A 256x16 tall-tile form with 2 KB column runs measured 3-5 % slower.)
            const int side = (nt + TD_TS - 1) / TD_TS;
            const int64_t ntiles = (int64_t)side * (side + 1) / 2;
            int ktiles = kt_env > 0 ? kt_env : (int)((ntiles + tg_env - 1) / tg_env);
            if (ktiles < 1) ktiles = 1;
            if (ktiles > 64) ktiles = 64;
            const int nstrips = side * ((side + ktiles - 1) / ktiles);
            const bool sample = (i == pw / 2);
            if (sample) {
                // start / stop events bound to this dispatch's own begin / end time stamps (what rocprofv3 reports)
                hipExtLaunchKernelGGL(k_symv, dim3(nstrips + nchunks), dim3(TD_THREADS), 0, st,
                                      ev[2 * nsamp], ev[2 * nsamp + 1], 0, P, j, j0, ktiles, nstrips);
            } else {
                hipLaunchKernelGGL(k_symv, dim3(nstrips + nchunks), dim3(TD_THREADS), 0, st, P, j, j0,
                                   ktiles, nstrips);
            }
            if (sample) {
                // lower triangle incl. diagonal, f64 (this rank's share of it in the distributed form)
                samp_bytes += (4.0 * (double)nt * (double)nt + 4.0 * (double)nt) / (double)P.world;
                ++nsamp;
            }
            if (dist_on) {   // sum the ranks' partial y / v'Tv (+ rank 0's V'v, W'v): staging copy, collective, copy back
                double *reg = P.acc[j & 1].y;
                const int pk_grid = (int)((red_doubles + 255) / 256);
                hipLaunchKernelGGL(sytrd_dist_pack_kernel, dim3(pk_grid), dim3(256), 0, st, reg, n, g_dist.staging);
                if (g_dist.allreduce(g_dist.user)) return fail("sytrd: the all-reduce callback failed");
                hipLaunchKernelGGL(sytrd_dist_unpack_kernel, dim3(pk_grid), dim3(256), 0, st, reg, n, g_dist.staging);
            }
            hipLaunchKernelGGL(k_update, dim3(nchunks), dim3(TD_THREADS), 0, st, P, j, j0,
                               (i + 1 < pw) ? 1 : 0);
        }
        JX_LAUNCH_CHECK();
        if (getenv("JXGPU_EIGH_TRACE") && (j0 % (64 * TD_NB)) == 0) {
            JX_HIP(hipStreamSynchronize(st));
            fprintf(stderr, "[jxgpu sytrd n=%d] panel at column %d reduced\n", n, j0);
            fflush(stderr);
        }
        const int j1 = j0 + pw;      // first row/col of the trailing matrix after this panel
        const int n2 = n - j1;
        if (n2 > 0) {
            // T -= V W' + W V'  (lower), V = A(j1:n, j0:j1), W = w(j1:n, 0:pw)
            // own f64 MFMA GEMM on the lower tiles (k_dgemm.hip), the two halves of the rank-2k update in turn; JXGPU_SYTRD_SYR2K=
            // rocblas keeps the vendor routine for A/B runs
            static const bool vendor = getenv("JXGPU_SYTRD_SYR2K") && strcmp(getenv("JXGPU_SYTRD_SYR2K"), "rocblas") == 0;
            if (vendor) {
                rocblas_status rs = rocblas_dsyr2k(h, rocblas_fill_lower, rocblas_operation_none, n2, pw, &minus1,
                                                   d_a + j1 + (int64_t)j0 * n, n, P.w + j1, n, &one,
                                                   d_a + j1 + (int64_t)j1 * n, n);
                if (rs != rocblas_status_success) return fail("rocblas_dsyr2k failed: " + std::to_string((int)rs));
            } else {
                if (dsyr2k_lower_nt(st, n2, pw, -1.0, d_a + j1 + (int64_t)j0 * n, n, P.w + j1, n, 1.0, d_a + j1 + (int64_t)j1 * n, n))
                    return 1;
                if (dsyr2k_lower_nt(st, n2, pw, -1.0, P.w + j1, n, d_a + j1 + (int64_t)j0 * n, n, 1.0, d_a + j1 + (int64_t)j1 * n, n))
                    return 1;
            }
            if (getenv("JXGPU_EIGH_TRACE") && (j0 % (64 * TD_NB)) == 0) {
                JX_HIP(hipStreamSynchronize(st));
                fprintf(stderr, "[jxgpu sytrd n=%d] syr2k after column %d done\n", n, j0);
                fflush(stderr);
            }
        }
    }
    if (tail_n) {
        const size_t lds = sizeof(double) * ((size_t)tail_n * (tail_n + 1) / 2 + 8 * (size_t)tail_n + 32);
        static bool attr_set = false;
        if (!attr_set) {
            JX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sytrd_tail_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(double) * ((size_t)TD_TAILN * (TD_TAILN + 1) / 2 + 8 * TD_TAILN + 32))));
            attr_set = true;
        }
        static long long *d_dbg = nullptr;
        if (getenv("JXGPU_EIGH_TRACE") && !d_dbg) JX_HIP(hipMalloc(&d_dbg, 6 * sizeof(long long)));
        hipLaunchKernelGGL(sytrd_tail_kernel, dim3(1), dim3(TD_TAIL_THREADS), lds, st, P, n - tail_n, d_dbg);
        if (d_dbg) {
            long long hk[6];
            JX_HIP(hipMemcpyAsync(hk, d_dbg, sizeof(hk), hipMemcpyDeviceToHost, st));
            JX_HIP(hipStreamSynchronize(st));
            fprintf(stderr, "[jxgpu sytrd tail t=%d] 100 MHz ticks: load %lld, scalars %lld, symv->B %lld, w->C %lld, update->D %lld, store %lld\n",
                    tail_n, hk[0], hk[1], hk[2], hk[3], hk[4], hk[5]);
        }
    }
    hipLaunchKernelGGL(sytrd_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, P);
    JX_LAUNCH_CHECK();
    for (int k = 0; k < 4; ++k) {
        JX_HIP(hipEventRecord(ev[2 * npanels + 2 * k], st));
        JX_HIP(hipEventRecord(ev[2 * npanels + 2 * k + 1], st));
    }
    JX_HIP(hipStreamSynchronize(st));  // workspace is freed on return
    double samp_ms = 0.0, pair_ms = 0.0;
    for (int k = 0; k < nsamp; ++k) {
        float ms = 0.f;
        JX_HIP(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
        samp_ms += ms;
    }
    for (int k = 0; k < 4; ++k) {
        float ms = 0.f;
        JX_HIP(hipEventElapsedTime(&ms, ev[2 * npanels + 2 * k], ev[2 * npanels + 2 * k + 1]));
        pair_ms += ms;
    }
    pair_ms /= 4.0;
    if (getenv("JXGPU_EIGH_TRACE"))
        fprintf(stderr, "[jxgpu sytrd n=%d] symv samples: %d, mean %.2f us (dispatch-bound events), empty hipEventRecord pair %.2f us\n", n, nsamp,
                nsamp ? samp_ms / nsamp * 1e3 : 0.0, pair_ms * 1e3);
    g_last_ms[2] = nsamp ? (float)(samp_ms / nsamp) : 0.f;
    g_last_ms[3] = nsamp ? (float)(samp_bytes / nsamp / 1e6) : 0.f;
    return 0;
}

// whether sytrd_lower would deal the symv tiles of an n-row problem over ranks (the eigensolver then keeps the one-stage form)
int sytrd_dist_active(int n) {
    static const bool force_single = getenv("JXGPU_DIST_EIGH_FORCE") && atoi(getenv("JXGPU_DIST_EIGH_FORCE")) != 0;
    return (!g_dist_paused && (g_dist.world > 1 || force_single) && g_dist.allreduce && n >= g_dist.min_n) ? 1 : 0;
}

void sytrd_dist_rank(int *rank, int *world) {
    *rank = g_dist_paused ? 0 : g_dist.rank;
    *world = g_dist_paused ? 1 : g_dist.world;
}

void sytrd_dist_pause(int on) { g_dist_paused = on != 0; }

int sytrd_set_dist(int rank, int world, int (*allreduce)(void *), void *user, double *staging, int64_t staging_doubles,
                   int min_n) {
    if (world < 1 || rank < 0 || rank >= world) return fail("sytrd_set_dist: bad rank / world");
    g_dist.rank = rank;
    g_dist.world = world;
    g_dist.allreduce = allreduce;
    g_dist.user = user;
    g_dist.staging = staging;
    g_dist.staging_doubles = staging_doubles;
    if (min_n > 0) g_dist.min_n = min_n;
    return 0;
}

}  // namespace jx
