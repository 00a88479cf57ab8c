// Stage 2 of the two-stage symmetric eigensolver behind src/math/eigh.rs:1422-1528: symmetric band (half bandwidth
// SB, k_sy2sb.hip) -> tridiagonal by bulge chasing, as ONE persistent launch.
//
// Sweep s (s = 0 .. n-3) eliminates column s below its sub-diagonal with a reflector on rows s+1 .. s+SB and chases
// the bulge down the band in steps k = 0, 1, ...: step k works on the window of columns r .. r+L-1, r = s+1+k SB,
//   D = A[r:r+L, r:r+L] (two-sided by the step's reflector),  B = A[r+L:r+L+L1, r:r+L] (reflector from the right, then a
//   new reflector from B's first column, applied from the left; it is the reflector of step k+1).
// Only the first column of every bulge is annihilated (Haidar, Ltaief, Dongarra 2011), so a window is 2 SB x SB
// doubles = one contiguous 64 KB run of the compact band storage ab[d + j ldab] = A[j+d, j], ldab = 2 SB.
// Step (s, k) overlaps the windows of (s-1, k) and (s-1, k+1) only: sweep s may run step k once sweep s-1 has finished
// step k+1.
// Two kernels: the POSITION-owned one below (sb2st_owned_kernel, the default: workgroup k keeps the window of step k in
// registers for the whole chase, 4.4 us per sweep) and the SWEEP-owned one (sb2st_chase_kernel, 18 us per sweep; the fallback
// when the positions do not all fit the device at once, and the reference the other is bit-identical to):
// workgroup g owns the sweeps g, g+G, g+2G, ...; progress[s] counts finished steps.  Windows are exchanged
// through memory with write-through (sc1) stores and L1-bypassing (sc1) loads, every storing wave drains its stores
// before the workgroup's barrier and ONE lane then publishes the counter (cdna_hip_programming.md guideline 16, form R1
// with sc1 loads on the consumer side); every spin is bounded and raises an abort flag the other workgroups honour.
// The reflectors are kept for the back-transformation (k_sbback.hip): v2[row + s n] (entry 1 explicit), tau2[s KS + k].
#include <stdlib.h>

#include "jx_common.h"

namespace jx {

constexpr int BC_SB = 64;                 // must equal SB of k_sy2sb.hip
constexpr int BC_LD = 2 * BC_SB;          // ldab
constexpr int BC_P = BC_SB + 1;           // LDS pitch
constexpr int BC_THREADS = 256;
constexpr int BC_DONE = 1 << 30;
constexpr unsigned BC_SPIN_LIMIT = 1u << 22;

struct BcParams {
    double *ab;
    int n;
    double *v2;        // (n, n) column-major: reflector of sweep s in column s, by matrix row
    double *tau2;      // (n, ks)
    int ks;
    int *prog;         // (n) finished steps per sweep
    int *abort_flag;
    int skip;          // diagnostic bit mask (JXGPU_BC_SKIP): 1 no arithmetic phases, 2 no window stores / loads
};

__device__ __forceinline__ double bc_ld(const double *p) {
    const unsigned long long u =
        __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)u);
}
__device__ __forceinline__ void bc_st(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum over the 64 lanes, the same value in every lane: DPP prefix steps (row_shr 1, 2, 4, 8, row_bcast 15, row_bcast 31: the
// total arrives in lane 63) and one v_readlane -- ~20 instructions on the chain instead of six ds_bpermute round trips.
// Both chase kernels use it, so they add in the same order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double bc_dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(phi, plo);
}
__device__ __forceinline__ double bc_bcast_lane(double v, int src_lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
__device__ __forceinline__ double bc_wave_sum(double v) {
    v = bc_dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = bc_dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = bc_dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = bc_dpp_add<0x118, 0xf>(v);      // row_shr:8
    v = bc_dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1, 3
    v = bc_dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2, 3
    return bc_bcast_lane(v, 63);
}

// LAPACK dlarfg on x[0 .. len-1] held one element per lane of wave 0 (lane < len): returns v (v[0] = 1), tau, beta
__device__ __forceinline__ void bc_house_wave(double x, int lane, int len, double &v, double &tau, double &beta) {
    double ss = (lane >= 1 && lane < len) ? x * x : 0.0;
    ss = bc_wave_sum(ss);
    const double alpha = bc_bcast_lane(x, 0);
    if (ss == 0.0) {
        tau = 0.0;
        beta = alpha;
        v = (lane == 0) ? 1.0 : 0.0;
        return;
    }
    const double nrm = sqrt(alpha * alpha + ss);
    beta = (alpha >= 0.0) ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    v = (lane == 0) ? 1.0 : ((lane < len) ? x * sc : 0.0);
}

typedef int bc_v4i __attribute__((ext_vector_type(4)));
constexpr int BC_NV = BC_SB * BC_LD / 2 / BC_THREADS;     // 16-byte pieces of a window per thread (16)
constexpr int BC_AUX_SC1 = 16;                            // cache-policy bit of the raw buffer builtins: sc1

// buffer descriptor over the window that starts at column r: accesses past the end of the band storage return zero /
// are dropped, so every thread always issues all BC_NV pieces (the counted s_waitcnt below relies on that)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bc_window_rsrc(double *ab, int n, int r) {
    const int64_t cols = (int64_t)n - r;
    const int64_t bytes = cols <= 0 ? 0 : (cols >= BC_SB ? (int64_t)BC_SB * BC_LD * 8 : cols * BC_LD * 8);
    return __builtin_amdgcn_make_buffer_rsrc(ab + (int64_t)(r < n ? r : 0) * BC_LD, 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void bc_window_load(__amdgpu_buffer_rsrc_t rs, int t, bc_v4i (&x)[BC_NV]) {
#pragma unroll
    for (int u = 0; u < BC_NV; ++u) x[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (u * BC_THREADS + t) * 16, 0, BC_AUX_SC1);
}

__device__ __forceinline__ bool bc_wait_progress(const BcParams &P, int s, int need, int prefetched) {
    if (s == 0 || prefetched >= need) return true;
    unsigned spins = 0;
    while (__hip_atomic_load(P.prog + s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > BC_SPIN_LIMIT || __hip_atomic_load(P.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
            return false;
    }
    return true;
}

__global__ __launch_bounds__(BC_THREADS) void sb2st_chase_kernel(BcParams P) {
    extern __shared__ __attribute__((aligned(16))) double bc_smem[];
    double(*E)[BC_P] = reinterpret_cast<double(*)[BC_P]>(bc_smem);          // [2 SB][SB + 1] window, dense
    double *vv = bc_smem + 2 * BC_SB * BC_P;    // [SB] current reflector
    double *vn = vv + BC_SB;                    // [SB] next reflector
    double *ww = vn + BC_SB;                    // [SB]
    double *part = ww + BC_SB;                  // [8][SB]
    double *sc = part + 8 * BC_SB;              // [8] scalars: 0 tau, 1 tau_next
    double *wcopy = sc + 8;                     // [4][SB] per-wave copy of w
    double *usum = wcopy + 4 * BC_SB;           // [4][16] column sums of the left application
    double *colscr = usum + 64;                 // [4][16][SB + 1] scratch of the column sums
    int *ish = reinterpret_cast<int *>(colscr + 4 * 16 * BC_P); // [2] 0: ok flag
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = P.n;
    const int row = t & 63, quarter = t >> 6;   // (row or column) x 16-wide slice decomposition of a 64 x 64 block

    for (int s = blockIdx.x; s < n - 2; s += gridDim.x) {
        int r = s + 1;
        int L = min(BC_SB, n - r);
        // ---- wait for sweep s-1 to have finished steps 0 and 1, then form the sweep's first reflector from column s
        if (t == 0) ish[0] = bc_wait_progress(P, s, 2, -1) ? 1 : 0;
        __syncthreads();
        if (!ish[0]) {
            if (t == 0) __hip_atomic_store(P.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        bc_v4i x[BC_NV];
        bc_window_load(bc_window_rsrc(P.ab, n, r), t, x);          // window of step 0
        if (wave == 0) {
            double *col = P.ab + (int64_t)s * BC_LD;
            const double xc = (lane < L) ? bc_ld(col + 1 + lane) : 0.0;
            double v, tau, beta;
            bc_house_wave(xc, lane, L, v, tau, beta);
            if (lane < L) {
                bc_st(col + 1 + lane, (lane == 0) ? beta : 0.0);
                vv[lane] = v;
            }
            if (lane == 0) sc[0] = tau;
        }
        for (int k = 0;; ++k) {
            const int L1 = min(BC_SB, n - (r + L));       // rows of the off-diagonal block (<= 0: none)
            const int rowsB = L1 > 0 ? L1 : 0;
            const int nrow = L + rowsB;
            // ---- window (registers, loaded during the previous step) -> LDS, dense, D mirrored to a full square
#pragma unroll
            for (int u = 0; u < BC_NV; ++u) {
                const int idx = 2 * (u * BC_THREADS + t);
                const int i = idx / BC_LD, d = idx % BC_LD;
                const int q = i + d;
                const double x0 = __hiloint2double(x[u][1], x[u][0]), x1 = __hiloint2double(x[u][3], x[u][2]);
                if (i < L) {
                    if (q < nrow) {
                        E[q][i] = x0;
                        if (q < L && q > i) E[i][q] = x0;
                    }
                    if (q + 1 < nrow) {
                        E[q + 1][i] = x1;
                        if (q + 1 < L) E[i][q + 1] = x1;
                    }
                }
            }
            // progress of the previous sweep, read early: by the end of this step it usually already allows step k + 1
            int seen = -1;
            if (t == 0 && s > 0 && rowsB > 0) seen = __hip_atomic_load(P.prog + s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const double tau = sc[0];
            double taun = 0.0;
            if (P.skip & 1) goto after_compute;
            // reflector of this step -> v2 / tau2 (read by the back-transformation after the launch)
            if (!(P.skip & 4)) {
                if (t < L) P.v2[(int64_t)s * n + r + t] = vv[t];
                if (t == 0) P.tau2[(int64_t)s * P.ks + k] = tau;
            }
            {
                // The arithmetic runs on registers: wave q (= quarter) holds the columns 16 q .. 16 q + 15 of D (dD) and of B
                // (dB), lane = row.  Row sums cross the four waves through LDS (one barrier), column sums are butterfly
                // reductions inside a wave, vector entries of other rows come by lane permutes.
                const int c0 = quarter * 16;
                double dD[16], dB[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    dD[c] = (row < L && c0 + c < L) ? E[row][c0 + c] : 0.0;
                    dB[c] = (row < rowsB && c0 + c < L) ? E[L + row][c0 + c] : 0.0;
                }
                const double v_lane = (lane < L) ? vv[lane] : 0.0;
                double vq[16];                                   // v of this wave's columns (vv is zero beyond L)
#pragma unroll
                for (int c = 0; c < 16; ++c) vq[c] = (c0 + c < L) ? vv[c0 + c] : 0.0;
                // ---- y = D v, z = B v: partial sums over this wave's 16 columns
                {
                    double py = 0.0, pz = 0.0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        py += dD[c] * vq[c];
                        pz += dB[c] * vq[c];
                    }
                    part[quarter * BC_SB + row] = py;
                    part[(4 + quarter) * BC_SB + row] = pz;
                }
                __syncthreads();
                const double y = (part[row] + part[BC_SB + row]) + (part[2 * BC_SB + row] + part[3 * BC_SB + row]);
                const double z = tau * ((part[4 * BC_SB + row] + part[5 * BC_SB + row]) + (part[6 * BC_SB + row] + part[7 * BC_SB + row]));
                double vy = y * v_lane;
                vy = bc_wave_sum(vy);
                const double w_lane = tau * y - (0.5 * tau * tau * vy) * v_lane;
                // w of this wave's columns: through a per-wave LDS copy (a wave's LDS operations complete in order)
                double *wsh = wcopy + wave * BC_SB;
                wsh[lane] = w_lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // ---- D <- H D H = D - v w' - w v';  B <- B H = B - (tau B v) v'
                if (!(P.skip & 32))
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const double wc = wsh[c0 + c];
                    dD[c] -= __dadd_rn(__dmul_rn(v_lane, wc), __dmul_rn(w_lane, vq[c]));   // no contraction: D stays exactly symmetric
                    dB[c] -= z * vq[c];
                }
                if (rowsB > 0) {
                    // ---- new reflector from the first column of B (wave 0, register 0)
                    if (wave == 0 && !(P.skip & 16)) {
                        double v, tn, beta;
                        bc_house_wave(dB[0], lane, rowsB, v, tn, beta);
                        dB[0] = (lane == 0) ? beta : 0.0;
                        vn[lane] = (lane < rowsB) ? v : 0.0;
                        if (lane == 0) sc[1] = tn;
                    }
                    __syncthreads();
                    taun = sc[1];
                    const double vn_lane = vn[lane];
                    // ---- B <- H1 B on the columns 1 .. L-1:  u = B' vn,  B -= taun vn u'.  The 16 column sums of a wave go
                    // through an LDS scratch (16 butterfly reductions are 96 dependent cross-lane steps, 3.4 us per step of
                    // the chase): every lane stores its 16 products, lane l then adds 16 rows of column l / 4 and two
                    // lane-exchange steps finish the sum.
                    if (!(P.skip & 8)) {
                        double *scr = colscr + wave * (16 * BC_P);
#pragma unroll
                        for (int c = 0; c < 16; ++c) scr[c * BC_P + lane] = vn_lane * dB[c];
                        __syncthreads();
                        {
                            const int cl = lane >> 2, sub = lane & 3;
                            double acc = 0.0;
#pragma unroll
                            for (int p = 0; p < 16; ++p) acc += scr[cl * BC_P + sub * 16 + p];
                            acc += __shfl_xor(acc, 1);
                            acc += __shfl_xor(acc, 2);
                            if (sub == 0) usum[wave * 16 + cl] = acc;
                        }
                        __syncthreads();
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            if (c0 + c == 0) continue;                  // wave-uniform: the eliminated column
                            dB[c] -= taun * vn_lane * usum[wave * 16 + c];
                        }
                    }
                }
                // registers -> LDS image for the write-back (lower part of D, B)
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (row < L && c0 + c < L) E[row][c0 + c] = dD[c];
                    if (row < rowsB && c0 + c < L) E[L + row][c0 + c] = dB[c];
                }
            }
        after_compute:
            __syncthreads();
            // ---- window -> memory (write-through 16-byte pieces); the positions below the window's rows are zero by the
            // band structure and are rewritten as such
            if (!(P.skip & 2)) {
                const __amdgpu_buffer_rsrc_t rs = bc_window_rsrc(P.ab, n, r);
#pragma unroll
                for (int u = 0; u < BC_NV; ++u) {
                    const int idx = 2 * (u * BC_THREADS + t);
                    const int i = idx / BC_LD, d = idx % BC_LD;
                    const int q = i + d;
                    const double x0 = (i < L && q < nrow) ? E[q][i] : 0.0;
                    const double x1 = (i < L && q + 1 < nrow) ? E[q + 1][i] : 0.0;
                    bc_v4i o;
                    o[0] = __double2loint(x0);
                    o[1] = __double2hiint(x0);
                    o[2] = __double2loint(x1);
                    o[3] = __double2hiint(x1);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs, (u * BC_THREADS + t) * 16, 0, BC_AUX_SC1);
                }
            }
            // may step k + 1 start?  (the previous sweep must have finished its step k + 2)
            if (t == 0) ish[0] = (rowsB > 0) ? (bc_wait_progress(P, s, k + 3, seen) ? 1 : 0) : 1;
            __syncthreads();                              // also: every wave has read its part of E
            if (!ish[0]) {
                if (t == 0) __hip_atomic_store(P.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (rowsB > 0) {
                // the next window's loads go out behind the stores; the counted wait below retires the stores (memory
                // operations of a wave complete in issue order) and leaves the BC_NV loads in flight
                if (!(P.skip & 2)) bc_window_load(bc_window_rsrc(P.ab, n, r + L), t, x);
                if (!(P.skip & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BC_NV) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            if (t == 0)
                __hip_atomic_store(P.prog + s, (rowsB > 0) ? (k + 1) : BC_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rowsB <= 0) break;
            // next step: the reflector just formed
            if (t < BC_SB) vv[t] = (t < rowsB) ? vn[t] : 0.0;
            if (t == 0) sc[0] = taun;
            r += L;
            L = rowsB;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Position-owned form.  Workgroup k owns STEP k of every sweep: its window (D, B of the columns [s + 1 + 64 k, + 64)) stays
// in registers for the whole chase and slides by one row and column per sweep; what travels between workgroups is
//   R(s, k + 1): the reflector formed by step (s, k)                    (k -> k + 1, 64 doubles + tau),
//   C(s, k - 1): column 0 of the window after step (s, k) (final for the sweep: D[0..63][0], B[0][0]; it is the column
//                that enters the window of position k - 1 when that slides to sweep s + 1)   (k -> k - 1, 65 doubles),
// instead of a 64 KB window per step.  Both leave as soon as the new reflector exists (before the left application on
// B, which is off the chain).  Position 0 keeps its column: it is the source of the next sweep's first reflector and of
// d[s + 1], e[s + 1].  The arithmetic of a step is the statement sequence of sb2st_chase_kernel on the same values (the
// symmetric update is written so that D stays exactly symmetric in both), so d, e, v2 and tau2 are bit-identical to it.
// Every workgroup must be resident (the host checks the occupancy and falls back to the sweep-owned kernel); every poll is
// bounded and raises the abort flag (the driver then reduces the matrix in one stage, as for the sweep-owned kernel).
struct BoParams {
    const double *ab;
    int n;
    double *v2, *tau2;
    int ks;
    double *d, *e;
    double *rmsg;      // [positions][2 slots][BO_MSG] cells of 16 bytes: v[0..63], tau
    double *cmsg;      // [positions][2 slots][BO_MSG] cells: D[0..63][0], B[0][0]
    int *abort_flag;
    int poll_sleep;    // diagnostic (JXGPU_BO_SLEEP): s_sleep units between two polls
};
constexpr int BO_MSG = 72;
// A message has no flag: every value travels in a 16-byte cell {value, value bits ^ pattern(sequence number)} written by ONE
// 16-byte write-through store and the receiver polls the cells themselves -- one store and one load on the chain instead of
// store, drain, flag store, flag load, payload load.  A cell whose halves belonged to different messages would fail the
// check (the tag binds the value), so the protocol does not rest on the 16 bytes arriving together.
__device__ __forceinline__ unsigned long long bo_pattern(int seq) {
    return ((unsigned long long)(unsigned)seq << 32) | (unsigned long long)(unsigned)seq;
}
__device__ __forceinline__ void bo_put(__amdgpu_buffer_rsrc_t rs, int cell, double v, int seq) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v), tag = bits ^ bo_pattern(seq);
    bc_v4i o;
    o[0] = (int)(unsigned)bits;
    o[1] = (int)(unsigned)(bits >> 32);
    o[2] = (int)(unsigned)tag;
    o[3] = (int)(unsigned)(tag >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(o, rs, cell * 16, 0, BC_AUX_SC1);
}
__device__ __forceinline__ bool bo_check(const bc_v4i &x, int seq, double &v) {
    const unsigned long long bits = ((unsigned long long)(unsigned)x[1] << 32) | (unsigned)x[0];
    const unsigned long long tag = ((unsigned long long)(unsigned)x[3] << 32) | (unsigned)x[2];
    v = __longlong_as_double((long long)bits);
    return (bits ^ tag) == bo_pattern(seq);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bo_slot(double *base, int pos, int slot) {
    return __builtin_amdgcn_make_buffer_rsrc(base + ((int64_t)pos * 2 + slot) * (2 * BO_MSG), 0, BO_MSG * 16, 0x00020000);
}

constexpr int BO_LDS_DOUBLES = 2 * BC_SB + BC_SB + 8 * BC_SB + 8 + 64 + 4 * 16 * BC_P + BC_SB;

// DEPTH = polls in flight per wave (1: measured best), WPE = waves per SIMD the register budget is cut for (WPE workgroups per CU)
template <int DEPTH, int WPE>
__global__ __launch_bounds__(BC_THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void sb2st_owned_kernel(BoParams P) {
    // dynamic LDS (39 KB): a static size would make the compiler assume one workgroup per CU and take 250 registers
    extern __shared__ __attribute__((aligned(16))) double bo_smem[];
    double *vv0 = bo_smem;                       // [2][SB] position 0: first reflector of the sweep, by sweep parity
    double *vn = vv0 + 2 * BC_SB;                // [SB] the reflector formed in this step
    double *part = vn + BC_SB;                   // [8][SB] partial row sums
    double *sc = part + 8 * BC_SB;               // [8] 1: tau of vn, 2 + parity: tau of vv0
    double *usum = sc + 8;                       // [4][16]
    double *colscr = usum + 64;                  // [4][16][SB + 1]
    double *brow = colscr + 4 * 16 * BC_P;       // [SB]
    double *edge = colscr;                       // [2][4][SB]: column 0 of every wave's D and B (slide; colscr is idle then)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = P.n, k = blockIdx.x;
    const int row = lane, quarter = wave, c0 = quarter * 16;

    // ---- the window of sweep 0
    double dD[16], dB[16];
    {
        const int r = 1 + BC_SB * k, L = min(BC_SB, n - r);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int col = c0 + c;
            const int lo = min(row, col), hi = max(row, col);
            dD[c] = (row < L && col < L) ? P.ab[(int64_t)(r + lo) * BC_LD + (hi - lo)] : 0.0;
            dB[c] = (col < L && r + BC_SB + row < n) ? P.ab[(int64_t)(r + col) * BC_LD + (BC_SB + row - col)] : 0.0;
        }
    }
    // position 0: the reflector of the next sweep is formed by wave 0 and handed to the other waves through vv0[parity]
    if (k == 0 && wave == 0) {
        const int L = min(BC_SB, n - 1);
        const double xc = (lane < L) ? P.ab[1 + lane] : 0.0;
        double v0, tau0, beta;
        bc_house_wave(xc, lane, L, v0, tau0, beta);
        vv0[lane] = v0;
        if (lane == 0) {
            sc[2] = tau0;
            P.d[0] = P.ab[0];
            P.e[0] = beta;
        }
    }
    const int c0s = __builtin_amdgcn_readfirstlane(c0);
    __syncthreads();

    for (int s = 0; s < n - 2; ++s) {
        const int r = s + 1 + BC_SB * k;
        if (r >= n) break;
        const int L = min(BC_SB, n - r);
        const int L1 = min(BC_SB, n - (r + L));
        const int rowsB = L1 > 0 ? L1 : 0;
        const bool need_r = k > 0, need_c = s > 0 && r + BC_SB - 1 < n && wave == 3;
        // ---- messages, polled by every wave for itself (no LDS hop, no barrier): the reflector of this step (lane = entry);
        // wave 3 also takes the column that completes the slide (lane l: cell 1 + l = its row of B's entering column, lane 63
        // also cell 0 = the new diagonal entry)
        double v_lane, tau, ca = 0.0, cb = 0.0;
        if (need_r || need_c) {
            const __amdgpu_buffer_rsrc_t rr = bo_slot(P.rmsg, k, s & 1), rc = bo_slot(P.cmsg, k, (s - 1) & 1);
            // DEPTH polls in flight: memory is sampled every 1 / DEPTH of a round trip
            struct Raw {
                bc_v4i a, b, c, d;
            };
            auto issue = [&](Raw &x) {
                if (need_r) {
                    x.a = __builtin_amdgcn_raw_buffer_load_b128(rr, lane * 16, 0, BC_AUX_SC1);
                    if (lane == 0) x.b = __builtin_amdgcn_raw_buffer_load_b128(rr, BC_SB * 16, 0, BC_AUX_SC1);
                }
                if (need_c) {
                    x.c = __builtin_amdgcn_raw_buffer_load_b128(rc, (1 + lane) * 16, 0, BC_AUX_SC1);
                    if (lane == 63) x.d = __builtin_amdgcn_raw_buffer_load_b128(rc, 0, 0, BC_AUX_SC1);
                }
            };
            double ra = 0.0, rb = 0.0;
            auto take = [&](const Raw &x) {
                bool good = true;
                if (need_r) {
                    good = bo_check(x.a, s + 1, ra) && good;
                    if (lane == 0) good = bo_check(x.b, s + 1, rb) && good;
                }
                if (need_c) {
                    good = bo_check(x.c, s, ca) && good;
                    if (lane == 63) good = bo_check(x.d, s, cb) && good;
                }
                return __ballot(good) == ~0ull;
            };
            Raw x[DEPTH];
            unsigned spins = 0;
#pragma unroll
            for (int d = 0; d + 1 < DEPTH; ++d) issue(x[d]);
            for (bool got = false; !got;) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    if (!got) {
                        issue(x[(d + DEPTH - 1) % DEPTH]);
                        got = take(x[d]);
                    }
                }
                if (got) break;
                for (int q = 0; q < P.poll_sleep; ++q) __builtin_amdgcn_s_sleep(1);
                ++spins;
                if (spins > (BC_SPIN_LIMIT >> 2) ||
                    ((spins & 63u) == 0 && __hip_atomic_load(P.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    // a wave that ends leaves the workgroup's barriers; the other waves meet the flag in their next poll
                    if (lane == 0) __hip_atomic_store(P.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
            }
            v_lane = ra;
            tau = bc_bcast_lane(rb, 0);
        }
        if (!need_r) {
            v_lane = vv0[(s & 1) * BC_SB + lane];
            tau = sc[2 + (s & 1)];
        }
        v_lane = (lane < L) ? v_lane : 0.0;
        // ---- the slide's last piece: the entering column (logical column 63 = wave 3, register 15)
        if (s > 0 && wave == 3) {
            if (lane == 63) dD[15] = cb;
            dB[15] = ca;
        }
        double taun = 0.0;
        if (wave == 0) {
            if (lane < L) P.v2[(int64_t)s * n + r + lane] = v_lane;
            if (lane == 0) P.tau2[(int64_t)s * P.ks + k] = tau;
        }
        double vq[16];                                   // v of this wave's columns: wave-uniform (scalar registers)
#pragma unroll
        for (int c = 0; c < 16; ++c) vq[c] = bc_bcast_lane(v_lane, c0s + c);
        // ---- y = D v, z = B v
        {
            double py = 0.0, pz = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                py += dD[c] * vq[c];
                pz += dB[c] * vq[c];
            }
            part[quarter * BC_SB + row] = py;
            part[(4 + quarter) * BC_SB + row] = pz;
        }
        __syncthreads();
        const double z = tau * ((part[4 * BC_SB + row] + part[5 * BC_SB + row]) + (part[6 * BC_SB + row] + part[7 * BC_SB + row]));
        // ---- wave 0: the new reflector needs only column 0 of B H; it leaves before anything else is updated
        double vnl = 0.0, beta_n = 0.0;
        if (wave == 0 && rowsB > 0) {
            double tn;
            bc_house_wave(dB[0] - z * vq[0], lane, rowsB, vnl, tn, beta_n);
            vnl = (lane < rowsB) ? vnl : 0.0;
            const __amdgpu_buffer_rsrc_t rs = bo_slot(P.rmsg, k + 1, s & 1);
            bo_put(rs, lane, vnl, s + 1);
            if (lane == 0) bo_put(rs, BC_SB, tn, s + 1);
            vn[lane] = vnl;
            if (lane == 0) sc[1] = tn;
        }
        const double y = (part[row] + part[BC_SB + row]) + (part[2 * BC_SB + row] + part[3 * BC_SB + row]);
        double vy = y * v_lane;
        vy = bc_wave_sum(vy);
        const double w_lane = tau * y - (0.5 * tau * tau * vy) * v_lane;
        // ---- D <- H D H = D - v w' - w v';  B <- B H = B - (tau B v) v'
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double wc = bc_bcast_lane(w_lane, c0s + c);
            dD[c] -= __dadd_rn(__dmul_rn(v_lane, wc), __dmul_rn(w_lane, vq[c]));
            dB[c] -= z * vq[c];
        }
        if (wave == 0) {
            if (rowsB > 0) dB[0] = (lane == 0) ? beta_n : 0.0;
            if (k > 0) {
                const __amdgpu_buffer_rsrc_t rs = bo_slot(P.cmsg, k - 1, s & 1);
                bo_put(rs, lane, dD[0], s + 1);
                if (lane == 0) bo_put(rs, BC_SB, dB[0], s + 1);
            } else {
                // position 0: column s + 1 is final: d[s + 1], and either the next sweep's reflector or the last entries
                const double below = __shfl(dD[0], (lane + 1) & 63), b00 = bc_bcast_lane(dB[0], 0);
                const double xc = (lane < 63) ? below : b00;          // A(s + 2 + lane, s + 1)
                if (lane == 0) P.d[s + 1] = dD[0];
                if (s + 1 < n - 2) {
                    const int Ln = min(BC_SB, n - (s + 2));
                    double v0, tau0, beta;
                    bc_house_wave((lane < Ln) ? xc : 0.0, lane, Ln, v0, tau0, beta);
                    vv0[((s + 1) & 1) * BC_SB + lane] = v0;
                    if (lane == 0) {
                        sc[2 + ((s + 1) & 1)] = tau0;
                        P.e[s + 1] = beta;
                    }
                } else {
                    if (lane == 0) P.e[s + 1] = xc;
                    if (lane == 1) {
                        P.d[s + 2] = dD[1];
                        P.e[s + 2] = 0.0;
                    }
                }
            }
        }
        if (rowsB > 0) {
            __syncthreads();
            taun = sc[1];
            const double vn_lane = vn[lane];
            // ---- B <- H1 B on the columns 1 .. L-1 (off the chain)
            double *scr = colscr + wave * (16 * BC_P);
#pragma unroll
            for (int c = 0; c < 16; ++c) scr[c * BC_P + lane] = vn_lane * dB[c];
            __syncthreads();
            {
                const int cl = lane >> 2, sub = lane & 3;
                double acc = 0.0;
#pragma unroll
                for (int p = 0; p < 16; ++p) acc += scr[cl * BC_P + sub * 16 + p];
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                if (sub == 0) usum[wave * 16 + cl] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c0 + c == 0) continue;
                dB[c] -= taun * vn_lane * usum[wave * 16 + c];
            }
        }
        __syncthreads();                                   // colscr (edge) and part are free
        // ---- slide by one row and column: D'[i][j] = D[i+1][j+1], D'[63][j] = B[0][j+1] = D'[j][63], B'[i][j] = B[i+1][j+1],
        // row 63 of B' is zero up to the entering column (set at the top of the next sweep)
        edge[quarter * BC_SB + lane] = dD[0];
        edge[(4 + quarter) * BC_SB + lane] = dB[0];
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c0 + c >= 1) brow[c0 + c - 1] = dB[c];
        }
        __syncthreads();
        {
            const double eD = (quarter < 3) ? edge[(quarter + 1) * BC_SB + lane] : 0.0;
            const double eB = (quarter < 3) ? edge[(4 + quarter + 1) * BC_SB + lane] : 0.0;
            const int src = (lane + 1) & 63;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double tD = (c < 15) ? dD[(c + 1) & 15] : eD;
                const double tB = (c < 15) ? dB[(c + 1) & 15] : eB;
                const double rD = __shfl(tD, src), rB = __shfl(tB, src);
                dD[c] = (lane < 63) ? rD : rB;
                dB[c] = (lane < 63) ? rB : 0.0;
            }
            if (quarter == 3) {
                dD[15] = (lane < 63) ? brow[lane] : 0.0;
                dB[15] = 0.0;
            }
        }
        // (the barrier at the top of the next sweep separates these reads of edge / brow from their next writes)
    }
}

__global__ void sb2st_de_kernel(const double *__restrict__ ab, int n, double *__restrict__ d, double *__restrict__ e) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    d[j] = ab[(int64_t)j * BC_LD];
    e[j] = (j < n - 1) ? ab[(int64_t)j * BC_LD + 1] : 0.0;     // e[n-1] = 0: the divide and conquer reads n entries
}

int sb2st_ldab() { return BC_LD; }
int sb2st_steps(int n) { return (n + BC_SB - 1) / BC_SB + 1; }
static int sb2st_positions(int n) { return n > 2 ? (n - 2) / BC_SB + 1 : 0; }
static size_t sb2st_ctrl_ints(int n) { return (((size_t)n + 4) + 3) & ~(size_t)3; }
// bytes of d_ctrl: progress counters / message flags + abort flag, then the message slots of the position-owned kernel
size_t sb2st_ctrl_bytes(int n) {
    return sizeof(int) * sb2st_ctrl_ints(n) + sizeof(double) * (size_t)sb2st_positions(n) * 4 * (2 * BO_MSG);
}

// d_ab (2 SB x n band, lower, ld = 2 SB; the sweep-owned kernel destroys it) -> d, e of the tridiagonal matrix; reflectors to
// d_v2 (n x n, column s = sweep s; only the entries written are meaningful) and d_tau2 (n x sb2st_steps(n)).
// d_ctrl: sb2st_ctrl_bytes(n), zeroed here.  d_ctrl[n] != 0 after synchronisation = a bounded wait expired (never
// observed; the caller then reports an error instead of hanging).
// JXGPU_BC_OWNED=0 selects the sweep-owned kernel (also taken when the positions do not all fit the device at once).
int sb2st_chase(hipStream_t st, double *d_ab, int n, double *d_d, double *d_e, double *d_v2, double *d_tau2, int *d_ctrl) {
    const int ks = sb2st_steps(n);
    JX_HIP(hipMemsetAsync(d_ctrl, 0, sizeof(int) * ((size_t)n + 4), st));
    JX_HIP(hipMemsetAsync(d_tau2, 0, sizeof(double) * (size_t)n * ks, st));
    static int cus = 0, occ_fast = 0, occ_wide = 0;
    const size_t owned_lds = sizeof(double) * BO_LDS_DOUBLES;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        JX_HIP(hipGetDevice(&dev));
        JX_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_fast, (const void *)sb2st_owned_kernel<1, 2>, BC_THREADS, owned_lds) != hipSuccess)
            occ_fast = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_wide, (const void *)sb2st_owned_kernel<1, 4>, BC_THREADS, owned_lds) != hipSuccess)
            occ_wide = 0;
    }
    const int np = sb2st_positions(n);
    const char *om = getenv("JXGPU_BC_OWNED");
    // every position must be resident at once: the spill-free form where two workgroups per CU hold them all (n <= 32 768),
    // else the 128-register form at four workgroups per CU (n <= 65 536); beyond that the sweep-owned kernel.
    // (More than one poll in flight per wave is slower -- 88.8 / 94.1 / 106.3 / 116.8 ms at n = 20 000 for 1 / 2 / 3 / 4 --
    // and so is pausing between polls: 91.8 / 93.6 / 98.6 / 105.6 ms for 0 / 2 / 6 / 16 s_sleep units.)
    // (the occupancy query does not know the waves-per-SIMD cap of the register budget: <1, 2> is held to two workgroups
    // per CU by its attribute, <1, 4> to four)
    const bool fast = np <= cus * (occ_fast < 2 ? occ_fast : 2), wide = np <= cus * (occ_wide < 4 ? occ_wide : 4);
    const bool owned = n > 2 * BC_SB + 2 && !(om && atoi(om) == 0) && (fast || wide) && 2 * np <= n;
    if (owned) {
        double *msg = reinterpret_cast<double *>(d_ctrl + sb2st_ctrl_ints(n));
        BoParams P{d_ab, n, d_v2, d_tau2, ks, d_d, d_e, msg, msg + (size_t)np * 2 * (2 * BO_MSG), d_ctrl + n,
                   getenv("JXGPU_BO_SLEEP") ? atoi(getenv("JXGPU_BO_SLEEP")) : 0};
        JX_HIP(hipMemsetAsync(msg, 0, sizeof(double) * (size_t)np * 4 * (2 * BO_MSG), st));   // tags of an earlier chase
        if (fast && !(om && atoi(om) == 2))
            hipLaunchKernelGGL((sb2st_owned_kernel<1, 2>), dim3(np), dim3(BC_THREADS), owned_lds, st, P);
        else
            hipLaunchKernelGGL((sb2st_owned_kernel<1, 4>), dim3(np), dim3(BC_THREADS), owned_lds, st, P);
        JX_LAUNCH_CHECK();
        return 0;
    }
    if (n > 2) {
        BcParams P{d_ab, n, d_v2, d_tau2, ks, d_ctrl, d_ctrl + n, getenv("JXGPU_BC_SKIP") ? atoi(getenv("JXGPU_BC_SKIP")) : 0};
        // sweeps in flight are at most half the steps of a sweep (lag of two steps); one workgroup per CU
        int g = (ks + 1) / 2 + 1;
        if (getenv("JXGPU_SB2ST_WGS") && atoi(getenv("JXGPU_SB2ST_WGS")) > 0) g = atoi(getenv("JXGPU_SB2ST_WGS"));
        if (g > cus) g = cus;
        if (g > n - 2) g = n - 2;
        // > 80 KB of LDS per workgroup: at most one workgroup per CU (the hand-off form is measured for that geometry)
        const size_t lds = 120 * 1024;
        static bool attr_set = false;
        if (!attr_set) {
            JX_HIP(hipFuncSetAttribute((const void *)sb2st_chase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(sb2st_chase_kernel, dim3(g), dim3(BC_THREADS), lds, st, P);
        JX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(sb2st_de_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_ab, n, d_d, d_e);
    JX_LAUNCH_CHECK();
    return 0;
}

}  // namespace jx
