// Symmetric eigendecomposition on the device, replacing LAPACK dsyevd/dsyevr behind src/math/eigh.rs:1422-1528
// (`symmetric_eigh_f64_row_major_with_driver`): own tridiagonalisation (k_sytrd.hip), divide and conquer (k_stedc.hip)
// and back-transformation (k_ormtr.hip); rocSOLVER dsyevd below n = 256 or with JXGPU_EIGH=rocsolver.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <memory>
#include <mutex>
#include <vector>

#include "jx_common.h"

namespace jx {
// kernels from k_misc.hip
int launch_add_diag(double *d_a, int n, int64_t ld, double ridge, hipStream_t st);
int launch_symmetrize(double *d_a, int n, hipStream_t st);
// k_sytrd.hip: two-kernels-per-column Householder tridiagonalisation, LAPACK dsytrd(lower) output format
int sytrd_lower(rocblas_handle h, hipStream_t st, double *d_a, int n, double *d_d, double *d_e, double *d_tau);
// k_stedc.hip: Cuppen divide and conquer (batched QL leaves, merges level by level; no 32-bit size limit)
int stedc_split(rocblas_handle h, hipStream_t st, int n, double *d_d, double *d_e, double *d_c, int leaf,
                std::vector<int> &h_perm, int sel_lo = 0, int sel_hi = 0);
int launch_gather_cols(const double *src, const int *d_perm, int n, double *dst, hipStream_t st);
void gather_cols_grid(int n, unsigned *gx, unsigned *gy);
int sytrd_set_dist(int rank, int world, int (*allreduce)(void *), void *user, double *staging, int64_t staging_doubles,
                   int min_n);
// k_ormtr.hip: C <- Q C with wide compact-WY blocks (dormtr left / lower / no-transpose)
int ormtr_lower(hipStream_t st, const double *d_a, int n, const double *d_tau, double *d_c);
// k_sy2sb.hip / k_sb2st.hip: two-stage reduction (dense -> band -> tridiagonal)
size_t sy2sb_work_doubles(int n);
int sy2sb_bandwidth();
int sy2sb_lower(hipStream_t st, double *d_a, int n, double *d_tau, double *d_ab, int ldab, double *d_work, int *d_flags,
                bool allow_shard = false);
int sy2sb_set_band_dist(int rank, int world, int (*allreduce)(void *, int64_t), void *user, double *staging,
                        int64_t staging_doubles, int min_n, int block);
int64_t sy2sb_band_staging_doubles(int n);
int sy2sb_band_dist_active(int n);
int sb2st_ldab();
int sb2st_steps(int n);
size_t sb2st_ctrl_bytes(int n);
int sb2st_chase(hipStream_t st, double *d_ab, int n, double *d_d, double *d_e, double *d_v2, double *d_tau2, int *d_ctrl);
// k_sbback.hip: C <- Q2 C (reflectors of the bulge chasing); k_ormtr.hip: C <- Q1 C (reflectors of the band reduction)
size_t sbback_tq_doubles(int n, int ks);
int sbback_apply_q2(hipStream_t st, const double *d_v2, const double *d_tau2, int n, int ks, double *d_c, int ncols,
                    double *d_tq, hipEvent_t ev_start, hipEvent_t ev_stop);
extern float g_last_ms[24];   // [4] Q2 apply kernel ms, [5] its algorithmic GFLOP, [6] band reduction ms, [7] bulge chasing ms,
                              // [8] divide and conquer ms, [9] Q1 back-transformation ms, [10] 1 = two-stage path taken
int ormtr_lower_off(hipStream_t st, const double *d_a, int n, int off, int nref, const double *d_tau, double *d_c, int ncols);
// the C-independent part of the Q1 back-transformation (V images, Gram, T, V T per block) prepared ahead on a side stream
struct OrmtrPlan;
OrmtrPlan *ormtr_plan_new();
void ormtr_plan_free(OrmtrPlan *p);
int ormtr_prepare(hipStream_t st, const double *d_a, int n, int off, int nref, const double *d_tau, OrmtrPlan &plan);
int ormtr_apply(hipStream_t st, const OrmtrPlan &plan, double *d_c, int ncols);
int sytrd_dist_active(int n);
void sytrd_dist_rank(int *rank, int *world);
void sytrd_dist_pause(int on);
int launch_gather_cols_range(const double *src, const int *d_perm, int n, int count, double *dst, hipStream_t st);
// Node-level distribution of the two-stage path (jxg_eigh_set_gather): reduction stages and divide and conquer run
// replicated (bit-reproducible kernels, identical inputs), every rank back-transforms its own share of the eigenvectors
// and `gather` fills in the others' rows.
struct EighGather {
    int (*gather)(void *) = nullptr;
    void *user = nullptr;
    // optional: agree(user, checksum) -> 1 when every rank reports the same checksum of its replicated intermediate
    // results (eigenvalues of T, the divide and conquer's permutation, a row of its eigenvector matrix), 0 when they
    // differ, < 0 on failure
    int (*agree)(void *, uint64_t) = nullptr;
    void *agree_user = nullptr;
};
static EighGather g_gather;
constexpr int kRocsolverStedcMaxN = 46340;   // n^2 < 2^31: rocSOLVER 7.2 dstedc faults above (measured at n = 50000)

static rocblas_handle g_handle = nullptr;
static std::mutex g_handle_mu;

static rocblas_handle get_handle() {
    std::lock_guard<std::mutex> lk(g_handle_mu);
    if (!g_handle) {
        if (rocblas_create_handle(&g_handle) != rocblas_status_success) g_handle = nullptr;
        // replicas of a multi-rank decomposition must agree bit for bit: no atomics-based split-K inside rocBLAS
        if (g_handle) rocblas_set_atomics_mode(g_handle, rocblas_atomics_not_allowed);
    }
    return g_handle;
}
}  // namespace jx

using namespace jx;

// Node-level distribution of the tridiagonalisation's symv (k_sytrd.hip): every rank calls jxg_eigh_f64 on the same
// matrix; from `min_n` rows on each rank streams 1 / world of the tiles per column and `allreduce(user)` has to sum the
// `jxg_eigh_dist_staging_doubles(n)` doubles at `d_staging` over the ranks on the stream passed to jxg_eigh_f64.
extern "C" int64_t jxg_eigh_dist_staging_doubles(int n) { return (int64_t)n + 32 * 16 + 2 * 64; }

// A rank of a multi-rank job decomposes matrices of its OWN between on = 1 and on = 0 (the diagonal blocks of a sparse GRM, dealt
// over the ranks by the caller): no collective, no sharding, the one-rank thresholds.  The registered callbacks stay in place.
extern "C" int jxg_eigh_set_local(int on) {
    jx::sytrd_dist_pause(on);
    return 0;
}

extern "C" int jxg_eigh_set_dist(int rank, int world, int (*allreduce)(void *), void *user, double *d_staging,
                                 int64_t staging_doubles, int min_n) {
    return sytrd_set_dist(rank, world, allreduce, user, d_staging, staging_doubles, min_n);
}

// Two-stage path on several ranks (rank / world from jxg_eigh_set_dist): rank r finishes the eigenvectors
// [n r / world, n (r + 1) / world) -- rows of the row-major result in d_a -- and then calls gather(user), which must
// deliver every other rank's rows into the same d_a (all ranks call it; e.g. one broadcast per rank).  NULL: off.
extern "C" int jxg_eigh_set_gather(int (*gather)(void *), void *user) {
    g_gather.gather = gather;
    g_gather.user = user;
    return 0;
}

// Band reduction with the trailing matrix sharded over the ranks (k_sy2sb.hip, BandDist): ownership by block rows of `block`
// samples dealt cyclically, two collectives per panel through allreduce(user, count) -- the sum over the ranks of the first
// `count` doubles of d_staging, on the stream passed to jxg_eigh_f64 -- from `min_n` rows on (<= 0: 8192).  Takes effect on the
// two-stage path of a multi-rank decomposition (jxg_eigh_set_dist + jxg_eigh_set_gather).  d_staging needs
// jxg_eigh_band_staging_doubles(n) doubles.  allreduce = NULL: off.
extern "C" int jxg_eigh_set_band_dist(int rank, int world, int (*allreduce)(void *, int64_t), void *user, double *d_staging,
                                      int64_t staging_doubles, int min_n, int block) {
    return sy2sb_set_band_dist(rank, world, allreduce, user, d_staging, staging_doubles, min_n, block);
}
extern "C" int64_t jxg_eigh_band_staging_doubles(int n) { return sy2sb_band_staging_doubles(n); }
static int g_last_dc_windowed = 0;    // 1 when the last decomposition's top-level merge formed this rank's columns only
extern "C" int jxg_eigh_last_dc_windowed(void) { return g_last_dc_windowed; }
static int g_last_band_sharded = 0;   // 1 when the last decomposition ran the band reduction on a sharded trailing matrix
extern "C" int jxg_eigh_last_band_sharded(void) { return g_last_band_sharded; }

// Agreement check in front of the sharded back-transformations: the ranks' replicated results (tridiagonal eigenvalues,
// divide-and-conquer permutation, first row of its eigenvector matrix) are hashed and `agree(user, checksum)` has to say
// whether every rank holds the same value (e.g. a MIN and a MAX all-reduce).  When they differ every rank takes the
// unsharded back-transformation (its own, self-consistent result) instead of mixing rows of different bases.  NULL: off.
extern "C" int jxg_eigh_set_agree(int (*agree)(void *, uint64_t), void *user) {
    g_gather.agree = agree;
    g_gather.agree_user = user;
    return 0;
}

static int g_last_dist_agree = -1;   // -1 not checked, 1 replicas agreed, 0 they differed (unsharded fallback taken)
extern "C" int jxg_eigh_last_dist_agree(void) { return g_last_dist_agree; }

static uint64_t fnv1a(const void *p, size_t bytes, uint64_t h) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < bytes; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

// d_a: (n,n) symmetric, f64. On return row j of d_a (row-major) = eigenvector j (= column j of the
// column-major LAPACK result), eigenvalues ascending in d_w.
// launch geometry of the n-dependent two-dimensional grids of the eigensolver (gridDim.y <= 65535, gridDim.x < 2^31):
// 1 when every one of them is valid for an n-row problem.  No GPU needed (CPU test for n beyond 65535).
extern "C" int jxg_eigh_grid_check(int n) {
    if (n <= 0) return 0;
    unsigned gx = 0, gy = 0;
    gather_cols_grid(n, &gx, &gy);
    if (gy > 65535u || gx > 0x7fffffffu) return 0;
    const int ks = sb2st_steps(n);                       // sbback_tfactor_kernel: (ks, groups of 32 sweeps)
    const int64_t groups = ((int64_t)n - 2 + 31) / 32;
    if (ks > 0x7fffffff || groups > 65535) return 0;
    const int64_t band_blocks = ((int64_t)n * sb2st_ldab() + 255) / 256;   // sb_extract_band_kernel (1-D)
    if (band_blocks > 0x7fffffffLL) return 0;
    return 1;
}

static std::mutex g_eigh_mu;   // one decomposition at a time: the rocBLAS handle and the kept scratch blocks are process-global

extern "C" int jxg_eigh_f64(double *d_a, int n, double ridge, double *d_w, void *stream) {
    if (n <= 0) return fail("jxg_eigh_f64: n must be > 0");
    std::lock_guard<std::mutex> eigh_lock(g_eigh_mu);
    hipStream_t st = (hipStream_t)stream;
    rocblas_handle h = get_handle();
    if (!h) return fail("rocblas_create_handle failed");
    if (rocblas_set_stream(h, st) != rocblas_status_success) return fail("rocblas_set_stream failed");
    if (ridge != 0.0) {
        if (launch_add_diag(d_a, n, n, ridge, st)) return 1;
    }
    DevBuf e, info;
    if (e.alloc(sizeof(double) * (size_t)n)) return 1;
    if (info.alloc(sizeof(rocblas_int))) return 1;
    JX_HIP(hipMemsetAsync(info.p, 0, sizeof(rocblas_int), st));
    // symmetric input: row-major == column-major; the lower triangle is referenced.
    const char *mode = getenv("JXGPU_EIGH");
    const bool use_lib = (n < 256) || (mode && strcmp(mode, "rocsolver") == 0);
    if (use_lib) {
        rocblas_status rs = rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, n, d_a, n, d_w,
                                             e.as<double>(), info.as<rocblas_int>());
        if (rs != rocblas_status_success)
            return fail("rocsolver_dsyevd failed with status " + std::to_string((int)rs));
    } else {
        // own tridiagonalisation (k_sytrd.hip) + divide & conquer on T (k_stedc.hip) + back-transformation Z = Q C
        DevBuf tau;
        ScratchLease c;
        if (tau.alloc(sizeof(double) * (size_t)n)) return 1;
        if (c.take(0, sizeof(double) * (size_t)n * (size_t)n)) return 1;
        // JXGPU_EIGH_TRACE=1: synchronise and report after every stage (stderr), to locate a failing stage
        const bool trace = getenv("JXGPU_EIGH_TRACE") != nullptr;
        const auto t_begin = std::chrono::steady_clock::now();
        static hipEvent_t ev[8] = {nullptr};
        if (!ev[0])
            for (int q = 0; q < 8; ++q) JX_HIP(hipEventCreate(&ev[q]));
        JX_HIP(hipEventRecord(ev[0], st));
        auto stage_done = [&](const char *what) -> int {
            if (!trace) return 0;
            JX_HIP(hipStreamSynchronize(st));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
            fprintf(stderr, "[jxgpu eigh n=%d] %s done at %.1f ms\n", n, what, ms);
            fflush(stderr);
            return 0;
        };
        // Two-stage reduction (dense -> band -> tridiagonal; k_sy2sb.hip, k_sb2st.hip) from JXGPU_EIGH_TWOSTAGE_MIN rows
        // on: its O(n^3) work is f64-MFMA products instead of one HBM-bound symv per column.  JXGPU_EIGH=onestage keeps
        // the one-stage form; it is also the fallback when a panel of the band reduction cannot be factored, and the
        // form the rank-sharded tridiagonalisation (jxg_eigh_set_dist) uses.
        static const int ts_min = getenv("JXGPU_EIGH_TWOSTAGE_MIN") ? atoi(getenv("JXGPU_EIGH_TWOSTAGE_MIN")) : 1500;
        int drank = 0, dworld = 1;
        sytrd_dist_rank(&drank, &dworld);
        // several ranks: the two-stage path with column-sharded back-transformations when a gather callback is registered
        // (JXGPU_DIST_EIGH_ONESTAGE=1 keeps the rank-sharded one-stage tridiagonalisation instead)
        static const bool dist_onestage = getenv("JXGPU_DIST_EIGH_ONESTAGE") && atoi(getenv("JXGPU_DIST_EIGH_ONESTAGE")) != 0;
        const bool dist_two = dworld > 1 && g_gather.gather != nullptr && !dist_onestage;
        // several ranks: the sharded back-transformations keep the round-2 threshold (what the multi-rank tests cover)
        const int ts_eff = (dworld > 1 && !getenv("JXGPU_EIGH_TWOSTAGE_MIN")) ? std::max(ts_min, 10000) : ts_min;
        bool twostage = n >= ts_eff && n > 4 * sy2sb_bandwidth() && !(mode && strcmp(mode, "onestage") == 0) &&
                        (dist_two || !sytrd_dist_active(n));
        if (mode && strcmp(mode, "twostage") == 0 && n > 2 * sy2sb_bandwidth() + 2) twostage = true;
        // Several ranks: a stage's failure flag (a panel the band reduction could not factor, a bulge-chasing spin that expired
        // -- the latter depends on timing) must send EVERY rank down the same branch, or the ranks' collectives no longer
        // match (one rank in the one-stage fallback, the others in the gather): the flag is compared over the ranks through
        // the agreement callback (checksum = the flag); any disagreement means some rank failed, and all of them fall back.
        // Without a callback a rank cannot coordinate: it fails loudly instead of diverging.
        auto flag_on_any_rank = [&](int local_flag, int *any) -> int {
            *any = local_flag != 0;
            if (!dist_two) return 0;
            if (!g_gather.agree) {
                if (local_flag != 0)
                    return fail("jxg_eigh_f64: a reduction stage failed on this rank and no agreement callback is registered "
                                "(jxg_eigh_set_agree): the ranks cannot take the fallback together");
                return 0;
            }
            const int ag = g_gather.agree(g_gather.agree_user, local_flag != 0 ? 0x9e3779b97f4a7c15ull : 0ull);
            if (ag < 0) return fail("jxg_eigh_f64: the agreement callback failed");
            if (ag == 0) *any = 1;
            return 0;
        };
        // Q1 back-transformation, C-independent part (reflector images, Gram, T^-1, V T of every block: ~40 of its 165 ms at
        // n = 20 000): prepared in line in front of the Q1 products.  Preparing it on a side stream beside the bulge chasing or the
        // divide and conquer was measured in round 4 and moved the 40 ms instead of hiding them; removed.
        std::unique_ptr<OrmtrPlan, void (*)(OrmtrPlan *)> q1_plan(nullptr, ormtr_plan_free);
        DevBuf ts_work, ts_ab, ts_tau2, ts_ctrl, ts_flags, ts_tq;
        ScratchLease ts_v2;
        const int ldab = sb2st_ldab(), ks = sb2st_steps(n);
        if (twostage) {
            if (ts_work.alloc(sizeof(double) * sy2sb_work_doubles(n))) return 1;
            if (ts_ab.alloc(sizeof(double) * (size_t)ldab * n)) return 1;
            if (ts_v2.take(4, sizeof(double) * (size_t)n * n)) return 1;
            if (ts_tau2.alloc(sizeof(double) * (size_t)n * ks)) return 1;
            if (ts_ctrl.alloc(sb2st_ctrl_bytes(n))) return 1;
            if (ts_flags.alloc(sizeof(int) * 4)) return 1;
            // the band reduction overwrites A: keep a copy (in the buffer the divide and conquer fills later) in case a
            // panel cannot be factored
            JX_HIP(hipMemcpyAsync(c.p, d_a, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, st));
            // JXGPU_DIST_EIGH_FORCE: the sharded code path with ONE rank (every block row is its own: same launches, the
            // collectives through the registered callback -- how the RCCL leg is exercised on a one-GPU box)
            static const bool force_single = getenv("JXGPU_DIST_EIGH_FORCE") && atoi(getenv("JXGPU_DIST_EIGH_FORCE")) != 0;
            const bool shard_band = (dist_two || force_single) && sy2sb_band_dist_active(n);
            g_last_band_sharded = shard_band ? 1 : 0;
            if (sy2sb_lower(st, d_a, n, tau.as<double>(), ts_ab.as<double>(), ldab, ts_work.as<double>(),
                            ts_flags.as<int>(), shard_band))
                return 1;
            int hf[4] = {0, 0, 0, 0};
            JX_HIP(hipMemcpyAsync(hf, ts_flags.p, sizeof(hf), hipMemcpyDeviceToHost, st));
            {
                // the workspace of the Q2 back-transformation (tens of GB at n = 50 000: a hipMalloc of that size takes up to
                // seconds) is allocated here, behind the band reduction's launches, while the device works through them
                size_t fr = 0, tot = 0;
                const size_t need = sizeof(double) * sbback_tq_doubles(n, ks);
                if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > need + ((size_t)8 << 30) && ts_tq.alloc(need)) return 1;
            }
            JX_HIP(hipStreamSynchronize(st));
            int flagged = 0;
            if (flag_on_any_rank(hf[0], &flagged)) return 1;
            if (flagged) {
                if (trace) (void)stage_done("sy2sb (flagged)");
                if (trace) fprintf(stderr, "[jxgpu eigh n=%d] band reduction flagged a panel (code %d): one-stage fallback\n", n, hf[0]);
                JX_HIP(hipMemcpyAsync(d_a, c.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, st));
                ts_tq.release();
                twostage = false;
            }
        }
        if (twostage) {
            JX_HIP(hipEventRecord(ev[1], st));
            if (stage_done("sy2sb")) return 1;
            if (sb2st_chase(st, ts_ab.as<double>(), n, d_w, e.as<double>(), ts_v2.as<double>(), ts_tau2.as<double>(),
                            ts_ctrl.as<int>()))
                return 1;
            int habort = 0;
            JX_HIP(hipMemcpyAsync(&habort, ts_ctrl.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, st));
            JX_HIP(hipStreamSynchronize(st));
            int aborted = 0;
            if (flag_on_any_rank(habort, &aborted)) return 1;
            if (aborted) {
                // a sweep's bounded spin on its predecessor expired (the persistent launch assumes its workgroups are
                // co-resident: a shared device can break that): restore the input from the copy and reduce it in one stage
                if (trace) fprintf(stderr, "[jxgpu eigh n=%d] bulge chasing gave up waiting for a neighbour sweep: one-stage fallback\n", n);
                JX_HIP(hipMemcpyAsync(d_a, c.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, st));
                ts_tq.release();
                twostage = false;
            } else {
                JX_HIP(hipEventRecord(ev[2], st));
                if (stage_done("sb2st")) return 1;
            }
        }
        if (twostage) q1_plan.reset(ormtr_plan_new());
        if (!twostage) {
            if (sytrd_lower(h, st, d_a, n, d_w, e.as<double>(), tau.as<double>())) return 1;
            if (stage_done("sytrd")) return 1;
        }
        const char *sm = getenv("JXGPU_STEDC");
        // own divide-and-conquer merges above 1280 rows, independent halves on concurrent streams (rocSOLVER's
        // dstedc is a chain of latency-bound launches: 27 ms for the two 2500-row halves of n = 5000 run back to
        // back, ~8 ms as four concurrent 1250-row leaves), and the only way past n = 46340;
        // JXGPU_STEDC=rocsolver / split, JXGPU_STEDC_LEAF and JXGPU_STEDC_PAR override
        bool split = n >= 2048;
        if (sm && strcmp(sm, "rocsolver") == 0 && n <= kRocsolverStedcMaxN) split = false;
        if (sm && strcmp(sm, "split") == 0 && n >= 64) split = true;
        int leaf = getenv("JXGPU_STEDC_LEAF") ? atoi(getenv("JXGPU_STEDC_LEAF")) : 1280;
        if (leaf < 2) leaf = 2;
        if (leaf > kRocsolverStedcMaxN) leaf = kRocsolverStedcMaxN;
        std::vector<int> perm;
        rocblas_status rs = rocblas_status_success;
        // Several ranks on the two-stage path: the replicated stages (band reduction -- its collectives hand every rank the
        // same bits --, bulge chasing) must have produced the SAME tridiagonal matrix everywhere before anything is shared out:
        // its (d, e) are hashed and compared over the ranks.  When they agree, the divide and conquer's top-level merge forms
        // only the eigenvector columns this rank back-transforms (JXGPU_DIST_DC_WINDOW=0: all of them) and Q2 / Q1 run on that
        // block; when they differ every rank finishes its own replica unsharded (self-consistent, only slower).
        bool replicas_agree = true;
        g_last_dist_agree = -1;
        if (twostage && dist_two && split && g_gather.agree) {
            std::vector<double> hw((size_t)2 * n, 0.0);
            JX_HIP(hipMemcpyAsync(hw.data(), d_w, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
            JX_HIP(hipMemcpyAsync(hw.data() + n, e.p, sizeof(double) * (size_t)(n - 1), hipMemcpyDeviceToHost, st));
            JX_HIP(hipStreamSynchronize(st));
            const uint64_t cs = fnv1a(hw.data(), sizeof(double) * hw.size(), 1469598103934665603ull);
            const int ag = g_gather.agree(g_gather.agree_user, cs);
            if (ag < 0) return fail("jxg_eigh_f64: the agreement callback failed");
            replicas_agree = ag != 0;
            g_last_dist_agree = replicas_agree ? 1 : 0;
            if (!replicas_agree && trace)
                fprintf(stderr, "[jxgpu eigh n=%d] the ranks' tridiagonal matrices differ: every rank finishes its own replica\n", n);
        }
        bool shard_cols = twostage && dist_two && split && replicas_agree;
        const int sh_r0 = (int)((int64_t)n * drank / dworld), sh_r1 = (int)((int64_t)n * (drank + 1) / dworld);
        static const bool dc_window = !(getenv("JXGPU_DIST_DC_WINDOW") && atoi(getenv("JXGPU_DIST_DC_WINDOW")) == 0);
        bool dc_windowed = shard_cols && dc_window && sh_r1 > sh_r0;
        g_last_dc_windowed = (dc_windowed && split) ? 1 : 0;
        if (split) {
            // (d, e) as they enter the divide and conquer: kept on the host when the ranks are about to share columns out, so
            // that a replica which fails the second comparison below can redo the stage without the window
            std::vector<double> de_keep;
            if (shard_cols && g_gather.agree) {
                de_keep.resize((size_t)2 * n);
                JX_HIP(hipMemcpyAsync(de_keep.data(), d_w, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
                JX_HIP(hipMemcpyAsync(de_keep.data() + n, e.p, sizeof(double) * (size_t)(n - 1), hipMemcpyDeviceToHost, st));
                JX_HIP(hipStreamSynchronize(st));
            }
            if (stedc_split(h, st, n, d_w, e.as<double>(), c.as<double>(), leaf, perm, dc_windowed ? sh_r0 : 0,
                            dc_windowed ? sh_r1 : 0))
                return 1;
            if (shard_cols && g_gather.agree) {
                // SECOND comparison: the divide and conquer itself ran replicated, and from here on a rank
                // forms only the columns whose eigenvalues rank inside ITS window -- if the replicas' results differed, the
                // windows would overlap or leave gaps and the gathered matrix would be silently wrong.  Every rank holds all n
                // eigenvalues in ascending order (the rank of an eigenvalue is what assigns its column to a window): they are
                // hashed and compared; the window's own column list differs per rank by construction and is not part of it.
                std::vector<double> hw2((size_t)n);
                JX_HIP(hipMemcpyAsync(hw2.data(), d_w, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
                JX_HIP(hipStreamSynchronize(st));
                uint64_t cs2 = fnv1a(hw2.data(), sizeof(double) * hw2.size(), 1469598103934665603ull ^ (uint64_t)n);
                static const int test_rank2 = getenv("JXGPU_DIST_EIGH_TEST_DISAGREE2") ? atoi(getenv("JXGPU_DIST_EIGH_TEST_DISAGREE2")) : -1;
                if (test_rank2 == drank) cs2 ^= 2;       // test hook: this rank pretends its divide and conquer differed
                const int ag2 = g_gather.agree(g_gather.agree_user, cs2);
                if (ag2 < 0) return fail("jxg_eigh_f64: the agreement callback failed");
                if (ag2 == 0) {
                    if (trace)
                        fprintf(stderr, "[jxgpu eigh n=%d] the ranks' divide-and-conquer results differ: every rank finishes its own replica\n", n);
                    g_last_dist_agree = 0;
                    shard_cols = false;
                    if (dc_windowed) {                   // only a window of the columns exists: redo the stage for all of them
                        dc_windowed = false;
                        g_last_dc_windowed = 0;
                        JX_HIP(hipMemcpyAsync(d_w, de_keep.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
                        JX_HIP(hipMemcpyAsync(e.p, de_keep.data() + n, sizeof(double) * (size_t)(n - 1), hipMemcpyHostToDevice, st));
                        if (stedc_split(h, st, n, d_w, e.as<double>(), c.as<double>(), leaf, perm, 0, 0)) return 1;
                    }
                }
            }
        } else {
            rs = rocsolver_dstedc(h, rocblas_evect_tridiagonal, n, d_w, e.as<double>(), c.as<double>(), n,
                                  info.as<rocblas_int>());
            if (rs != rocblas_status_success)
                return fail("rocsolver_dstedc failed with status " + std::to_string((int)rs));
        }
        JX_HIP(hipEventRecord(ev[3], st));
        if (stage_done("dstedc")) return 1;
        bool sharded_rows = false;                       // d_a already holds this rank's rows + the gathered rest
        int q2_cols = n;                                 // eigenvector columns the Q2 kernel of this rank processed
        if (twostage) {
            if (!ts_tq.p && ts_tq.alloc(sizeof(double) * sbback_tq_doubles(n, ks))) return 1;
            const int ncol = n - sy2sb_bandwidth() - 1;
            if (shard_cols) {
                // this rank's eigenvectors only: columns perm[r0 .. r1) of C (the window's own column list when the divide
                // and conquer formed nothing else), gathered into a contiguous (n, nr) block
                const int r0 = sh_r0, r1 = sh_r1;
                const int nr = r1 - r0;
                const int poff = dc_windowed ? 0 : r0;
                q2_cols = nr;
                DevBuf dperm, blk;
                if (dperm.alloc(sizeof(int) * (size_t)(perm.size() + 1))) return 1;
                if (blk.alloc(sizeof(double) * (size_t)n * (size_t)(nr > 0 ? nr : 1))) return 1;
                JX_HIP(hipMemcpyAsync(dperm.p, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice, st));
                if (launch_gather_cols_range(c.as<double>(), dperm.as<int>() + poff, n, nr, blk.as<double>(), st)) return 1;
                if (sbback_apply_q2(st, ts_v2.as<double>(), ts_tau2.as<double>(), n, ks, blk.as<double>(), nr,
                                    ts_tq.as<double>(), ev[6], ev[7]))
                    return 1;
                JX_HIP(hipEventRecord(ev[4], st));
                if (stage_done("Q2 back-transformation (this rank's columns)")) return 1;
                if (ormtr_prepare(st, d_a, n, sy2sb_bandwidth(), ncol, tau.as<double>(), *q1_plan)) return 1;
                if (ormtr_apply(st, *q1_plan, blk.as<double>(), nr)) return 1;
                // column j of the block = eigenvector r0 + j = row r0 + j of the row-major result
                JX_HIP(hipMemcpyAsync(d_a + (size_t)r0 * n, blk.p, sizeof(double) * (size_t)n * nr, hipMemcpyDeviceToDevice, st));
                JX_HIP(hipStreamSynchronize(st));
                if (g_gather.gather(g_gather.user)) return fail("jxg_eigh_f64: the gather callback failed");
                sharded_rows = true;
            } else {
                if (sbback_apply_q2(st, ts_v2.as<double>(), ts_tau2.as<double>(), n, ks, c.as<double>(), n, ts_tq.as<double>(),
                                    ev[6], ev[7]))
                    return 1;
                JX_HIP(hipEventRecord(ev[4], st));
                if (stage_done("Q2 back-transformation")) return 1;
                if (ormtr_prepare(st, d_a, n, sy2sb_bandwidth(), ncol, tau.as<double>(), *q1_plan)) return 1;
                if (ormtr_apply(st, *q1_plan, c.as<double>(), n)) return 1;
            }
        } else {
            const char *om = getenv("JXGPU_ORMTR");
            if (om && strcmp(om, "rocsolver") == 0) {
                rs = rocsolver_dormtr(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, n, n, d_a, n,
                                      tau.as<double>(), c.as<double>(), n);
                if (rs != rocblas_status_success)
                    return fail("rocsolver_dormtr failed with status " + std::to_string((int)rs));
            } else {
                if (ormtr_lower(st, d_a, n, tau.as<double>(), c.as<double>())) return 1;
            }
        }
        JX_HIP(hipEventRecord(ev[5], st));
        if (stage_done("dormtr")) return 1;
        g_last_ms[10] = twostage ? 1.f : 0.f;
        {
            JX_HIP(hipEventSynchronize(ev[5]));
            float ms = 0.f;
            if (twostage) {
                JX_HIP(hipEventElapsedTime(&ms, ev[0], ev[1])); g_last_ms[6] = ms;
                JX_HIP(hipEventElapsedTime(&ms, ev[1], ev[2])); g_last_ms[7] = ms;
                JX_HIP(hipEventElapsedTime(&ms, ev[2], ev[3])); g_last_ms[8] = ms;
                JX_HIP(hipEventElapsedTime(&ms, ev[4], ev[5])); g_last_ms[9] = ms;
                JX_HIP(hipEventElapsedTime(&ms, ev[6], ev[7])); g_last_ms[4] = ms;
                // algorithmic flops of C <- Q2 C: 4 * (reflector length) * n per reflector, lengths ~ 64, n (n - 1) / 2 / 64 of them
                double refl = 0.0;
                for (int sidx = 0; sidx < n - 2; ++sidx) refl += (double)(n - 1 - sidx);
                g_last_ms[5] = (float)(4.0 * refl * (double)q2_cols / 1e9);
            }
        }
        if (sharded_rows) {
            JX_HIP(hipStreamSynchronize(st));
        } else if (split) {
            DevBuf dperm;
            if (dperm.alloc(sizeof(int) * (size_t)n)) return 1;
            JX_HIP(hipMemcpyAsync(dperm.p, perm.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
            if (launch_gather_cols(c.as<double>(), dperm.as<int>(), n, d_a, st)) return 1;
            JX_HIP(hipStreamSynchronize(st));
        } else {
            JX_HIP(hipMemcpyAsync(d_a, c.p, sizeof(double) * (size_t)n * (size_t)n, hipMemcpyDeviceToDevice, st));
            JX_HIP(hipStreamSynchronize(st));   // the scratch block `c` is released at the end of this scope
        }
    }
    rocblas_int hinfo = 0;
    JX_HIP(hipMemcpyAsync(&hinfo, info.p, sizeof(hinfo), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    if (hinfo != 0) return fail("rocsolver_dsyevd did not converge (info=" + std::to_string(hinfo) + ")");
    return 0;
}

// Diagnostic entry (tests, timing scripts): the two reduction stages alone.  d_a (n,n) symmetric f64 is overwritten
// (band + stage-1 reflectors); d_d / d_e receive the tridiagonal matrix; d_ab_out (optional, 2 SB x n doubles) the band
// matrix after stage 1; h_flags[0] = stage-1 failure flag, h_flags[1] = stage-2 abort flag.
extern "C" int jxg_sy2st_f64(double *d_a, int n, double *d_d, double *d_e, double *d_ab_out, int *h_flags, void *stream) {
    if (n <= 0) return fail("jxg_sy2st_f64: n must be > 0");
    hipStream_t st = (hipStream_t)stream;
    const int ldab = sb2st_ldab(), ks = sb2st_steps(n);
    DevBuf work, ab, tau, v2, tau2, ctrl, flags;
    if (work.alloc(sizeof(double) * sy2sb_work_doubles(n))) return 1;
    if (ab.alloc(sizeof(double) * (size_t)ldab * n)) return 1;
    if (tau.alloc(sizeof(double) * (size_t)n)) return 1;
    if (v2.alloc(sizeof(double) * (size_t)n * n)) return 1;
    if (tau2.alloc(sizeof(double) * (size_t)n * ks)) return 1;
    if (ctrl.alloc(sb2st_ctrl_bytes(n))) return 1;
    if (flags.alloc(sizeof(int) * 4)) return 1;
    const bool trace = getenv("JXGPU_EIGH_TRACE") != nullptr;
    if (trace) JX_HIP(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    if (sy2sb_lower(st, d_a, n, tau.as<double>(), ab.as<double>(), ldab, work.as<double>(), flags.as<int>())) return 1;
    if (trace) {
        JX_HIP(hipStreamSynchronize(st));
        fprintf(stderr, "[jxgpu sy2st n=%d] band reduction %.2f ms\n", n,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (d_ab_out)
        JX_HIP(hipMemcpyAsync(d_ab_out, ab.p, sizeof(double) * (size_t)ldab * n, hipMemcpyDeviceToDevice, st));
    if (sb2st_chase(st, ab.as<double>(), n, d_d, d_e, v2.as<double>(), tau2.as<double>(), ctrl.as<int>())) return 1;
    if (trace) {
        JX_HIP(hipStreamSynchronize(st));
        fprintf(stderr, "[jxgpu sy2st n=%d] bulge chasing %.2f ms\n", n,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    }
    int hf[4] = {0, 0, 0, 0}, habort = 0;
    JX_HIP(hipMemcpyAsync(hf, flags.p, sizeof(hf), hipMemcpyDeviceToHost, st));
    JX_HIP(hipMemcpyAsync(&habort, ctrl.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    if (h_flags) {
        h_flags[0] = hf[0];
        h_flags[1] = habort;
    }
    return 0;
}
