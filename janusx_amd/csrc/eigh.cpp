// Symmetric eigendecomposition on the device, replacing LAPACK dsyevd/dsyevr behind src/math/eigh.rs:1422-1528
// (`symmetric_eigh_f64_row_major_with_driver`): own tridiagonalisation (k_sytrd.hip), divide and conquer (k_stedc.hip)
// and back-transformation (k_ormtr.hip); rocSOLVER dsyevd below n = 256 or with JXGPU_EIGH=rocsolver.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
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
                std::vector<int> &h_perm);
int launch_gather_cols(const double *src, const int *d_perm, int n, double *dst, hipStream_t st);
int sytrd_set_dist(int rank, int world, int (*allreduce)(void *), void *user, double *staging, int64_t staging_doubles,
                   int min_n);
// k_ormtr.hip: C <- Q C with wide compact-WY blocks (dormtr left / lower / no-transpose)
int ormtr_lower(rocblas_handle h, hipStream_t st, const double *d_a, int n, const double *d_tau, double *d_c);
constexpr int kRocsolverStedcMaxN = 46340;   // n^2 < 2^31: rocSOLVER 7.2 dstedc faults above (measured at n = 50000)

static rocblas_handle g_handle = nullptr;
static std::mutex g_handle_mu;

static rocblas_handle get_handle() {
    std::lock_guard<std::mutex> lk(g_handle_mu);
    if (!g_handle) {
        if (rocblas_create_handle(&g_handle) != rocblas_status_success) g_handle = nullptr;
    }
    return g_handle;
}
}  // namespace jx

using namespace jx;

// Node-level distribution of the tridiagonalisation's symv (k_sytrd.hip): every rank calls jxg_eigh_f64 on the same
// matrix; from `min_n` rows on each rank streams 1 / world of the tiles per column and `allreduce(user)` has to sum the
// `jxg_eigh_dist_staging_doubles(n)` doubles at `d_staging` over the ranks on the stream passed to jxg_eigh_f64.
extern "C" int64_t jxg_eigh_dist_staging_doubles(int n) { return (int64_t)n + 32 * 16 + 2 * 64; }

extern "C" int jxg_eigh_set_dist(int rank, int world, int (*allreduce)(void *), void *user, double *d_staging,
                                 int64_t staging_doubles, int min_n) {
    return sytrd_set_dist(rank, world, allreduce, user, d_staging, staging_doubles, min_n);
}

// d_a: (n,n) symmetric, f64. On return row j of d_a (row-major) = eigenvector j (= column j of the
// column-major LAPACK result), eigenvalues ascending in d_w.
extern "C" int jxg_eigh_f64(double *d_a, int n, double ridge, double *d_w, void *stream) {
    if (n <= 0) return fail("jxg_eigh_f64: n must be > 0");
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
        auto stage_done = [&](const char *what) -> int {
            if (!trace) return 0;
            JX_HIP(hipStreamSynchronize(st));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
            fprintf(stderr, "[jxgpu eigh n=%d] %s done at %.1f ms\n", n, what, ms);
            fflush(stderr);
            return 0;
        };
        if (sytrd_lower(h, st, d_a, n, d_w, e.as<double>(), tau.as<double>())) return 1;
        if (stage_done("sytrd")) return 1;
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
        if (split) {
            if (stedc_split(h, st, n, d_w, e.as<double>(), c.as<double>(), leaf, perm)) return 1;
        } else {
            rs = rocsolver_dstedc(h, rocblas_evect_tridiagonal, n, d_w, e.as<double>(), c.as<double>(), n,
                                  info.as<rocblas_int>());
            if (rs != rocblas_status_success)
                return fail("rocsolver_dstedc failed with status " + std::to_string((int)rs));
        }
        if (stage_done("dstedc")) return 1;
        const char *om = getenv("JXGPU_ORMTR");
        if (om && strcmp(om, "rocsolver") == 0) {
            rs = rocsolver_dormtr(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none, n, n, d_a, n,
                                  tau.as<double>(), c.as<double>(), n);
            if (rs != rocblas_status_success)
                return fail("rocsolver_dormtr failed with status " + std::to_string((int)rs));
        } else {
            if (ormtr_lower(h, st, d_a, n, tau.as<double>(), c.as<double>())) return 1;
        }
        if (stage_done("dormtr")) return 1;
        if (split) {
            DevBuf dperm;
            if (dperm.alloc(sizeof(int) * (size_t)n)) return 1;
            JX_HIP(hipMemcpyAsync(dperm.p, perm.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
            if (launch_gather_cols(c.as<double>(), dperm.as<int>(), n, d_a, st)) return 1;
            JX_HIP(hipStreamSynchronize(st));
        } else {
            JX_HIP(hipMemcpyAsync(d_a, c.p, sizeof(double) * (size_t)n * (size_t)n, hipMemcpyDeviceToDevice, st));
        }
    }
    rocblas_int hinfo = 0;
    JX_HIP(hipMemcpyAsync(&hinfo, info.p, sizeof(hinfo), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    if (hinfo != 0) return fail("rocsolver_dsyevd did not converge (info=" + std::to_string(hinfo) + ")");
    return 0;
}
