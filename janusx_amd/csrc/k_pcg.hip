// rrBLUP marker effects by preconditioned conjugate gradients over the 2-bit payload.
//
// Reference: `rrblup_pcg_bed` (src/stats/rrblup.rs:3494-4307): solve (Z_c Z_c' + lambda I) beta = Z y_c over the m
// markers, Z (m, n_train) = standardised genotypes of the training samples ((g - 2p) / sqrt(2p(1-p)), missing -> 0,
// `decode_standardized_packed_block_rows_f32_with_plan`, src/math/bedmath.rs:1161-1249), Z_c its row-centred form
// applied implicitly (`RrblupPcgOperator::apply`, rrblup.rs:865-929); Jacobi preconditioner 1 / (ss_j + lambda) and
// right-hand side from one pre-pass (`rrblup_prepare_rhs_diag` :650-845); the iteration is `pcg_solve_into` for
// T = f32 (src/math/pcg.rs:870-949: f32 vectors, f64 dot products).
//
// Here Z is never materialised: both halves of the operator stream the P32 payload (`packed_dot_kernel`,
// `packed_tdot_kernel`, k_gblup.hip: 2 bits per genotype, f64 accumulation), the row sums / sums of squares of the
// pre-pass come from the per-SNP genotype counts, and the m-vectors of the iteration live in HBM as f32 like the
// reference's; only three scalars per iteration cross to the host (the convergence test is the reference's).
#include <math.h>

#include <chrono>
#include <mutex>
#include <vector>

#include "jx_common.h"

namespace jx {

extern float g_last_ms[24];   // [18] iteration loop of the last rrBLUP PCG solve (ms, wall), [19] its iterations, [20] summed
                              // durations of its operator kernels (HIP events), [21] its set-up (ms, wall); [22] / [23]: summed
                              // operator kernel time (ms) and operator applications of the last Haseman-Elston call

constexpr int PCG_T = 256;

__device__ __forceinline__ void pcg_block_add(double v, double *out) {
    __shared__ double sh[PCG_T / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < PCG_T / 64; ++w) t += sh[w];
        if (t != 0.0) unsafeAtomicAdd(out, t);
    }
}

// out += sum a_j b_j (f64 products of f32 values, `PcgScalar::dot_to_f64`)
__global__ __launch_bounds__(PCG_T) void pcg_dot_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                        int64_t n, double *__restrict__ out) {
    double acc = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * PCG_T + threadIdx.x; j < n; j += (int64_t)gridDim.x * PCG_T)
        acc += (double)a[j] * (double)b[j];
    pcg_block_add(acc, out);
}

__global__ __launch_bounds__(PCG_T) void pcg_widen_kernel(const float *__restrict__ a, double *__restrict__ o, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * PCG_T + threadIdx.x;
    if (j < n) o[j] = (double)a[j];
}

// the reference's GEMV output is f32: round the f64 accumulations once
__global__ __launch_bounds__(PCG_T) void pcg_round_kernel(double *__restrict__ a, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * PCG_T + threadIdx.x;
    if (j < n) a[j] = (double)(float)a[j];
}

// ap = f32(Z Z'p) - n_train * mu * mean_dot + lambda * p  (rrblup.rs:902-916);  denom += p . ap
__global__ __launch_bounds__(PCG_T) void pcg_finish_ap_kernel(const double *__restrict__ ap64, const float *__restrict__ p,
                                                              const float *__restrict__ mu, float n_train_f,
                                                              const double *__restrict__ mean_dot, float lambda,
                                                              int64_t n, float *__restrict__ ap,
                                                              double *__restrict__ denom) {
    const float md = (float)mean_dot[0];
    double acc = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * PCG_T + threadIdx.x; j < n; j += (int64_t)gridDim.x * PCG_T) {
        float v = (float)ap64[j];
        v = __fsub_rn(v, __fmul_rn(__fmul_rn(n_train_f, mu[j]), md));
        v = __fadd_rn(v, __fmul_rn(lambda, p[j]));
        ap[j] = v;
        acc += (double)p[j] * (double)v;
    }
    pcg_block_add(acc, denom);
}

// x += alpha p; r -= alpha ap; rr += r . r   (pcg.rs:672-694)
__global__ __launch_bounds__(PCG_T) void pcg_update_xr_kernel(float *__restrict__ x, float *__restrict__ r,
                                                              const float *__restrict__ p, const float *__restrict__ ap,
                                                              float alpha, int64_t n, double *__restrict__ rr) {
    double acc = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * PCG_T + threadIdx.x; j < n; j += (int64_t)gridDim.x * PCG_T) {
        x[j] = __fadd_rn(x[j], __fmul_rn(alpha, p[j]));
        const float rv = __fsub_rn(r[j], __fmul_rn(alpha, ap[j]));
        r[j] = rv;
        acc += (double)rv * (double)rv;
    }
    pcg_block_add(acc, rr);
}

// z = r * inv_diag; rz += r . z   (Jacobi, pcg.rs:223-245)
__global__ __launch_bounds__(PCG_T) void pcg_precond_kernel(const float *__restrict__ r, const float *__restrict__ dinv,
                                                            float *__restrict__ z, int64_t n, double *__restrict__ rz) {
    double acc = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * PCG_T + threadIdx.x; j < n; j += (int64_t)gridDim.x * PCG_T) {
        const float zv = __fmul_rn(r[j], dinv[j]);
        z[j] = zv;
        acc += (double)r[j] * (double)zv;
    }
    pcg_block_add(acc, rz);
}

// p = z + beta p   (pcg.rs:696-707)
__global__ __launch_bounds__(PCG_T) void pcg_update_p_kernel(float *__restrict__ p, const float *__restrict__ z, float beta,
                                                             int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * PCG_T + threadIdx.x;
    if (j < n) p[j] = __fadd_rn(z[j], __fmul_rn(beta, p[j]));
}

static inline unsigned pcg_grid(int64_t n) {
    int64_t g = (n + PCG_T - 1) / PCG_T;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (unsigned)g;
}
static inline unsigned pcg_grid_full(int64_t n) { return (unsigned)((n + PCG_T - 1) / PCG_T); }

}  // namespace jx

using namespace jx;

// value_lut (eff_m, 4) f32: standardised design values by 2-bit code (entry 1, the missing code, must be 0).
// row_indices (eff_m) int64 or NULL (all m_total rows).  out_scalars: [0] converged, [1] iterations, [2] relative
// residual, [3] sum of the centred row sums of squares (k_trace numerator), [4] intercept alpha.
// ---- marker-sharded solve over several ranks (last row) ----------------------------------------------
// The operator Z (Z'p) sums over markers: with the markers dealt over the ranks (contiguous ranges of the kept rows) rank r
// holds p_r, forms its partial Z_r' p_r (an n_train-vector) and ONE all-reduce per iteration completes Z'p on every rank
// (the implicit row-centring term mu'p rides in the same buffer); the three scalars of an iteration (p'Ap, r'r, r'z) are
// all-reduced too, so every rank takes the same step and the same stopping decision.  `allreduce(user)` has to sum the
// first `jx_pcg_dist_count()` doubles of `d_staging` over the ranks in place (torch.distributed: RCCL over xGMI with the
// nccl backend; janusx_amd/dist.py).  world = 1 (the default): nothing changes.
namespace {
struct PcgDist {
    int rank = 0, world = 1;
    int (*allreduce)(void *) = nullptr;
    void *user = nullptr;
    double *staging = nullptr;
    int64_t cap = 0;
    int64_t count = 0;       // doubles of the pending collective
};
PcgDist g_pcg_dist;

// sum `na` doubles at device pointer a (and `nb` at b, optional) over the ranks, in place
int pcg_allreduce_dev(double *a, int64_t na, double *b, int64_t nb, hipStream_t st) {
    PcgDist &D = g_pcg_dist;
    if (D.world <= 1) return 0;
    if (na + nb > D.cap) return fail("jx_pcg_set_dist: staging buffer too small for the solve");
    JX_HIP(hipMemcpyAsync(D.staging, a, sizeof(double) * (size_t)na, hipMemcpyDeviceToDevice, st));
    if (nb > 0) JX_HIP(hipMemcpyAsync(D.staging + na, b, sizeof(double) * (size_t)nb, hipMemcpyDeviceToDevice, st));
    JX_HIP(hipStreamSynchronize(st));
    D.count = na + nb;
    if (D.allreduce(D.user)) return fail("jx_rrblup_pcg_packed: the all-reduce callback failed");
    JX_HIP(hipMemcpyAsync(a, D.staging, sizeof(double) * (size_t)na, hipMemcpyDeviceToDevice, st));
    if (nb > 0) JX_HIP(hipMemcpyAsync(b, D.staging + na, sizeof(double) * (size_t)nb, hipMemcpyDeviceToDevice, st));
    JX_HIP(hipStreamSynchronize(st));
    return 0;
}
// the same for a few host doubles.  `lerr` (this rank failed since the last collective: an allocation, a launch) rides in one
// more slot: every rank learns it in the same collective and all of them leave together -- a rank that returned on its own
// would leave the others blocked in their next all-reduce.  -> 0, 1 (collective failed), 2 (a
// rank reported a failure; the message of the failing rank is its own, the others get a generic one)
int pcg_allreduce_host(double *h, int cnt, hipStream_t st, int lerr = 0) {
    PcgDist &D = g_pcg_dist;
    if (D.world <= 1) return lerr ? 2 : 0;
    if (cnt + 1 > D.cap) return fail("jx_pcg_set_dist: staging buffer too small");
    double buf[8];
    if (cnt > 7) return fail("pcg_allreduce_host: at most 7 scalars");
    for (int i = 0; i < cnt; ++i) buf[i] = lerr ? 0.0 : h[i];
    buf[cnt] = lerr ? 1.0 : 0.0;
    JX_HIP(hipMemcpyAsync(D.staging, buf, sizeof(double) * (size_t)(cnt + 1), hipMemcpyHostToDevice, st));
    JX_HIP(hipStreamSynchronize(st));
    D.count = cnt + 1;
    if (D.allreduce(D.user)) return fail("jx_rrblup_pcg_packed: the all-reduce callback failed");
    JX_HIP(hipMemcpyAsync(buf, D.staging, sizeof(double) * (size_t)(cnt + 1), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < cnt; ++i) h[i] = buf[i];
    if (buf[cnt] != 0.0) {
        if (!lerr) set_error("jx_rrblup_pcg_packed: another rank of the marker-sharded solve failed (" +
                             std::to_string((int)buf[cnt]) + " of " + std::to_string(D.world) + "); all ranks stop");
        return 2;
    }
    return 0;
}
// a collective a failed rank still has to take part in: same count as the healthy ranks use, zeros as its share
int pcg_allreduce_dev_or_zero(double *a, int64_t na, double *b, int64_t nb, hipStream_t st, int lerr) {
    if (g_pcg_dist.world <= 1) return 0;
    if (lerr) {
        (void)hipMemsetAsync(a, 0, sizeof(double) * (size_t)na, st);
        if (nb > 0) (void)hipMemsetAsync(b, 0, sizeof(double) * (size_t)nb, st);
    }
    return pcg_allreduce_dev(a, na, b, nb, st);
}
}  // namespace

extern "C" int jx_pcg_set_dist(int rank, int world, int (*allreduce)(void *), void *user, double *d_staging,
                               int64_t staging_doubles) {
    if (world <= 1 || !allreduce) {
        g_pcg_dist = PcgDist{};
        return 0;
    }
    if (rank < 0 || rank >= world || !d_staging || staging_doubles < 8) return fail("jx_pcg_set_dist: bad arguments");
    g_pcg_dist.rank = rank;
    g_pcg_dist.world = world;
    g_pcg_dist.allreduce = allreduce;
    g_pcg_dist.user = user;
    g_pcg_dist.staging = d_staging;
    g_pcg_dist.cap = staging_doubles;
    return 0;
}
extern "C" int64_t jx_pcg_dist_count(void) { return g_pcg_dist.count; }

// ---- image scope -------------------------------------------------------------------------------------------------------
// `jx gs -rrBLUP -rr-solver pcg` calls he_pcg_bed (lambda) and then rrblup_pcg_bed on the SAME payload, rows and training samples.
// Each of them builds the same two images of the training payload (SNP-major P32 and sample-major T32: 40 GB each at BASELINE
// configs[4]) and frees them again -- and a hipMalloc right behind the hipFree of a 40 GB block stalls for seconds on this stack
// (measured: he_pcg_bed 1.7 s alone, 6.7 s directly behind rrblup_pcg_bed).  Inside a scope (jx_pcg_image_scope(1) ... (0), opened by
// the caller that knows the payload does not change in between) the images of the last (payload pointer, rows, samples) stay in HBM
// and the second call reuses them: no allocation, no repack, no transpose.  Outside a scope nothing is kept.
namespace {
struct PcgImages {
    bool scope = false, valid = false;
    const void *src = nullptr;
    int64_t m_total = 0, eff_m = 0;
    int n_samples = 0, n_train = 0;
    uint64_t h_train = 0, h_rows = 0;
    DevBuf p32, t32;
    void drop() {
        p32.release();
        t32.release();
        valid = false;
    }
};
PcgImages g_img;
std::mutex g_img_mu;
uint64_t img_hash(const void *p, size_t bytes) {
    uint64_t h = 1469598103934665603ull;
    const uint8_t *b = static_cast<const uint8_t *>(p);
    for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}
// -> 0 and *p32 / *t32 (device pointers: borrowed from the scope's cache, or owned by `own_p32` / `own_t32`); `built` = the images
// had to be made (the caller then continues with its own pre-pass as before)
// `key`: the CALLER's payload pointer when the payload already lives in HBM, nullptr for a host payload (its device copy is a
// buffer of this call: the next call's upload usually gets the same address back, so that address identifies nothing -- a host
// payload uses the scope's buffers but never hits, and leaves the cache invalid)
int pcg_images(const void *key, const uint8_t *d_raw, int64_t bps, int64_t m_total, int n_samples, const int64_t *row_indices,
               const int64_t *d_rowidx, int64_t eff_m, const int64_t *train_idx, const int32_t *d_train32, int n_train,
               DevBuf &own_p32, DevBuf &own_t32, const uint8_t **p32, const uint8_t **t32, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_img_mu);
    const uint64_t ht = img_hash(train_idx, sizeof(int64_t) * (size_t)n_train);
    const uint64_t hr = row_indices ? img_hash(row_indices, sizeof(int64_t) * (size_t)eff_m) : 0ull;
    PcgImages &G = g_img;
    if (key && G.scope && G.valid && G.src == key && G.m_total == m_total && G.eff_m == eff_m && G.n_samples == n_samples &&
        G.n_train == n_train && G.h_train == ht && G.h_rows == hr) {
        *p32 = G.p32.as<uint8_t>();
        *t32 = G.t32.as<uint8_t>();
        return 0;
    }
    const size_t b_p32 = (size_t)num_tiles(n_train) * (size_t)eff_m * 32, b_t32 = (size_t)jxg_t32_bytes(n_train, (int)eff_m);
    DevBuf &dp = G.scope ? G.p32 : own_p32, &dt = G.scope ? G.t32 : own_t32;
    if (G.scope) G.valid = false;
    if (dp.bytes < b_p32 && dp.alloc(b_p32)) return 1;
    if (dt.bytes < b_t32 && dt.alloc(b_t32)) return 1;
    if (jxg_repack_p32(d_raw, bps, n_samples, m_total, d_train32, n_train, d_rowidx, eff_m, dp.as<uint8_t>(), st)) return 1;
    if (jxg_p32_transpose(dp.as<uint8_t>(), eff_m, n_train, nullptr, (int)eff_m, dt.as<uint8_t>(), st)) return 1;
    if (G.scope && key) {
        G.valid = true;
        G.src = key;
        G.m_total = m_total;
        G.eff_m = eff_m;
        G.n_samples = n_samples;
        G.n_train = n_train;
        G.h_train = ht;
        G.h_rows = hr;
    }
    *p32 = dp.as<uint8_t>();
    *t32 = dt.as<uint8_t>();
    return 0;
}
// HBM the next pcg_images call of this shape still has to allocate: nothing on a cache hit, the shortfall of the scope's buffers
// inside a scope, both images otherwise
double pcg_images_pending_bytes(int n_train, int64_t eff_m) {
    std::lock_guard<std::mutex> lk(g_img_mu);
    const size_t b_p32 = (size_t)num_tiles(n_train) * (size_t)eff_m * 32, b_t32 = (size_t)jxg_t32_bytes(n_train, (int)eff_m);
    if (!g_img.scope) return (double)b_p32 + (double)b_t32;
    return (double)(g_img.p32.bytes < b_p32 ? b_p32 : 0) + (double)(g_img.t32.bytes < b_t32 ? b_t32 : 0);
}
// HIP events of ONE call (created for the device that is current now, destroyed with the call): concurrent solves do not share
// timing events
struct EvSet {
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    int make(int k) {
        for (int i = 0; i < k; ++i) JX_HIP(hipEventCreate(&e[i]));
        return 0;
    }
    ~EvSet() {
        for (auto &x : e)
            if (x) (void)hipEventDestroy(x);
    }
    hipEvent_t operator[](int i) const { return e[i]; }
};
}  // namespace

extern "C" int jx_pcg_image_scope(int on) {
    std::lock_guard<std::mutex> lk(g_img_mu);
    g_img.drop();
    g_img.scope = on != 0;
    return 0;
}

extern "C" int jx_rrblup_pcg_packed(const uint8_t *packed, int64_t m_total, int n_samples, const int64_t *row_indices,
                                    int64_t eff_m, const float *value_lut, const int64_t *train_idx, int n_train,
                                    const double *y_train, const int64_t *test_idx, int n_test, double lambda_value,
                                    double tol, int max_iter, float *out_beta, double *out_pred_train,
                                    double *out_pred_test, double *out_scalars) {
    // Argument checks first: they see the same values on every rank of a marker-sharded solve (or fail on all of them).
    if (n_samples <= 0) return fail("n_samples must be > 0");
    if (m_total <= 0 || eff_m <= 0) return fail("No SNP rows found in BED input.");
    if (n_train <= 0) return fail("train_sample_indices must not be empty.");
    if (max_iter <= 0) return fail("max_iter must be > 0");
    if (!(isfinite(tol) && tol > 0.0)) return fail("tol must be finite and > 0");
    if (!isfinite(lambda_value) || lambda_value < 0.0) return fail("lambda_value must be finite and >= 0");
    if (eff_m > 0x7fffffffLL) return fail("too many markers for one call");
    const int64_t bps = ((int64_t)n_samples + 3) / 4;
    std::vector<int32_t> tr32(n_train), te32(n_test > 0 ? n_test : 0);
    for (int i = 0; i < n_train; ++i) {
        if (train_idx[i] < 0 || train_idx[i] >= n_samples) return fail("train_sample_indices out of range");
        tr32[i] = (int32_t)train_idx[i];
    }
    for (int i = 0; i < n_test; ++i) {
        if (test_idx[i] < 0 || test_idx[i] >= n_samples) return fail("test_sample_indices out of range");
        te32[i] = (int32_t)test_idx[i];
    }
    double y_mean = 0.0;
    for (int i = 0; i < n_train; ++i) {
        if (!isfinite(y_train[i])) return fail("y_train contains non-finite values.");
        y_mean += y_train[i];
    }
    y_mean /= (double)n_train;
    const float lambda_use = (float)(lambda_value > 1e-8 ? lambda_value : 1e-8);
    const double tol_use = tol > 1e-12 ? tol : 1e-12;
    const bool multi = g_pcg_dist.world > 1;
    // stage timers of the last solve (jxg_last_kernel_ms 18 - 21): wall time of the set-up (images, pre-pass) and of the
    // iteration loop, and the summed HIP-event durations of the two streaming operator kernels (Z'p on the sample-major image,
    // Z (Z'p) on the SNP-major one): the kernels an HBM roofline of this route is quoted on
    EvSet ev;
    if (ev.make(4)) return 1;
    double op_ms = 0.0;
    const auto wall0 = std::chrono::steady_clock::now();

    // From here on a failure may be this rank's alone (its shard's row list, an allocation, a launch): with several ranks it is
    // carried as `lerr` into the next collective, where every rank learns it and all of them leave together.
    hipStream_t st = nullptr;
    DevBuf raw, didx, drow, p32, dlut, dcnt, t32, dwork;
    DevBuf dmu, ddinv, dx, dr, dz, dp, dap, dv64m, dv64n, dsc;
    const uint8_t *d_raw = packed;                 // a payload that already lives in HBM is used in place
    const int64_t *d_rowidx = nullptr;
    std::vector<float> mu(eff_m);
    const size_t mb = sizeof(float) * (size_t)eff_m;
    float *x = nullptr, *r = nullptr, *z = nullptr, *p = nullptr, *ap = nullptr;
    double *v64m = nullptr, *v64n = nullptr, *sc = nullptr;
    const uint8_t *P = nullptr, *T32 = nullptr;     // SNP-major / sample-major image of the training payload
    const float *L = nullptr;
    const unsigned gm = pcg_grid(eff_m), gmf = pcg_grid_full(eff_m), gnf = pcg_grid_full(n_train);
    double sum_ss = 0.0, bb = 0.0, rz_old = 0.0;

    auto scalar = [&](int k, double &out) -> int {
        JX_HIP(hipMemcpyAsync(&out, sc + k, sizeof(double), hipMemcpyDeviceToHost, st));
        JX_HIP(hipStreamSynchronize(st));
        return 0;
    };

    static const bool pcg_trace = getenv("JXGPU_PCG_TRACE") != nullptr;
    auto tmark = [&](const char *what) {
        if (!pcg_trace) return;
        (void)hipDeviceSynchronize();
        fprintf(stderr, "[jxgpu pcg] %-28s %8.1f ms\n", what,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
    };
    auto setup = [&]() -> int {
        if (row_indices)
            for (int64_t j = 0; j < eff_m; ++j)
                if (row_indices[j] < 0 || row_indices[j] >= m_total) return fail("site_keep row index out of range");
        const bool on_device = is_device_ptr(packed);
        const int nt = num_tiles(n_train);
        {
            size_t fr = 0, tot = 0;
            // images a surrounding scope already holds (jx_he_traces_packed before this call, the CLI's default order) are
            // reused, not allocated again: only what is still to be allocated counts
            (void)nt;
            const double need = (on_device ? 0.0 : (double)m_total * (double)bps) + pcg_images_pending_bytes(n_train, eff_m) +
                                64.0 * (double)eff_m;
            if (hipMemGetInfo(&fr, &tot) == hipSuccess && need > 0.97 * (double)fr)
                return fail("rrBLUP PCG: the payload images of " + std::to_string((long long)eff_m) + " markers x " +
                            std::to_string(n_train) + " training samples need " + std::to_string((long long)(need / 1048576.0)) +
                            " MiB of HBM, " + std::to_string((long long)(fr >> 20)) +
                            " MiB are free (deal the markers over more ranks: dist.enable_distributed_pcg)");
        }
        // Every allocation of the solve BEFORE the first kernel is launched: a hipMalloc issued while kernels are in flight takes
        // hundreds of milliseconds to seconds on this stack (measured at BASELINE configs[4]: set-up 2.3 - 3.6 s with the
        // allocations interleaved with the repack / transpose launches, 0.33 s with the device idle at every allocation)
        if (!on_device && raw.alloc((size_t)(m_total * bps))) return 1;
        if (dwork.alloc(16 * (size_t)eff_m + 16) || dlut.alloc(sizeof(float) * 4 * (size_t)eff_m) ||
            dcnt.alloc(sizeof(int32_t) * 3 * (size_t)eff_m) || dmu.alloc(mb) || ddinv.alloc(mb) || dx.alloc(mb) || dr.alloc(mb) ||
            dz.alloc(mb) || dp.alloc(mb) || dap.alloc(mb) || dv64m.alloc(sizeof(double) * (size_t)eff_m) ||
            dv64n.alloc(sizeof(double) * (size_t)(n_train > n_test ? n_train : n_test)) || dsc.alloc(sizeof(double) * 8))
            return 1;
        if (!on_device) {
            JX_HIP(hipMemcpy(raw.p, packed, (size_t)(m_total * bps), hipMemcpyHostToDevice));
            d_raw = raw.as<uint8_t>();
        }
        if (didx.alloc(sizeof(int32_t) * (size_t)n_train)) return 1;
        JX_HIP(hipMemcpy(didx.p, tr32.data(), sizeof(int32_t) * (size_t)n_train, hipMemcpyHostToDevice));
        if (row_indices) {
            if (drow.alloc(sizeof(int64_t) * (size_t)eff_m)) return 1;
            JX_HIP(hipMemcpy(drow.p, row_indices, sizeof(int64_t) * (size_t)eff_m, hipMemcpyHostToDevice));
            d_rowidx = drow.as<int64_t>();
        }
        tmark("allocations + small uploads");
        // the two images of the training payload: made here (allocated with the device idle), or borrowed from the caller's scope
        if (pcg_images(on_device ? (const void *)packed : nullptr, d_raw, bps, m_total, n_samples, row_indices, d_rowidx, eff_m,
                       train_idx, didx.as<int32_t>(), n_train, p32, t32, &P, &T32, st))
            return 1;
        tmark("images (repack + transpose, or the scope's)");
        JX_HIP(hipMemcpy(dlut.p, value_lut, sizeof(float) * 4 * (size_t)eff_m, hipMemcpyHostToDevice));
        if (jxg_row_counts_p32(P, eff_m, n_train, dcnt.as<int32_t>(), st)) return 1;
        tmark("row counts");
        std::vector<int32_t> cnt(3 * (size_t)eff_m);
        JX_HIP(hipMemcpy(cnt.data(), dcnt.p, sizeof(int32_t) * 3 * (size_t)eff_m, hipMemcpyDeviceToHost));
        tmark("counts download");

        // pre-pass (row_major_block_prepare_rhs_diag_f32, rrblup.rs:396-466): the f64 sums over the f32 row values are
        // count-weighted sums of the three genotype values
        std::vector<float> dinv(eff_m);
        for (int64_t j = 0; j < eff_m; ++j) {
            const double c1 = cnt[3 * j + 1], c2 = cnt[3 * j + 2];
            const double c0 = (double)n_train - (double)cnt[3 * j] - c1 - c2;
            const double v0 = value_lut[4 * j], v2 = value_lut[4 * j + 2], v3 = value_lut[4 * j + 3];
            const double sum = c0 * v0 + c1 * v2 + c2 * v3;
            const double mean = sum / (double)n_train;
            const double ss_raw = c0 * v0 * v0 + c1 * v2 * v2 + c2 * v3 * v3;
            double ss = ss_raw - (double)n_train * mean * mean;
            if (!(ss > 0.0)) ss = 0.0;
            sum_ss += ss;
            mu[j] = (float)mean;
            float d = (float)ss + lambda_use;
            if (!(d > 1e-12f)) d = 1e-12f;
            dinv[j] = 1.0f / d;
        }
        JX_HIP(hipMemcpy(dmu.p, mu.data(), mb, hipMemcpyHostToDevice));
        JX_HIP(hipMemcpy(ddinv.p, dinv.data(), mb, hipMemcpyHostToDevice));
        x = dx.as<float>(), r = dr.as<float>(), z = dz.as<float>(), p = dp.as<float>(), ap = dap.as<float>();
        v64m = dv64m.as<double>(), v64n = dv64n.as<double>(), sc = dsc.as<double>();
        L = dlut.as<float>();
        // b = Z y_c (f32 GEMV in the reference; f64 accumulation rounded once here)
        {
            std::vector<double> yc(n_train);
            for (int i = 0; i < n_train; ++i) yc[i] = (double)(float)(y_train[i] - y_mean);
            JX_HIP(hipMemcpy(v64n, yc.data(), sizeof(double) * (size_t)n_train, hipMemcpyHostToDevice));
            if (jxg_packed_tdot(P, eff_m, n_train, nullptr, (int)eff_m, L, v64n, v64m, st)) return 1;
        }
        // r = b (f32), x = 0, z = M^-1 r, p = z
        std::vector<double> b64(eff_m);
        JX_HIP(hipMemcpy(b64.data(), v64m, sizeof(double) * (size_t)eff_m, hipMemcpyDeviceToHost));
        std::vector<float> b32(eff_m);
        for (int64_t j = 0; j < eff_m; ++j) {
            b32[j] = (float)b64[j];
            bb += (double)b32[j] * (double)b32[j];
        }
        JX_HIP(hipMemcpy(r, b32.data(), mb, hipMemcpyHostToDevice));
        JX_HIP(hipMemsetAsync(x, 0, mb, st));
        JX_HIP(hipMemsetAsync(sc, 0, sizeof(double) * 8, st));
        hipLaunchKernelGGL(pcg_precond_kernel, dim3(gm), dim3(PCG_T), 0, st, r, ddinv.as<float>(), z, eff_m, sc + 0);
        JX_LAUNCH_CHECK();
        JX_HIP(hipMemcpyAsync(p, z, mb, hipMemcpyDeviceToDevice, st));
        if (scalar(0, rz_old)) return 1;
        tmark("pre-pass, b, first z");
        return 0;
    };
    int lerr = setup() ? 1 : 0;
    // test hook: JXGPU_PCG_TEST_FAIL="<rank>:<where>" makes that rank fail on its own -- where = 0: in the set-up, k > 0: in the
    // first half of iteration k -- so that the agreed failure (every rank leaves in the same collective) can be exercised
    int test_fail_it = -1;
    if (const char *tf = getenv("JXGPU_PCG_TEST_FAIL")) {
        int r = -1, w = -1;
        if (sscanf(tf, "%d:%d", &r, &w) == 2 && r == g_pcg_dist.rank && multi) {
            if (w == 0 && !lerr) lerr = fail("jx_rrblup_pcg_packed: test hook, this rank fails in its set-up");
            test_fail_it = w;
        }
    }
    const auto wall1 = std::chrono::steady_clock::now();
    {
        // markers sharded over ranks: |b|^2, the trace and r'z over all markers -- and every rank's setup verdict
        double three[3] = {bb, sum_ss, rz_old};
        if (pcg_allreduce_host(three, 3, st, lerr)) return 1;
        bb = three[0], sum_ss = three[1], rz_old = three[2];
    }
    if (!isfinite(bb)) return fail("PCG invalid RHS norm.");
    const double bnorm = sqrt(bb);
    const double denom_b = bnorm > 1e-12 ? bnorm : 1e-12;
    double rel_res = bnorm / denom_b;   // r = b
    if (rel_res < 0.0) rel_res = 0.0;
    bool converged = false;
    int iters = 0;
    const double tiny_use = 1e-20;
    if (isfinite(rel_res) && rel_res <= tol_use) {
        converged = true;
    } else {
        for (int it = 0; it < max_iter; ++it) {
            // ap = A p
            auto half1 = [&]() -> int {
                hipLaunchKernelGGL(pcg_widen_kernel, dim3(gmf), dim3(PCG_T), 0, st, p, v64m, eff_m);
                JX_LAUNCH_CHECK();
                JX_HIP(hipEventRecord(ev[0], st));
                if (jxg_packed_dot_t32(T32, n_train, (int)eff_m, L, v64m, dwork.p, v64n, st)) return 1;   // Z'p
                JX_HIP(hipEventRecord(ev[1], st));
                JX_HIP(hipMemsetAsync(sc + 1, 0, 2 * sizeof(double), st));
                hipLaunchKernelGGL(pcg_dot_kernel, dim3(gm), dim3(PCG_T), 0, st, dmu.as<float>(), p, eff_m, sc + 1);    // mu'p
                JX_LAUNCH_CHECK();
                return 0;
            };
            if (!lerr) lerr = half1() ? 1 : 0;
            if (!lerr && test_fail_it == it + 1) lerr = fail("jx_rrblup_pcg_packed: test hook, this rank fails in iteration " + std::to_string(it + 1));
            if (pcg_allreduce_dev_or_zero(v64n, n_train, sc + 1, 1, st, lerr)) return 1;      // this rank's markers -> all markers
            double denom = 0.0;
            auto half2 = [&]() -> int {
                hipLaunchKernelGGL(pcg_round_kernel, dim3(gnf), dim3(PCG_T), 0, st, v64n, (int64_t)n_train);
                JX_LAUNCH_CHECK();
                JX_HIP(hipEventRecord(ev[2], st));
                if (jxg_packed_tdot_f32(P, eff_m, n_train, nullptr, (int)eff_m, L, v64n, v64m, st)) return 1;  // Z (Z'p)
                JX_HIP(hipEventRecord(ev[3], st));
                hipLaunchKernelGGL(pcg_finish_ap_kernel, dim3(gm), dim3(PCG_T), 0, st, v64m, p, dmu.as<float>(),
                                   (float)n_train, sc + 1, lambda_use, eff_m, ap, sc + 2);
                JX_LAUNCH_CHECK();
                if (scalar(2, denom)) return 1;
                float a = 0.f, b = 0.f;
                if (hipEventElapsedTime(&a, ev[0], ev[1]) == hipSuccess && hipEventElapsedTime(&b, ev[2], ev[3]) == hipSuccess)
                    op_ms += (double)a + (double)b;
                return 0;
            };
            if (!lerr) lerr = half2() ? 1 : 0;
            if (pcg_allreduce_host(&denom, 1, st, lerr)) return 1;
            if (!isfinite(denom) || denom <= tiny_use) break;
            const double alpha = rz_old / denom;
            double two[2] = {0.0, 0.0};     // r'r, r'z
            auto half3 = [&]() -> int {
                JX_HIP(hipMemsetAsync(sc + 3, 0, sizeof(double), st));
                hipLaunchKernelGGL(pcg_update_xr_kernel, dim3(gm), dim3(PCG_T), 0, st, x, r, p, ap, (float)alpha, eff_m,
                                   sc + 3);
                JX_LAUNCH_CHECK();
                JX_HIP(hipMemsetAsync(sc + 0, 0, sizeof(double), st));
                hipLaunchKernelGGL(pcg_precond_kernel, dim3(gm), dim3(PCG_T), 0, st, r, ddinv.as<float>(), z, eff_m, sc + 0);
                JX_LAUNCH_CHECK();
                JX_HIP(hipMemcpyAsync(&two[0], sc + 3, sizeof(double), hipMemcpyDeviceToHost, st));
                return scalar(0, two[1]);
            };
            lerr = half3() ? 1 : 0;
            if (pcg_allreduce_host(two, 2, st, lerr)) return 1;
            const double rr = two[0], rz_new = two[1];
            rel_res = sqrt(rr) / denom_b;
            if (rel_res < 0.0) rel_res = 0.0;
            iters = it + 1;
            if (isfinite(rel_res) && rel_res <= tol_use) {
                converged = true;
                break;
            }
            if (!isfinite(rz_new) || rz_new <= tiny_use) break;
            const double beta = rz_new / (rz_old > tiny_use ? rz_old : tiny_use);
            hipLaunchKernelGGL(pcg_update_p_kernel, dim3(gmf), dim3(PCG_T), 0, st, p, z, (float)beta, eff_m);
            if (hipGetLastError() != hipSuccess) lerr = fail("kernel launch failed: pcg_update_p_kernel");   // reported in the next collective
            rz_old = rz_new;
        }
    }
    g_last_ms[18] = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall1).count();
    g_last_ms[19] = (float)iters;
    g_last_ms[20] = (float)op_ms;
    g_last_ms[21] = (float)std::chrono::duration<double, std::milli>(wall1 - wall0).count();
    float acc = 0.0f;   // `.map(mu * b).sum::<f32>()`, sequential (rrblup.rs:4057-4062)
    if (!lerr) {
        if (hipMemcpy(out_beta, x, mb, hipMemcpyDeviceToHost) != hipSuccess) lerr = fail("download of the marker effects failed");
        for (int64_t j = 0; j < eff_m && !lerr; ++j) {
            volatile float t = mu[j] * out_beta[j];
            acc = acc + t;
        }
    }
    if (multi) {      // sum over all markers (the single-rank form keeps the reference's sequential f32 sum)
        double a64 = (double)acc;
        if (pcg_allreduce_host(&a64, 1, st, lerr)) return 1;
        acc = (float)a64;
    } else if (lerr) {
        return 1;
    }
    const float alpha_use = (float)y_mean - acc;

    // predictions: alpha + Z_samples' beta (pcg_x_mul_samples, f32 output)
    auto pred_train = [&]() -> int {
        hipLaunchKernelGGL(pcg_widen_kernel, dim3(gmf), dim3(PCG_T), 0, st, x, v64m, eff_m);
        JX_LAUNCH_CHECK();
        if (out_pred_train && jxg_packed_dot(P, eff_m, n_train, nullptr, (int)eff_m, L, v64m, v64n, st)) return 1;
        return 0;
    };
    lerr = pred_train() ? 1 : 0;
    if (out_pred_train) {
        if (pcg_allreduce_dev_or_zero(v64n, n_train, nullptr, 0, st, lerr)) return 1;
        if (!lerr) {
            JX_HIP(hipStreamSynchronize(st));
            JX_HIP(hipMemcpy(out_pred_train, v64n, sizeof(double) * (size_t)n_train, hipMemcpyDeviceToHost));
            for (int i = 0; i < n_train; ++i) out_pred_train[i] = (double)((float)out_pred_train[i] + alpha_use);
        }
    }
    if (n_test > 0 && out_pred_test) {
        DevBuf dte, p32t;
        auto pred_test = [&]() -> int {
            // the training images are no longer needed: at biobank size they are what the test image has to fit beside
            JX_HIP(hipStreamSynchronize(st));                 // allocate / free with the device idle (see setup)
            t32.release();                                    // (a no-op when the images belong to the caller's scope)
            if (dte.alloc(sizeof(int32_t) * (size_t)n_test)) return 1;
            JX_HIP(hipMemcpy(dte.p, te32.data(), sizeof(int32_t) * (size_t)n_test, hipMemcpyHostToDevice));
            const int ntt = num_tiles(n_test);
            if (p32t.alloc((size_t)ntt * (size_t)eff_m * 32)) return 1;
            if (jxg_repack_p32(d_raw, bps, n_samples, m_total, dte.as<int32_t>(), n_test, d_rowidx, eff_m, p32t.as<uint8_t>(), st))
                return 1;
            if (jxg_packed_dot(p32t.as<uint8_t>(), eff_m, n_test, nullptr, (int)eff_m, L, v64m, v64n, st)) return 1;
            return 0;
        };
        if (!lerr) lerr = pred_test() ? 1 : 0;
        if (pcg_allreduce_dev_or_zero(v64n, n_test, nullptr, 0, st, lerr)) return 1;
        if (!lerr) {
            JX_HIP(hipStreamSynchronize(st));
            JX_HIP(hipMemcpy(out_pred_test, v64n, sizeof(double) * (size_t)n_test, hipMemcpyDeviceToHost));
            for (int i = 0; i < n_test; ++i) out_pred_test[i] = (double)(float)out_pred_test[i] + (double)alpha_use;
        }
    }
    if (multi) {
        if (pcg_allreduce_host(nullptr, 0, st, lerr)) return 1;      // the last verdict: every rank returns the same status
    } else if (lerr) {
        return 1;
    }
    JX_HIP(hipStreamSynchronize(st));
    out_scalars[0] = converged ? 1.0 : 0.0;
    out_scalars[1] = (double)iters;
    out_scalars[2] = rel_res;
    out_scalars[3] = sum_ss;
    out_scalars[4] = (double)alpha_use;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Haseman-Elston traces over the same matrix-free operator (second half of SURVEY 8f-4): src/stats/he.rs:1633-2070
// `he_variance_components_with_source`.  K = Z'Z / m_scale on the training samples, P = I - X (X'X)^-1 X' with
// X = [1, x_cov]; returns y'PKPy, y'Py and the Hutchinson (or exact) estimates of tr(PKP), tr((PKP)^2).
// The probes are the reference's (+-1 from splitmix64 chains, he.rs:1871-1881), so the estimates are deterministic.
// ---------------------------------------------------------------------------------------------------------------
namespace {
inline uint64_t he_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct HeProjector {   // he.rs:206-354
    int n = 0, p = 0;
    std::vector<double> x, chol;   // x (n,p) row-major, chol = lower Cholesky factor of X'X
    int init(int n_, const double *x_cov, int p_cov) {
        n = n_;
        p = 1 + p_cov;
        if (n <= p) return fail("HE projection requires n > rank(X): n=" + std::to_string(n) + ", p(with intercept)=" + std::to_string(p));
        x.assign((size_t)n * p, 0.0);
        for (int i = 0; i < n; ++i) {
            x[(size_t)i * p] = 1.0;
            for (int a = 0; a < p_cov; ++a) {
                const double v = x_cov[(size_t)i * p_cov + a];
                if (!isfinite(v)) return fail("x_cov contains non-finite values");
                x[(size_t)i * p + 1 + a] = v;
            }
        }
        chol.assign((size_t)p * p, 0.0);
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < p; ++a)
                for (int b = 0; b <= a; ++b) chol[(size_t)a * p + b] += x[(size_t)i * p + a] * x[(size_t)i * p + b];
        for (int a = 0; a < p; ++a) {   // in-place lower Cholesky
            for (int b = 0; b <= a; ++b) {
                double s = chol[(size_t)a * p + b];
                for (int k = 0; k < b; ++k) s -= chol[(size_t)a * p + k] * chol[(size_t)b * p + k];
                if (a == b) {
                    if (!(s > 0.0) || !isfinite(s))
                        return fail("HE covariate projection failed: X'X is singular or ill-conditioned (p=" + std::to_string(p) + ")");
                    chol[(size_t)a * p + a] = sqrt(s);
                } else {
                    chol[(size_t)a * p + b] = s / chol[(size_t)b * p + b];
                }
            }
        }
        return 0;
    }
    void coef(const std::vector<double> &xtv, std::vector<double> &beta) const {
        std::vector<double> t(p);
        for (int a = 0; a < p; ++a) {
            double s = xtv[a];
            for (int k = 0; k < a; ++k) s -= chol[(size_t)a * p + k] * t[k];
            t[a] = s / chol[(size_t)a * p + a];
        }
        for (int a = p - 1; a >= 0; --a) {
            double s = t[a];
            for (int k = a + 1; k < p; ++k) s -= chol[(size_t)k * p + a] * beta[k];
            beta[a] = s / chol[(size_t)a * p + a];
        }
    }
    void project64(std::vector<double> &v) const {
        std::vector<double> xtv(p, 0.0), beta(p, 0.0);
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < p; ++a) xtv[a] += x[(size_t)i * p + a] * v[i];
        coef(xtv, beta);
        for (int i = 0; i < n; ++i) {
            double xb = 0.0;
            for (int a = 0; a < p; ++a) xb += x[(size_t)i * p + a] * beta[a];
            v[i] -= xb;
        }
    }
    void project32(std::vector<float> &v) const {
        std::vector<double> xtv(p, 0.0), beta(p, 0.0);
        for (int i = 0; i < n; ++i)
            for (int a = 0; a < p; ++a) xtv[a] += x[(size_t)i * p + a] * (double)v[i];
        coef(xtv, beta);
        for (int i = 0; i < n; ++i) {
            double xb = 0.0;
            for (int a = 0; a < p; ++a) xb += x[(size_t)i * p + a] * beta[a];
            v[i] -= (float)xb;
        }
    }
};
}  // namespace

// out5: [0] y'PKPy, [1] y'Py, [2] tr(PKP), [3] tr((PKP)^2) (means over the probes, or exact sums), [4] tr(P)
extern "C" int jx_he_traces_packed(const uint8_t *packed, int64_t m_total, int n_samples, const int64_t *row_indices,
                                   int64_t eff_m, const float *value_lut, const int64_t *train_idx, int n_train,
                                   const double *y_train, const double *x_cov, int p_cov, int trace_samples,
                                   uint64_t seed, int exact_trace, double m_scale, double *out5) {
    if (n_samples <= 0) return fail("n_samples must be > 0");
    if (m_total <= 0 || eff_m <= 0) return fail("SNP row count must be > 0");
    if (n_train <= 0) return fail("sample_idx must not be empty");
    if (trace_samples <= 0) return fail("trace_samples must be > 0");
    if (eff_m > 0x7fffffffLL) return fail("too many markers for one call");
    if ((x_cov == nullptr) != (p_cov == 0)) return fail(x_cov ? "x_cov provided but p_cov == 0" : "p_cov > 0 but x_cov is None");
    const int64_t bps = ((int64_t)n_samples + 3) / 4;
    const int n = n_train;
    std::vector<int32_t> tr32(n);
    for (int i = 0; i < n; ++i) {
        if (train_idx[i] < 0 || train_idx[i] >= n_samples) return fail("train_sample_indices out of range");
        tr32[i] = (int32_t)train_idx[i];
    }
    if (row_indices)
        for (int64_t j = 0; j < eff_m; ++j)
            if (row_indices[j] < 0 || row_indices[j] >= m_total) return fail("row source index out of bounds");
    for (int i = 0; i < n; ++i)
        if (!isfinite(y_train[i])) return fail("y contains non-finite values");
    HeProjector P;
    if (P.init(n, x_cov, p_cov)) return 1;

    hipStream_t st = nullptr;
    DevBuf raw, didx, drow, p32, t32, dlut, dwork, dvn, dvm;
    const uint8_t *d_raw = packed;                 // a payload that already lives in HBM is used in place
    if (!is_device_ptr(packed)) {
        if (raw.alloc((size_t)(m_total * bps))) return 1;
        JX_HIP(hipMemcpy(raw.p, packed, (size_t)(m_total * bps), hipMemcpyHostToDevice));
        d_raw = raw.as<uint8_t>();
    }
    static const bool he_trace = getenv("JXGPU_PCG_TRACE") != nullptr;
    const auto he_t0 = std::chrono::steady_clock::now();
    auto hmark = [&](const char *what) {
        if (!he_trace) return;
        (void)hipDeviceSynchronize();
        fprintf(stderr, "[jxgpu he ] %-28s %8.1f ms\n", what,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - he_t0).count());
    };
    // every allocation before the first kernel launch (a hipMalloc behind kernels in flight costs up to seconds: see the PCG)
    if (dlut.alloc(sizeof(float) * 4 * (size_t)eff_m) || dwork.alloc(16 * (size_t)eff_m + 16) ||
        dvn.alloc(sizeof(double) * (size_t)n) || dvm.alloc(sizeof(double) * (size_t)eff_m) ||
        didx.alloc(sizeof(int32_t) * (size_t)n) || (row_indices && drow.alloc(sizeof(int64_t) * (size_t)eff_m)))
        return 1;
    JX_HIP(hipMemcpy(didx.p, tr32.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    const int64_t *d_rowidx = nullptr;
    if (row_indices) {
        JX_HIP(hipMemcpy(drow.p, row_indices, sizeof(int64_t) * (size_t)eff_m, hipMemcpyHostToDevice));
        d_rowidx = drow.as<int64_t>();
    }
    const uint8_t *P32 = nullptr, *T32 = nullptr;
    if (pcg_images(is_device_ptr(packed) ? (const void *)packed : nullptr, d_raw, bps, m_total, n_samples, row_indices, d_rowidx, eff_m,
                   train_idx, didx.as<int32_t>(), n, p32, t32, &P32, &T32, st))
        return 1;
    JX_HIP(hipStreamSynchronize(st));
    raw.release();
    hmark("images");
    JX_HIP(hipMemcpy(dlut.p, value_lut, sizeof(float) * 4 * (size_t)eff_m, hipMemcpyHostToDevice));
    const float ms = (float)m_scale;
    const float inv_m = 1.0f / (ms > 1.0f ? ms : 1.0f);
    std::vector<double> h64(n);
    // out = K v (f32 in, f32 out): Z v (f32 GEMV in the reference) -> Z'(.) -> * 1/m   (he.rs:1273-1327)
    EvSet hev;
    if (hev.make(2)) return 1;
    double he_ms = 0.0;
    int he_apps = 0;
    auto apply_k = [&](const std::vector<float> &v, std::vector<float> &o) -> int {
        for (int i = 0; i < n; ++i) h64[i] = (double)v[i];
        JX_HIP(hipMemcpyAsync(dvn.p, h64.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
        JX_HIP(hipEventRecord(hev[0], st));
        if (jxg_packed_tdot_f32(P32, eff_m, n, nullptr, (int)eff_m, dlut.as<float>(), dvn.as<double>(),
                                dvm.as<double>(), st))
            return 1;
        hipLaunchKernelGGL(pcg_round_kernel, dim3(pcg_grid_full(eff_m)), dim3(PCG_T), 0, st, dvm.as<double>(), eff_m);
        JX_LAUNCH_CHECK();
        if (jxg_packed_dot_t32(T32, n, (int)eff_m, dlut.as<float>(), dvm.as<double>(), dwork.p,
                               dvn.as<double>(), st))
            return 1;
        JX_HIP(hipEventRecord(hev[1], st));
        JX_HIP(hipMemcpyAsync(h64.data(), dvn.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        JX_HIP(hipStreamSynchronize(st));
        float ems = 0.f;
        if (hipEventElapsedTime(&ems, hev[0], hev[1]) == hipSuccess) he_ms += (double)ems;
        ++he_apps;
        o.resize(n);
        for (int i = 0; i < n; ++i) o[i] = (float)h64[i] * inv_m;
        return 0;
    };
    std::vector<double> yp(y_train, y_train + n);
    P.project64(yp);
    std::vector<float> y32(n), kv, probe(n);
    for (int i = 0; i < n; ++i) y32[i] = (float)yp[i];
    if (apply_k(y32, kv)) return 1;
    hmark("first application");
    double y_ky = 0.0, y_y = 0.0;
    for (int i = 0; i < n; ++i) {
        y_ky += (double)y32[i] * (double)kv[i];
        y_y += (double)y32[i] * (double)y32[i];
    }
    double trk = 0.0, trk2 = 0.0;
    if (exact_trace) {
        for (int c = 0; c < n; ++c) {
            std::fill(probe.begin(), probe.end(), 0.0f);
            probe[c] = 1.0f;
            P.project32(probe);
            if (apply_k(probe, kv)) return 1;
            P.project32(kv);
            trk += (double)kv[c];
            for (int i = 0; i < n; ++i) trk2 += (double)kv[i] * (double)kv[i];
        }
    } else {
        for (int t = 0; t < trace_samples; ++t) {
            uint64_t state = he_splitmix64(seed ^ ((uint64_t)t * 0x517CC1B727220A95ull));
            for (int i = 0; i < n; ++i) {
                state = he_splitmix64(state);
                probe[i] = ((state & 1ull) == 0) ? 1.0f : -1.0f;
            }
            P.project32(probe);
            if (apply_k(probe, kv)) return 1;
            P.project32(kv);
            double a = 0.0, b = 0.0;
            for (int i = 0; i < n; ++i) {
                a += (double)probe[i] * (double)kv[i];
                b += (double)kv[i] * (double)kv[i];
            }
            trk += a;
            trk2 += b;
        }
        trk /= (double)trace_samples;
        trk2 /= (double)trace_samples;
    }
    hmark("all probes");
    g_last_ms[22] = (float)he_ms;
    g_last_ms[23] = (float)he_apps;
    out5[0] = y_ky;
    out5[1] = y_y;
    out5[2] = trk;
    out5[3] = trk2;
    const double trp = (double)n - (double)P.p;
    out5[4] = trp > 1.0 ? trp : 1.0;
    return 0;
}
