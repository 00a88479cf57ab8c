// Host layer of libjxgpu: reference-shaped entry points (host arrays in, host arrays out) that stage through
// HBM and drive the device layer; per-SNP statistics / filter logic (integer counts -> f32/f64 decisions) is
// done on the host exactly as the reference writes it, so the kept-SNP set is bit-exact.
#include <errno.h>
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <chrono>
#include <vector>

#include "jx_common.h"

namespace jx {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(const std::string &msg) {
    g_err = msg;
    return 1;
}

namespace {
struct ScratchSlot {
    void *p = nullptr;
    size_t cap = 0;
    bool busy = false;
};
ScratchSlot g_scratch[8];
std::mutex g_scratch_mu;
}  // namespace

int scratch_acquire(int which, size_t bytes, void **p) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    ScratchSlot &s = g_scratch[which & 7];
    if (s.busy) return 1;
    if (s.cap < bytes) {
        if (s.p) (void)hipFree(s.p);
        s.p = nullptr;
        s.cap = 0;
        hipError_t e = hipMalloc(&s.p, bytes);
        if (e != hipSuccess) {
            s.p = nullptr;
            fail(std::string("hipMalloc(") + std::to_string(bytes) + ") failed: " + hipGetErrorString(e));
            return 2;
        }
        s.cap = bytes;
    }
    s.busy = true;
    *p = s.p;
    return 0;
}

void scratch_release(int which) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    g_scratch[which & 7].busy = false;
}

// Hand the kept scratch blocks (<= 6 GB each, grown on demand by the eigensolver's stages) back to the driver: for a long-lived
// host process between two problems of very different size.  Blocks leased by a running call stay.  -> bytes released
extern "C" int64_t jxg_scratch_trim(void) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    int64_t freed = 0;
    for (ScratchSlot &s : g_scratch) {
        if (s.busy || !s.p) continue;
        (void)hipFree(s.p);
        freed += (int64_t)s.cap;
        s.p = nullptr;
        s.cap = 0;
    }
    return freed;
}

// keep the stream-ordered pool's blocks across synchronisations (default threshold 0: every hipFreeAsync'ed block goes back to the
// driver at the next synchronisation and the next hipMallocAsync pays a fresh allocation): 256 MB cover the largest user
void async_pool_keep() {
    static bool pool_set = false;
    if (pool_set) return;
    int dev = 0;
    hipMemPool_t pool = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
        uint64_t thr = (uint64_t)256 << 20;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
    }
    (void)hipGetLastError();
    pool_set = true;
}

int launch_symmetrize(double *d_a, int n, hipStream_t st);
int launch_transpose_f64(const double *src, double *dst, int n, hipStream_t st);

// ---- value LUTs (host) ---------------------------------------------------------------------------

// GRM design LUT indexed by the 2-bit code [00, 01(missing), 10, 11]
// (src/decode/decode.rs:813-839, src/math/bedmath.rs:1208-1224, decode.rs:558-566).
static void grm_lut_from_maf(float maf, bool flip, int method, float out[4]) {
    float p = maf;
    if (p < 0.0f) p = 0.0f;
    if (p > 1.0f) p = 1.0f;
    const float mean_g = 2.0f * p;
    const float var = 2.0f * p * (1.0f - p);
    float sc = 1.0f;
    if (method == 2) sc = (var > 1e-12f) ? (1.0f / sqrtf(var)) : 0.0f;
    const float g0 = flip ? 2.0f : 0.0f, g1 = 1.0f, g2 = flip ? 0.0f : 2.0f;
    out[0] = (g0 - mean_g) * sc;
    out[1] = 0.0f;
    out[2] = (g1 - mean_g) * sc;
    out[3] = (g2 - mean_g) * sc;
}

// `PackedGeneticModel::apply` (src/decode/decode.rs:132-160): 0 add, 1 dom, 2 rec, 3 het
static inline float gm_apply(int gm, float gf) {
    const double g = (double)gf;
    switch (gm) {
    case 1: return g > 0.0 ? 1.0f : 0.0f;
    case 2: return fabs(g - 2.0) < 1e-6 ? 1.0f : 0.0f;
    case 3: return fabs(g - 1.0) < 1e-6 ? 1.0f : 0.0f;
    default: return gf;
    }
}
// scan design LUT: the genetic model applied to [0, mu, 1, 2] (or flipped), minus the actual row mean
// (src/decode/decode.rs:163-189, 218-221). counts = (missing, het, hom_alt) over the n selected samples.
static void scan_lut_from_counts(float maf, bool flip, const int32_t *cnt, int n, float out[4], int gm = 0) {
    const float mu = gm_apply(gm, (float)fmax(2.0 * (double)maf, 0.0));
    const float v0 = gm_apply(gm, flip ? 2.0f : 0.0f), v2 = gm_apply(gm, 1.0f), v3 = gm_apply(gm, flip ? 0.0f : 2.0f);
    const double c00 = (double)(n - cnt[0] - cnt[1] - cnt[2]);
    const double sum = c00 * (double)v0 + (double)cnt[0] * (double)mu + (double)cnt[1] * (double)v2 +
                       (double)cnt[2] * (double)v3;
    const float mean = (float)(sum / (double)n);
    out[0] = v0 - mean;
    out[1] = mu - mean;
    out[2] = v2 - mean;
    out[3] = v3 - mean;
}

// denominators, src/stats/grm.rs:91-111 (full-sample centred) and bedmath.rs:1411-1438 (subset route)
static double grm_varsum(const float *row_maf, int64_t m, int method, bool full) {
    if (method != 1) return (double)m;
    double acc = 0.0;
    if (full) {
        for (int64_t j = 0; j < m; ++j) {
            const double p = (double)row_maf[j];
            const double v = 2.0 * p * (1.0 - p);
            if (isfinite(v) && v > 0.0) acc += v;
        }
    } else {
        for (int64_t j = 0; j < m; ++j) {
            float p0 = row_maf[j];
            p0 = p0 < 0.0f ? 0.0f : (p0 > 1.0f ? 1.0f : p0);
            const float mean_g = 2.0f * p0;
            float pg = 0.5f * mean_g;
            pg = pg < 0.0f ? 0.0f : (pg > 1.0f ? 1.0f : pg);
            float v = 2.0f * pg * (1.0f - pg);
            if (v < 0.0f) v = 0.0f;
            acc += (double)v;
        }
    }
    return acc;
}

struct SampleSel {
    std::vector<int32_t> idx;
    bool identity = true;
    int n = 0;
};

static int make_sample_sel(const int64_t *sample_indices, int n_sel, int n_samples, SampleSel &s) {
    if (!sample_indices) {
        s.identity = true;
        s.n = n_samples;
        return 0;
    }
    if (n_sel <= 0) return fail("sample_indices must not be empty");
    s.idx.resize(n_sel);
    s.identity = (n_sel == n_samples);
    for (int i = 0; i < n_sel; ++i) {
        const int64_t v = sample_indices[i];
        if (v < 0 || v >= n_samples)
            return fail("sample index out of range: " + std::to_string(v) + " >= " + std::to_string(n_samples));
        s.idx[i] = (int32_t)v;
        if (v != i) s.identity = false;
    }
    s.n = n_sel;
    return 0;
}

// true when `p` points into device memory (a payload that is already resident in HBM: torch CUDA tensor, hipMalloc)
bool is_device_ptr(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();                     // plain (unregistered) host memory: not an error for the caller
        return false;
    }
    return at.type == hipMemoryTypeDevice;
}

// Re-tile a PLINK payload to P32 (with optional sample subset). p32 must stay alive.  `packed` may be a host pointer
// (uploaded first) or a device pointer (used in place: no host copy of the payload exists anywhere).  Refuses with a
// clear message, before anything is allocated, when the device copies do not fit the free HBM.
static int stage_p32(const uint8_t *packed, int64_t m, int n_samples, const SampleSel &sel, DevBuf &p32) {
    const int64_t bps = (n_samples + 3) / 4;
    const bool on_device = is_device_ptr(packed);
    const int nt = num_tiles(sel.n);
    {
        size_t fr = 0, tot = 0;
        const double need = (on_device ? 0.0 : (double)m * (double)bps) + (double)nt * (double)m * 32.0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && need > 0.95 * (double)fr)
            return fail("packed payload of " + std::to_string((long long)(m * bps >> 20)) + " MiB: its device images need " +
                        std::to_string((long long)(need / 1048576.0)) + " MiB of HBM, " + std::to_string((long long)(fr >> 20)) +
                        " MiB are free (split the SNP rows over several calls)");
    }
    static const bool trace = getenv("JXGPU_PCG_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!trace) return;
        (void)hipDeviceSynchronize();
        fprintf(stderr, "[jxgpu stage_p32] %-20s %8.1f ms\n", what,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    };
    // every allocation of the staging BEFORE the first copy or launch: a hipMalloc behind work in flight (or right behind a large
    // hipFree) has been measured at hundreds of milliseconds to seconds on this stack, with the device idle 0.3 ms
    DevBuf raw, didx;
    if (!on_device && raw.alloc((size_t)(m * bps))) return 1;
    if (!sel.identity && didx.alloc(sizeof(int32_t) * sel.idx.size())) return 1;
    if (p32.alloc((size_t)nt * (size_t)m * 32)) return 1;
    mark("hipMalloc raw / idx / p32");
    const uint8_t *d_raw = packed;
    if (!on_device) {
        JX_HIP(hipMemcpy(raw.p, packed, (size_t)(m * bps), hipMemcpyHostToDevice));
        d_raw = raw.as<uint8_t>();
    }
    const int32_t *d_idx = nullptr;
    if (!sel.identity) {
        JX_HIP(hipMemcpy(didx.p, sel.idx.data(), sizeof(int32_t) * sel.idx.size(), hipMemcpyHostToDevice));
        d_idx = didx.as<int32_t>();
    }
    mark("uploads");
    if (jxg_repack_p32(d_raw, bps, n_samples, m, d_idx, sel.n, nullptr, m, p32.as<uint8_t>(), nullptr))
        return 1;
    JX_HIP(hipDeviceSynchronize());
    mark("repack");
    return 0;
}

}  // namespace jx

using namespace jx;

extern "C" const char *jx_last_error(void) { return g_err.c_str(); }
extern "C" int jx_version(void) { return 100; }

// ---- progress hook ------------------------------------------------------------------------------------------------------
// The reference calls `progress_callback(done, total)` every `progress_every` rows (default: its rotate block) and lets a
// Python exception raised there (KeyboardInterrupt) end the scan (src/stats/lmm.rs:3214-3330, src/stats/grm.rs:3485-3495).
// The host layer's row-block loops report through this process-wide hook: `fn` returns nonzero to stop the call, which
// then fails with "interrupted by the progress callback".  every <= 0: once per internal block.
typedef int (*jx_progress_fn)(int64_t done, int64_t total, void *user);
namespace {
// per calling thread: ctypes runs a call on the Python thread that made it and releases the GIL, so two concurrent
// scans from different threads each see their own hook
thread_local jx_progress_fn g_progress = nullptr;
thread_local void *g_progress_user = nullptr;
thread_local int64_t g_progress_every = 0;
struct ProgressTicker {
    int64_t last = 0;
    // 0 = go on, 1 = the callback asked to stop
    int tick(int64_t done, int64_t total, int64_t block) {
        if (!g_progress) return 0;
        const int64_t step = g_progress_every > 0 ? g_progress_every : block;
        if (done < total && done < last + step) return 0;
        last = done;
        return g_progress(done, total, g_progress_user) != 0;
    }
};
}  // namespace
extern "C" int jx_set_progress(jx_progress_fn fn, void *user, int64_t every) {
    g_progress = fn;
    g_progress_user = user;
    g_progress_every = every;
    return 0;
}

extern "C" int jxg_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

extern "C" int jxg_set_device(int dev) {
    JX_HIP(hipSetDevice(dev));
    return 0;
}

extern "C" int jxg_device_info(int64_t *out4) {
    int dev = 0;
    JX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t pr;
    JX_HIP(hipGetDeviceProperties(&pr, dev));
    out4[0] = pr.multiProcessorCount;
    out4[1] = pr.clockRate;
    out4[2] = (int64_t)(pr.totalGlobalMem >> 20);
    out4[3] = (int64_t)pr.maxSharedMemoryPerMultiProcessor;
    return 0;
}

// per-SNP (missing, het, hom_alt) over the selected samples (src/io/gfreader.rs:1378-1395
// `count_packed_row_counts[_selected]`), host arrays in/out.
extern "C" int jx_row_counts(const uint8_t *packed, int64_t m, int n_samples, const int64_t *sample_indices,
                             int n_sel, int32_t *out_counts) {
    if (n_samples <= 0) return fail("n_samples must be > 0");
    if (m <= 0) return 0;
    SampleSel sel;
    if (make_sample_sel(sample_indices, n_sel, n_samples, sel)) return 1;
    DevBuf p32, dcnt;
    if (is_device_ptr(packed)) {
        // a payload that already lives in HBM: counts straight from it under a sample mask (no 40 GB image allocated, filled and
        // freed for three numbers per SNP) -- when the selection has no duplicate (a mask counts a sample once)
        const int64_t bps = ((int64_t)n_samples + 3) / 4;
        std::vector<uint8_t> mask((size_t)bps, 0);
        bool dup = false;
        if (sel.identity) {
            for (int i = 0; i < n_samples; ++i) mask[(size_t)(i >> 2)] |= (uint8_t)(3u << (2 * (i & 3)));
        } else {
            for (int32_t v : sel.idx) {
                const uint8_t bit = (uint8_t)(3u << (2 * (v & 3)));
                if (mask[(size_t)(v >> 2)] & bit) dup = true;
                mask[(size_t)(v >> 2)] |= bit;
            }
        }
        if (!dup) {
            DevBuf dmask;
            if (dmask.alloc((size_t)bps) || dcnt.alloc(sizeof(int32_t) * 3 * (size_t)m)) return 1;
            JX_HIP(hipMemcpy(dmask.p, mask.data(), (size_t)bps, hipMemcpyHostToDevice));
            if (jxg_row_counts_raw_masked(packed, bps, m, dmask.as<uint8_t>(), dcnt.as<int32_t>(), nullptr)) return 1;
            JX_HIP(hipMemcpy(out_counts, dcnt.p, sizeof(int32_t) * 3 * (size_t)m, hipMemcpyDeviceToHost));
            return 0;
        }
    }
    if (stage_p32(packed, m, n_samples, sel, p32)) return 1;
    if (dcnt.alloc(sizeof(int32_t) * 3 * (size_t)m)) return 1;
    if (jxg_row_counts_p32(p32.as<uint8_t>(), m, sel.n, dcnt.as<int32_t>(), nullptr)) return 1;
    JX_HIP(hipMemcpy(out_counts, dcnt.p, sizeof(int32_t) * 3 * (size_t)m, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// grm_packed_f32 / grm_packed_f64_with_stats (src/stats/grm.rs:204-360, 3066, 5611)
// ---------------------------------------------------------------------------------------------------
extern "C" int jx_grm_packed(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                             const float *row_maf, const int64_t *sample_indices, int n_sel, int method,
                             void *out_k, int out_is_f64, double *out_row_sum, double *out_varsum) {
    if (n_samples <= 0) return fail("n_samples must be > 0");
    if (method != 1 && method != 2)
        return fail("unsupported method=" + std::to_string(method) + "; expected 1 (centered) or 2 (standardized)");
    if (m <= 0) return fail("packed must contain at least one SNP row");
    SampleSel sel;
    if (make_sample_sel(sample_indices, n_sel, n_samples, sel)) return 1;
    const int n = sel.n;
    const double D = grm_varsum(row_maf, m, method, sel.identity);
    if (!(isfinite(D) && D > 0.0)) return fail("invalid centered GRM denominator: sum(2p(1-p)) <= 0");

    std::vector<float> lut((size_t)m * 4);
    for (int64_t j = 0; j < m; ++j) {
        grm_lut_from_maf(row_maf[j], row_flip[j] != 0, method, &lut[(size_t)j * 4]);
        if (out_row_sum) {
            float p = row_maf[j];
            p = p < 0.0f ? 0.0f : (p > 1.0f ? 1.0f : p);
            // decode.rs:795-799 (method 2) / :841 (method 1): mean_g * n_out
            out_row_sum[j] = (method == 2) ? 2.0 * (double)p * (double)n : (double)(2.0f * p) * (double)n;
        }
    }
    DevBuf p32, dlut, acc, dout;
    if (stage_p32(packed, m, n_samples, sel, p32)) return 1;
    if (dlut.alloc(lut.size() * sizeof(float))) return 1;
    JX_HIP(hipMemcpy(dlut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
    const int64_t npad = (int64_t)num_tiles(n) * JXG_TILE;
    if (acc.alloc(sizeof(double) * (size_t)(npad * npad))) return 1;
    JX_HIP(hipMemset(acc.p, 0, sizeof(double) * (size_t)(npad * npad)));
    if (jxg_grm_accumulate(p32.as<uint8_t>(), m, n, nullptr, dlut.as<float>(), m, acc.as<double>(), 0, 0, nullptr))
        return 1;
    const size_t esz = out_is_f64 ? sizeof(double) : sizeof(float);
    if (dout.alloc(esz * (size_t)n * (size_t)n)) return 1;
    if (jxg_grm_finalize(acc.as<double>(), n, 1.0 / D, dout.p, out_is_f64, nullptr)) return 1;
    JX_HIP(hipMemcpy(out_k, dout.p, esz * (size_t)n * (size_t)n, hipMemcpyDeviceToHost));
    if (out_varsum) *out_varsum = (method == 1) ? D : (double)m;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// spgrm_packed_to_jxgrm / the stream core behind spgrm_bed_to_jxgrm (src/stats/spgrm.rs:3769-3908, 3910-4264)
// ---------------------------------------------------------------------------------------------------
extern "C" int64_t jxg_spgrm_work_bytes(int n);
extern "C" int jxg_spgrm_count(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold,
                               void *d_work, uint64_t *d_colptr, void *stream);
extern "C" int jxg_spgrm_count_bands(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold,
                                     int band0, int band1, void *d_work, uint64_t *d_colptr, void *stream);
extern "C" int jxg_spgrm_fill_bands(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold,
                                    int band0, int band1, const void *d_work, const uint64_t *d_colptr, uint32_t *d_rows,
                                    double *d_vals, void *stream);
extern "C" int jxg_grm_accumulate_rows(const uint8_t *d_p32, int64_t m_total, int n_sel, const int32_t *d_rows,
                                       const float *d_lut, int64_t mk, double *d_acc, int kchunk, int precision,
                                       int tile_row_begin, int tile_row_end, void *stream);
extern "C" int jxg_spgrm_fill(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold,
                              const void *d_work, const uint64_t *d_colptr, uint32_t *d_rows, double *d_vals,
                              void *stream);

// ---- sparse GRM by row panels: entries of a panel, merge + file writer, part files for several ranks ---------------------
namespace {
struct PanelEntries {
    int64_t index = 0;                 // position of the panel (rows ascend with it)
    std::vector<uint64_t> cp;          // n + 1 offsets into r / v
    std::vector<uint32_t> r;
    std::vector<double> v;
};
// (part, nparts) of this process for jx_spgrm_packed_to_jxgrm: with nparts > 1 the row panels are dealt cyclically, a process
// builds only its own and writes them to `<out>.part<k>`; jx_spgrm_merge_parts joins the files (jx_spgrm_set_part)
int g_spgrm_part = 0, g_spgrm_nparts = 1;

// write_sparse_grm_csc (src/stats/spgrm.rs:3745-3767): u64 n, u64 nnz, col_ptr, row_indices, zero padding to 8 bytes, values (LE);
// the panels (ascending index) are merged column by column (rows ascend with the panels, so the order inside a column is kept)
int spgrm_merge_write(const std::vector<PanelEntries> &panels, int n, const char *out_path, uint64_t *out_nnz) {
    std::vector<uint64_t> colptr((size_t)n + 1);
    uint64_t nnz = 0;
    for (const auto &pe : panels) nnz += pe.cp[(size_t)n];
    colptr[0] = 0;
    for (int c = 0; c < n; ++c) {
        uint64_t k = 0;
        for (const auto &pe : panels) k += pe.cp[(size_t)c + 1] - pe.cp[(size_t)c];
        colptr[(size_t)c + 1] = colptr[(size_t)c] + k;
    }
    std::vector<uint32_t> rows;
    std::vector<double> vals;
    try {
        rows.resize((size_t)nnz);
        vals.resize((size_t)nnz);
    } catch (const std::bad_alloc &) {
        return fail("Sparse GRM: host allocation of " + std::to_string(nnz) + " entries failed");
    }
    std::vector<uint64_t> cursor(colptr.begin(), colptr.end() - 1);
    for (const auto &pe : panels)
        for (int c = 0; c < n; ++c)
            for (uint64_t k = pe.cp[(size_t)c]; k < pe.cp[(size_t)c + 1]; ++k) {
                rows[(size_t)cursor[(size_t)c]] = pe.r[(size_t)k];
                vals[(size_t)cursor[(size_t)c]++] = pe.v[(size_t)k];
            }
    FILE *fh = fopen(out_path, "wb");
    if (!fh) return fail(std::string("create ") + out_path + " failed");
    const uint64_t hdr[2] = {(uint64_t)n, nnz};
    const size_t pad = (size_t)((8 - ((nnz * 4) & 7)) & 7);
    const char zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = fwrite(hdr, 8, 2, fh) == 2 && fwrite(colptr.data(), 8, colptr.size(), fh) == colptr.size() &&
              fwrite(rows.data(), 4, rows.size(), fh) == rows.size() && fwrite(zeros, 1, pad, fh) == pad &&
              fwrite(vals.data(), 8, vals.size(), fh) == vals.size();
    const int werr = ok ? 0 : errno;
    const bool closed = fclose(fh) == 0;
    if (!ok || !closed) {
        const int e = ok ? errno : werr;
        remove(out_path);
        return fail(std::string("write sparse GRM file failed: ") + out_path + " (" + std::to_string(nnz) + " entries, " +
                    strerror(e) + ")");
    }
    if (out_nnz) *out_nnz = nnz;
    return 0;
}
}  // namespace

// Several processes (one per GPU) build ONE sparse GRM: process `part` of `nparts` computes the row panels p = part (mod nparts)
// in its next jx_spgrm_packed_to_jxgrm call and leaves them in `<out>.part<part>`; after a barrier ONE process calls
// jx_spgrm_merge_parts(out, n, nparts).  (0, 1): off.
extern "C" int jx_spgrm_set_part(int part, int nparts) {
    if (nparts < 1 || part < 0 || part >= nparts) return fail("jx_spgrm_set_part: bad part / nparts");
    g_spgrm_part = part;
    g_spgrm_nparts = nparts;
    return 0;
}

extern "C" int jx_spgrm_merge_parts(const char *out_path, int n, int nparts, int64_t *out_nnz) {
    if (!out_path || !out_path[0] || n <= 0 || nparts < 1) return fail("jx_spgrm_merge_parts: bad arguments");
    std::vector<PanelEntries> panels;
    for (int k = 0; k < nparts; ++k) {
        const std::string path = std::string(out_path) + ".part" + std::to_string(k);
        FILE *fh = fopen(path.c_str(), "rb");
        if (!fh) return fail("jx_spgrm_merge_parts: " + path + " is missing");
        uint64_t hdr[2] = {0, 0};
        bool ok = fread(hdr, 8, 2, fh) == 2 && hdr[0] == (uint64_t)n;
        for (uint64_t q = 0; ok && q < hdr[1]; ++q) {
            PanelEntries pe;
            uint64_t idx = 0;
            pe.cp.resize((size_t)n + 1);
            ok = fread(&idx, 8, 1, fh) == 1 && fread(pe.cp.data(), 8, pe.cp.size(), fh) == pe.cp.size();
            if (!ok) break;
            pe.index = (int64_t)idx;
            const uint64_t pn = pe.cp[(size_t)n];
            pe.r.resize((size_t)pn);
            pe.v.resize((size_t)pn);
            ok = fread(pe.r.data(), 4, (size_t)pn, fh) == (size_t)pn && fread(pe.v.data(), 8, (size_t)pn, fh) == (size_t)pn;
            panels.push_back(std::move(pe));
        }
        fclose(fh);
        if (!ok) return fail("jx_spgrm_merge_parts: " + path + " is truncated or belongs to another matrix");
    }
    std::sort(panels.begin(), panels.end(), [](const PanelEntries &a, const PanelEntries &b) { return a.index < b.index; });
    for (size_t q = 1; q < panels.size(); ++q)
        if (panels[q].index == panels[q - 1].index) return fail("jx_spgrm_merge_parts: a panel appears twice");
    uint64_t nnz = 0;
    if (spgrm_merge_write(panels, n, out_path, &nnz)) return 1;
    for (int k = 0; k < nparts; ++k) remove((std::string(out_path) + ".part" + std::to_string(k)).c_str());
    if (out_nnz) *out_nnz = (int64_t)nnz;
    return 0;
}

extern "C" int jx_spgrm_packed_to_jxgrm(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                                        const float *row_maf, const int64_t *sample_indices, int n_sel, int method,
                                        double threshold, int abs_threshold, int stream_denominator,
                                        const char *out_path, int64_t *out_n, int64_t *out_nnz) {
    // validate_spgrm_inputs (spgrm.rs:2858-2915)
    if (n_samples <= 0) return fail("Sparse GRM requires n_samples > 0");
    if (method != 1 && method != 2)
        return fail("Sparse GRM method must be 1 (centered) or 2 (standardized); got " + std::to_string(method));
    if (!isfinite(threshold)) return fail("Sparse GRM threshold must be finite");
    if (sample_indices && n_sel <= 0) return fail("Sparse GRM sample_indices must not be empty");
    if (m <= 0) return fail("Sparse GRM requires at least one SNP row");
    if (!out_path || !out_path[0]) return fail("Sparse GRM output prefix must not be empty");
    SampleSel sel;
    if (make_sample_sel(sample_indices, n_sel, n_samples, sel)) return 1;
    const int n = sel.n;
    // packed route: `centered_varsum_from_packed` (:2917-2979); stream route: sum of 2p(1-p) in f64 whatever the
    // sample selection (:3973-3989)
    const double D = grm_varsum(row_maf, m, method, sel.identity || stream_denominator != 0);
    if (!(isfinite(D) && D > 0.0))
        return fail(method == 1 ? "Sparse GRM centered denominator is not positive"
                                : "Sparse GRM denominator is not positive");
    std::vector<float> lut((size_t)m * 4);
    for (int64_t j = 0; j < m; ++j) grm_lut_from_maf(row_maf[j], row_flip[j] != 0, method, &lut[(size_t)j * 4]);
    DevBuf p32, dlut, acc, work, dcolptr, drows, dvals;
    if (stage_p32(packed, m, n_samples, sel, p32)) return 1;
    if (dlut.alloc(lut.size() * sizeof(float))) return 1;
    JX_HIP(hipMemcpy(dlut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
    const int64_t npad = (int64_t)num_tiles(n) * JXG_TILE;
    const double inv = 1.0 / D;
    std::vector<uint64_t> colptr((size_t)n + 1);
    std::vector<uint32_t> rows;
    std::vector<double> vals;
    uint64_t nnz = 0;
    // Row panels when the n x n f64 accumulator is too large for HBM (JXGPU_SPGRM_ACC_GB, default 96) or when
    // JXGPU_SPGRM_PANEL_ROWS asks for them: the GRM is computed and thresholded 256-row bands at a time (tile rows of the
    // lower triangle), every panel yields the CSC entries of its rows, and the panels are merged column by column on the
    // host (rows ascend with the panels, so the order inside a column is kept).  One pass over the payload per panel.
    const char *env_rows = getenv("JXGPU_SPGRM_PANEL_ROWS");
    const double acc_gb = getenv("JXGPU_SPGRM_ACC_GB") ? atof(getenv("JXGPU_SPGRM_ACC_GB")) : 96.0;
    int64_t panel_rows = 0;
    if (env_rows && atoll(env_rows) > 0) panel_rows = atoll(env_rows);
    else if ((double)npad * (double)npad * 8.0 > acc_gb * 1073741824.0) panel_rows = (int64_t)(32.0 * 1073741824.0 / (8.0 * npad));
    const int nparts = g_spgrm_nparts, part = g_spgrm_part;
    if (nparts > 1 && panel_rows <= 0)      // several processes: about four panels each (the same plan on every one of them)
        panel_rows = std::max<int64_t>(256, ((int64_t)n / (4 * nparts)) / 256 * 256);
    if (panel_rows > 0) {
        panel_rows = std::max<int64_t>(256, (panel_rows / 256) * 256);
        const int nbands = (n + 255) / 256;
        const int bands_per = (int)(panel_rows / 256);
        if (acc.alloc(sizeof(double) * (size_t)(panel_rows * npad))) return 1;
        if (work.alloc((size_t)((int64_t)bands_per * n * 4 + 16))) return 1;
        if (dcolptr.alloc(sizeof(uint64_t) * ((size_t)n + 1))) return 1;
        std::vector<PanelEntries> panels;
        int64_t pidx = 0;
        for (int b0 = 0; b0 < nbands; b0 += bands_per, ++pidx) {
            // dealt back and forth (0 .. nparts-1, nparts-1 .. 0, ...): a panel's cost grows with its row index (lower triangle)
            const int64_t round = pidx / nparts, pos = pidx % nparts;
            if (((round & 1) ? nparts - 1 - pos : pos) != part) continue;          // another process's panel
            const int b1 = std::min(nbands, b0 + bands_per);
            const int t0 = b0 * 2, t1 = std::min((int)num_tiles(n), b1 * 2);       // 128-row tiles of the bands
            JX_HIP(hipMemset(acc.p, 0, sizeof(double) * (size_t)((int64_t)(t1 - t0) * JXG_TILE * npad)));
            if (jxg_grm_accumulate_rows(p32.as<uint8_t>(), m, n, nullptr, dlut.as<float>(), m, acc.as<double>(), 0, 2, t0, t1,
                                        nullptr))
                return 1;
            if (jxg_spgrm_count_bands(acc.as<double>(), n, inv, threshold, abs_threshold, b0, b1, work.p,
                                      dcolptr.as<uint64_t>(), nullptr))
                return 1;
            PanelEntries pe;
            pe.index = pidx;
            pe.cp.resize((size_t)n + 1);
            JX_HIP(hipMemcpy(pe.cp.data(), dcolptr.p, sizeof(uint64_t) * pe.cp.size(), hipMemcpyDeviceToHost));
            const uint64_t pn = pe.cp[(size_t)n];
            if (pn) {
                DevBuf pr, pv;
                if (pr.alloc(sizeof(uint32_t) * (size_t)pn) || pv.alloc(sizeof(double) * (size_t)pn)) return 1;
                if (jxg_spgrm_fill_bands(acc.as<double>(), n, inv, threshold, abs_threshold, b0, b1, work.p,
                                         dcolptr.as<uint64_t>(), pr.as<uint32_t>(), pv.as<double>(), nullptr))
                    return 1;
                try {
                    pe.r.resize((size_t)pn);
                    pe.v.resize((size_t)pn);
                } catch (const std::bad_alloc &) {
                    return fail("Sparse GRM: host allocation of " + std::to_string(pn) + " entries failed");
                }
                JX_HIP(hipMemcpy(pe.r.data(), pr.p, sizeof(uint32_t) * (size_t)pn, hipMemcpyDeviceToHost));
                JX_HIP(hipMemcpy(pe.v.data(), pv.p, sizeof(double) * (size_t)pn, hipMemcpyDeviceToHost));
            }
            nnz += pn;
            panels.push_back(std::move(pe));
        }
        p32.release();
        if (nparts > 1) {
            // this process's panels only: left in `<out>.part<k>` for jx_spgrm_merge_parts
            const std::string path = std::string(out_path) + ".part" + std::to_string(part);
            FILE *fh = fopen(path.c_str(), "wb");
            if (!fh) return fail("create " + path + " failed");
            const uint64_t hdr[2] = {(uint64_t)n, (uint64_t)panels.size()};
            bool ok = fwrite(hdr, 8, 2, fh) == 2;
            for (const auto &pe : panels) {
                const uint64_t idx = (uint64_t)pe.index;
                ok = ok && fwrite(&idx, 8, 1, fh) == 1 && fwrite(pe.cp.data(), 8, pe.cp.size(), fh) == pe.cp.size() &&
                     fwrite(pe.r.data(), 4, pe.r.size(), fh) == pe.r.size() && fwrite(pe.v.data(), 8, pe.v.size(), fh) == pe.v.size();
            }
            const bool closed = fclose(fh) == 0;
            if (!ok || !closed) {
                remove(path.c_str());
                return fail("write sparse GRM part file failed: " + path);
            }
            if (out_n) *out_n = n;
            if (out_nnz) *out_nnz = (int64_t)nnz;
            return 0;
        }
        uint64_t total = 0;
        if (spgrm_merge_write(panels, n, out_path, &total)) return 1;
        if (out_n) *out_n = n;
        if (out_nnz) *out_nnz = (int64_t)total;
        return 0;
    } else {
        if (acc.alloc(sizeof(double) * (size_t)(npad * npad))) return 1;
        JX_HIP(hipMemset(acc.p, 0, sizeof(double) * (size_t)(npad * npad)));
        // precision 2: rows with missing calls on the general kernel, as in the row-panel form above (one file whatever the
        // memory plan)
        if (jxg_grm_accumulate(p32.as<uint8_t>(), m, n, nullptr, dlut.as<float>(), m, acc.as<double>(), 0, 2, nullptr))
            return 1;
        p32.release();
        if (work.alloc((size_t)jxg_spgrm_work_bytes(n))) return 1;
        if (dcolptr.alloc(sizeof(uint64_t) * ((size_t)n + 1))) return 1;
        if (jxg_spgrm_count(acc.as<double>(), n, inv, threshold, abs_threshold, work.p, dcolptr.as<uint64_t>(), nullptr))
            return 1;
        JX_HIP(hipMemcpy(colptr.data(), dcolptr.p, sizeof(uint64_t) * colptr.size(), hipMemcpyDeviceToHost));
        nnz = colptr[(size_t)n];
        try {   // up to n (n + 1) / 2 entries with a negative cut-off: 12 bytes each on the host
            rows.resize((size_t)nnz);
            vals.resize((size_t)nnz);
        } catch (const std::bad_alloc &) {
            return fail("Sparse GRM: host allocation of " + std::to_string(nnz) + " entries failed");
        }
        if (nnz) {
            if (drows.alloc(sizeof(uint32_t) * (size_t)nnz) || dvals.alloc(sizeof(double) * (size_t)nnz)) return 1;
            if (jxg_spgrm_fill(acc.as<double>(), n, inv, threshold, abs_threshold, work.p, dcolptr.as<uint64_t>(),
                               drows.as<uint32_t>(), dvals.as<double>(), nullptr))
                return 1;
            JX_HIP(hipMemcpy(rows.data(), drows.p, sizeof(uint32_t) * (size_t)nnz, hipMemcpyDeviceToHost));
            JX_HIP(hipMemcpy(vals.data(), dvals.p, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost));
        }
    }
    // write_sparse_grm_csc (:3745-3767): u64 n, u64 nnz, col_ptr, row_indices, zero padding to 8 bytes, values (LE)
    FILE *fh = fopen(out_path, "wb");
    if (!fh) return fail(std::string("create ") + out_path + " failed");
    const uint64_t hdr[2] = {(uint64_t)n, nnz};
    const size_t pad = (size_t)((8 - ((nnz * 4) & 7)) & 7);
    const char zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = fwrite(hdr, 8, 2, fh) == 2 && fwrite(colptr.data(), 8, colptr.size(), fh) == colptr.size() &&
              fwrite(rows.data(), 4, rows.size(), fh) == rows.size() && fwrite(zeros, 1, pad, fh) == pad &&
              fwrite(vals.data(), 8, vals.size(), fh) == vals.size();
    const int werr = ok ? 0 : errno;
    const bool closed = fclose(fh) == 0;
    if (!ok || !closed) {
        const int e = ok ? errno : werr;
        remove(out_path);
        return fail(std::string("write sparse GRM file failed: ") + out_path + " (" + std::to_string(nnz) + " entries, " +
                    strerror(e) + ")");
    }
    if (out_n) *out_n = n;
    if (out_nnz) *out_nnz = (int64_t)nnz;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// grm_stream_bed_f32 on an in-memory payload (src/stats/grm.rs:1465-1536 prepare, 4690-5455 driver)
// ---------------------------------------------------------------------------------------------------
extern "C" int jx_grm_stream_payload_f32(const uint8_t *packed, int64_t m, int n_samples, int method,
                                         float maf_threshold, float max_missing_rate, float het_threshold,
                                         float *out_k, int64_t *out_eff_m, uint8_t *out_keep) {
    if (n_samples <= 0) return fail("n_samples must be > 0");
    if (method != 1 && method != 2) return fail("unsupported method=" + std::to_string(method));
    if (m <= 0) return fail("No SNPs remained after filtering; GRM is empty.");
    // grm.rs:4709-4711 threshold clamps
    const float maf_thr = fminf(fmaxf(maf_threshold, 0.0f), 0.5f);
    const float miss_thr = fminf(fmaxf(max_missing_rate, 0.0f), 1.0f);
    const float het_thr = fminf(fmaxf(het_threshold, 0.0f), 1.0f);
    SampleSel sel;
    make_sample_sel(nullptr, 0, n_samples, sel);
    DevBuf p32, dcnt;
    if (stage_p32(packed, m, n_samples, sel, p32)) return 1;
    if (dcnt.alloc(sizeof(int32_t) * 3 * (size_t)m)) return 1;
    if (jxg_row_counts_p32(p32.as<uint8_t>(), m, n_samples, dcnt.as<int32_t>(), nullptr)) return 1;
    std::vector<int32_t> cnt((size_t)m * 3);
    JX_HIP(hipMemcpy(cnt.data(), dcnt.p, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));

    std::vector<int32_t> rows;
    std::vector<float> lut;
    rows.reserve(m);
    lut.reserve((size_t)m * 4);
    double varsum = 0.0;
    const double eps = (double)1e-12f;
    for (int64_t j = 0; j < m; ++j) {
        if (out_keep) out_keep[j] = 0;
        const int64_t missing = cnt[j * 3], het = cnt[j * 3 + 1], hom = cnt[j * 3 + 2];
        const int64_t nm = n_samples - missing;
        if (het_thr > 0.0f && nm > 0) {
            const double het_rate = (double)het / (double)nm;
            if (het_rate > (double)het_thr) continue;
        }
        const double missing_rate = 1.0 - ((double)nm / (double)n_samples);
        if (missing_rate > (double)miss_thr) continue;
        float mean_g, sc;
        bool flip = false;
        double var = 0.0;
        if (nm == 0) {
            if (maf_thr > 0.0f) continue;
            mean_g = 0.0f;
            sc = (method == 2) ? 0.0f : 1.0f;
        } else {
            double alt_sum = (double)(het + 2 * hom);
            double alt_freq = alt_sum / (2.0 * (double)nm);
            flip = alt_freq > 0.5;
            if (flip) {
                alt_sum = 2.0 * (double)nm - alt_sum;
                alt_freq = alt_sum / (2.0 * (double)nm);
            }
            const double maf = fmin(alt_freq, 1.0 - alt_freq);
            if (maf < (double)maf_thr) continue;
            mean_g = (float)(alt_sum / (double)nm);
            var = fmax(2.0 * alt_freq * (1.0 - alt_freq), 0.0);
            sc = 1.0f;
            if (method == 2) sc = (var > eps) ? (float)(1.0 / sqrt(var)) : 0.0f;
        }
        if (out_keep) out_keep[j] = 1;
        rows.push_back((int32_t)j);
        varsum += var;
        const float g0 = flip ? 2.0f : 0.0f, g2 = flip ? 0.0f : 2.0f;
        // decode.rs:446-461 `apply_prepared_grm_stream_row_copy_f32`
        lut.push_back((g0 - mean_g) * sc);
        lut.push_back(0.0f);
        lut.push_back((1.0f - mean_g) * sc);
        lut.push_back((g2 - mean_g) * sc);
    }
    const int64_t eff = (int64_t)rows.size();
    if (out_eff_m) *out_eff_m = eff;
    if (eff == 0) return fail("No SNPs remained after filtering; GRM is empty.");
    const double D = (method == 1) ? varsum : (double)eff;
    if (!(isfinite(D) && D > 0.0)) return fail("invalid centered GRM denominator: sum(2p(1-p)) <= 0");

    DevBuf drows, dlut, acc, dout;
    if (drows.alloc(rows.size() * sizeof(int32_t))) return 1;
    JX_HIP(hipMemcpy(drows.p, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (dlut.alloc(lut.size() * sizeof(float))) return 1;
    JX_HIP(hipMemcpy(dlut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
    const int n = n_samples;
    const int64_t npad = (int64_t)num_tiles(n) * JXG_TILE;
    if (acc.alloc(sizeof(double) * (size_t)(npad * npad))) return 1;
    JX_HIP(hipMemset(acc.p, 0, sizeof(double) * (size_t)(npad * npad)));
    if (jxg_grm_accumulate(p32.as<uint8_t>(), m, n, drows.as<int32_t>(), dlut.as<float>(), eff, acc.as<double>(), 0, 0,
                           nullptr))
        return 1;
    if (dout.alloc(sizeof(float) * (size_t)n * (size_t)n)) return 1;
    if (jxg_grm_finalize(acc.as<double>(), n, 1.0 / D, dout.p, 0, nullptr)) return 1;
    JX_HIP(hipMemcpy(out_k, dout.p, sizeof(float) * (size_t)n * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// rust_eigh_from_array_f64 (src/math/eigh.rs:1621-1703)
// ---------------------------------------------------------------------------------------------------
extern "C" int jx_eigh_f64(const double *a, int n, double diag_shift, double *evals, double *evecs) {
    if (n <= 0) return fail("matrix must be non-empty");
    DevBuf da, dw, dt;
    const size_t nn = (size_t)n * (size_t)n;
    if (da.alloc(sizeof(double) * nn)) return 1;
    if (dw.alloc(sizeof(double) * (size_t)n)) return 1;
    JX_HIP(hipMemcpy(da.p, a, sizeof(double) * nn, hipMemcpyHostToDevice));
    if (launch_symmetrize(da.as<double>(), n, nullptr)) return 1;  // eigh.rs:179
    if (jxg_eigh_f64(da.as<double>(), n, diag_shift, dw.as<double>(), nullptr)) return 1;
    JX_HIP(hipMemcpy(evals, dw.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    if (evecs) {
        if (dt.alloc(sizeof(double) * nn)) return 1;
        if (launch_transpose_f64(da.as<double>(), dt.as<double>(), n, nullptr)) return 1;  // U^T -> U (columns = vectors)
        JX_HIP(hipMemcpy(evecs, dt.p, sizeof(double) * nn, hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int jx_lmm_rotate_x_y_with_ut_f64(const float *u_t, int n, const double *x, int q, const double *y,
                                             double *out_x, double *out_y) {
    if (n <= 0) return fail("y must not be empty");
    if (q < 0 || q > 15) return fail("x must have between 0 and 15 columns");
    const int qq = q + 1;
    std::vector<double> xy((size_t)n * qq);
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < q; ++c) xy[(size_t)i * qq + c] = x[(size_t)i * q + c];
        xy[(size_t)i * qq + q] = y[i];
    }
    DevBuf dut, dxy, dout;
    if (dut.alloc(sizeof(float) * (size_t)n * n)) return 1;
    if (dxy.alloc(sizeof(double) * xy.size())) return 1;
    if (dout.alloc(sizeof(double) * xy.size())) return 1;
    JX_HIP(hipMemcpy(dut.p, u_t, sizeof(float) * (size_t)n * n, hipMemcpyHostToDevice));
    JX_HIP(hipMemcpy(dxy.p, xy.data(), sizeof(double) * xy.size(), hipMemcpyHostToDevice));
    if (jxg_rotate_xy(dut.as<float>(), n, dxy.as<double>(), qq, dout.as<double>(), nullptr)) return 1;
    JX_HIP(hipMemcpy(xy.data(), dout.p, sizeof(double) * xy.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < q; ++c) out_x[(size_t)i * q + c] = xy[(size_t)i * qq + c];
        out_y[i] = xy[(size_t)i * qq + q];
    }
    return 0;
}

namespace {
struct NullDev {
    DevBuf s, x, y;
    int upload(const double *hs, const double *hx, const double *hy, int n, int p) {
        if (s.alloc(sizeof(double) * (size_t)n) || x.alloc(sizeof(double) * (size_t)n * p) ||
            y.alloc(sizeof(double) * (size_t)n))
            return 1;
        JX_HIP(hipMemcpy(s.p, hs, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
        JX_HIP(hipMemcpy(x.p, hx, sizeof(double) * (size_t)n * p, hipMemcpyHostToDevice));
        JX_HIP(hipMemcpy(y.p, hy, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
        return 0;
    }
};
}  // namespace

extern "C" int jx_lmm_reml_null(const double *s, const double *xcov, const double *y_rot, int n, int p, double low,
                                double high, int max_iter, double tol, double *out3) {
    if (low >= high) return fail("low must be < high");
    NullDev nd;
    DevBuf o;
    if (nd.upload(s, xcov, y_rot, n, p) || o.alloc(3 * sizeof(double))) return 1;
    if (jxg_lmm_reml_null(nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), n, p, low, high, max_iter, tol,
                          o.as<double>(), nullptr))
        return 1;
    JX_HIP(hipMemcpy(out3, o.p, 3 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int jx_ml_loglike_null(const double *s, const double *xcov, const double *y_rot, int n, int p,
                                  double log10_lbd, double *ml) {
    NullDev nd;
    if (nd.upload(s, xcov, y_rot, n, p)) return 1;
    DevBuf o;
    if (o.alloc(2 * sizeof(double))) return 1;
    if (jxg_lmm_loglike_null(nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), n, p, log10_lbd, o.as<double>(),
                             nullptr))
        return 1;
    double h[2];
    JX_HIP(hipMemcpy(h, o.p, sizeof(h), hipMemcpyDeviceToHost));
    *ml = h[0];
    return 0;
}

static const int64_t kBlockRows = 4096;  // rotate/scan block (rows x n f32 kept in HBM)

extern "C" int jx_lmm_reml_chunk(const double *s, const double *xcov, const double *y_rot, int n, int p, double low,
                                 double high, const float *snp_chunk, int64_t m_chunk, const float *u_t, int max_iter,
                                 double tol, int has_nullml, double nullml, double *out) {
    if (low >= high) return fail("low must be < high");
    if (m_chunk <= 0) return 0;
    const int cols = has_nullml ? 4 : 3;
    NullDev nd;
    if (nd.upload(s, xcov, y_rot, n, p)) return 1;
    DevBuf dut, dg, drot, dout;
    if (u_t) {
        if (dut.alloc(sizeof(float) * (size_t)n * n)) return 1;
        JX_HIP(hipMemcpy(dut.p, u_t, sizeof(float) * (size_t)n * n, hipMemcpyHostToDevice));
        if (drot.alloc(sizeof(float) * (size_t)kBlockRows * n)) return 1;
    }
    if (dg.alloc(sizeof(float) * (size_t)kBlockRows * n)) return 1;
    if (dout.alloc(sizeof(double) * (size_t)kBlockRows * cols)) return 1;
    for (int64_t r0 = 0; r0 < m_chunk; r0 += kBlockRows) {
        const int rows = (int)std::min<int64_t>(kBlockRows, m_chunk - r0);
        JX_HIP(hipMemcpy(dg.p, snp_chunk + (size_t)r0 * n, sizeof(float) * (size_t)rows * n, hipMemcpyHostToDevice));
        const float *grot = dg.as<float>();
        if (u_t) {
            if (jxg_rotate_dense_f32(dg.as<float>(), rows, n, dut.as<float>(), drot.as<float>(), nullptr)) return 1;
            grot = drot.as<float>();
        }
        if (jxg_lmm_scan(grot, rows, n, nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), p, low, high, tol,
                         max_iter, 0, 0.0, has_nullml, nullml, dout.as<double>(), nullptr, nullptr))
            return 1;
        JX_HIP(hipMemcpy(out + (size_t)r0 * cols, dout.p, sizeof(double) * (size_t)rows * cols, hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int jx_lmm2_chunk(const double *s, const double *xcov, const double *y_rot, int n, int p, double low,
                             double high, const float *snp_chunk, int64_t m_chunk, const float *u_t, double nullml,
                             int max_iter, double tol, double *out) {
    if (low >= high) return fail("low must be < high");
    if (!std::isfinite(nullml)) return fail("nullml must be finite");
    if (m_chunk <= 0) return 0;
    NullDev nd;
    if (nd.upload(s, xcov, y_rot, n, p)) return 1;
    DevBuf dut, dg, drot, dout;
    if (u_t) {
        if (dut.alloc(sizeof(float) * (size_t)n * n)) return 1;
        JX_HIP(hipMemcpy(dut.p, u_t, sizeof(float) * (size_t)n * n, hipMemcpyHostToDevice));
        if (drot.alloc(sizeof(float) * (size_t)kBlockRows * n)) return 1;
    }
    if (dg.alloc(sizeof(float) * (size_t)kBlockRows * n)) return 1;
    if (dout.alloc(sizeof(double) * (size_t)kBlockRows * 6)) return 1;
    for (int64_t r0 = 0; r0 < m_chunk; r0 += kBlockRows) {
        const int rows = (int)std::min<int64_t>(kBlockRows, m_chunk - r0);
        JX_HIP(hipMemcpy(dg.p, snp_chunk + (size_t)r0 * n, sizeof(float) * (size_t)rows * n, hipMemcpyHostToDevice));
        const float *grot = dg.as<float>();
        if (u_t) {
            if (jxg_rotate_dense_f32(dg.as<float>(), rows, n, dut.as<float>(), drot.as<float>(), nullptr)) return 1;
            grot = drot.as<float>();
        }
        if (jxg_lmm2_scan(grot, rows, n, nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), p, low, high, tol,
                          max_iter, 0, 0.0, nullml, dout.as<double>(), nullptr))
            return 1;
        JX_HIP(hipMemcpy(out + (size_t)r0 * 6, dout.p, sizeof(double) * (size_t)rows * 6, hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int jx_lmm2_null_ml(const double *s, const double *xcov, const double *y_rot, int n, int p, double low,
                               double high, int max_iter, double tol, int has_init, double init, double *out2) {
    if (low >= high) return fail("low must be < high");
    NullDev nd;
    if (nd.upload(s, xcov, y_rot, n, p)) return 1;
    DevBuf o;
    if (o.alloc(2 * sizeof(double))) return 1;
    if (jxg_lmm2_null_ml(nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), n, p, low, high, max_iter, tol,
                         has_init, init, o.as<double>(), nullptr))
        return 1;
    JX_HIP(hipMemcpy(out2, o.p, 2 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

namespace {
struct FvDev {
    DevBuf w, py, wx;
    std::vector<double> a_chol;
    double sc[3];
    int prepare(const NullDev &nd, int n, int p, double lbd) {
        if (w.alloc(sizeof(float) * (size_t)n) || py.alloc(sizeof(float) * (size_t)n) ||
            wx.alloc(sizeof(float) * (size_t)n * p))
            return 1;
        a_chol.assign((size_t)p * p, 0.0);
        return jxg_fvlmm_prepare(nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), n, p, lbd, w.as<float>(),
                                 py.as<float>(), wx.as<float>(), a_chol.data(), sc);
    }
};
}  // namespace

extern "C" int jx_fvlmm_assoc_chunk(const double *s, const double *xcov, const double *y_rot, int n, int p,
                                    double log10_lbd, const float *snp_chunk, int64_t m_chunk, const float *u_t,
                                    int has_nullml, double nullml, double *out) {
    if (m_chunk <= 0) return 0;
    const int cols = has_nullml ? 4 : 3;
    NullDev nd;
    if (nd.upload(s, xcov, y_rot, n, p)) return 1;
    FvDev fv;
    if (fv.prepare(nd, n, p, pow(10.0, log10_lbd))) return 1;
    DevBuf dut, dg, drot, dout;
    if (u_t) {
        if (dut.alloc(sizeof(float) * (size_t)n * n)) return 1;
        JX_HIP(hipMemcpy(dut.p, u_t, sizeof(float) * (size_t)n * n, hipMemcpyHostToDevice));
        if (drot.alloc(sizeof(float) * (size_t)kBlockRows * n)) return 1;
    }
    if (dg.alloc(sizeof(float) * (size_t)kBlockRows * n)) return 1;
    if (dout.alloc(sizeof(double) * (size_t)kBlockRows * cols)) return 1;
    for (int64_t r0 = 0; r0 < m_chunk; r0 += kBlockRows) {
        const int rows = (int)std::min<int64_t>(kBlockRows, m_chunk - r0);
        JX_HIP(hipMemcpy(dg.p, snp_chunk + (size_t)r0 * n, sizeof(float) * (size_t)rows * n, hipMemcpyHostToDevice));
        const float *grot = dg.as<float>();
        if (u_t) {
            if (jxg_rotate_dense_f32(dg.as<float>(), rows, n, dut.as<float>(), drot.as<float>(), nullptr)) return 1;
            grot = drot.as<float>();
        }
        if (jxg_fvlmm_scan(grot, rows, n, p, fv.w.as<float>(), fv.py.as<float>(), fv.wx.as<float>(), fv.a_chol.data(),
                           fv.sc[0], (int)fv.sc[2], has_nullml, nullml, fv.sc[1], dout.as<double>(), nullptr))
            return 1;
        JX_HIP(hipMemcpy(out + (size_t)r0 * cols, dout.p, sizeof(double) * (size_t)rows * cols, hipMemcpyDeviceToHost));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// lmm_reml_assoc_packed_f32 (src/stats/lmm.rs:3040-3362) / fixed-lambda sibling
// ---------------------------------------------------------------------------------------------------
extern "C" int jx_assoc_packed_gm(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                                  const float *row_maf, const double *s, const double *xcov, const double *y_rot,
                                  const float *u_t, int p, const int64_t *sample_indices, int n_sel, int model,
                                  double low, double high, int max_iter, double tol, int warm, double init_log10_lbd,
                                  int has_nullml, double nullml, double *out, int genetic_model);
extern "C" int jx_assoc_packed(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                               const float *row_maf, const double *s, const double *xcov, const double *y_rot,
                               const float *u_t, int p, const int64_t *sample_indices, int n_sel, int model,
                               double low, double high, int max_iter, double tol, int warm, double init_log10_lbd,
                               int has_nullml, double nullml, double *out) {
    return jx_assoc_packed_gm(packed, m, n_samples, row_flip, row_maf, s, xcov, y_rot, u_t, p, sample_indices, n_sel, model, low,
                              high, max_iter, tol, warm, init_log10_lbd, has_nullml, nullml, out, 0);
}
static int assoc_packed_impl(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                             const float *row_maf, const double *s, const double *xcov, const double *y_rot,
                             const float *u_t, int p, const int64_t *sample_indices, int n_sel, int model,
                             double low, double high, int max_iter, double tol, int warm, double init_log10_lbd,
                             int has_nullml, double nullml, double *out, int genetic_model, const int64_t *chain_off,
                             int64_t n_chains);
extern "C" int jx_assoc_packed_gm(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                                  const float *row_maf, const double *s, const double *xcov, const double *y_rot,
                                  const float *u_t, int p, const int64_t *sample_indices, int n_sel, int model,
                                  double low, double high, int max_iter, double tol, int warm, double init_log10_lbd,
                                  int has_nullml, double nullml, double *out, int genetic_model) {
    return assoc_packed_impl(packed, m, n_samples, row_flip, row_maf, s, xcov, y_rot, u_t, p, sample_indices, n_sel, model, low,
                             high, max_iter, tol, warm, init_log10_lbd, has_nullml, nullml, out, genetic_model, nullptr, 0);
}
// The exact scan (model 0) along the reference's warm-start chains (`carry_warm_start`, src/stats/lmm.rs:134-161: on in
// `lmm_reml_assoc_packed_f32` :3244-3245 and, unless JX_LMM_UNIFIED_NO_WARM_START is set, in the BED route :2627): chain c is the
// rows [chain_off[c], chain_off[c + 1]) of the payload in order (chain_off[0] = 0, ascending, chain_off[n_chains] = m); the
// first valid SNP of a chain starts from init_log10_lbd (warm != 0) or the interval midpoint, every later one from the optimum
// of the valid SNP before it.  The chains run in parallel, the rows of a chain in sequence.
extern "C" int jx_assoc_packed_chain(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                                     const float *row_maf, const double *s, const double *xcov, const double *y_rot,
                                     const float *u_t, int p, const int64_t *sample_indices, int n_sel, double low, double high,
                                     int max_iter, double tol, int warm, double init_log10_lbd, int has_nullml, double nullml,
                                     double *out, int genetic_model, const int64_t *chain_off, int64_t n_chains) {
    if (m > 0) {
        if (!chain_off || n_chains <= 0) return fail("jx_assoc_packed_chain: chain offsets are required");
        if (chain_off[0] != 0 || chain_off[n_chains] != m) return fail("jx_assoc_packed_chain: chain offsets must run from 0 to m");
        for (int64_t c = 0; c < n_chains; ++c)
            if (chain_off[c + 1] < chain_off[c]) return fail("jx_assoc_packed_chain: chain offsets must ascend");
    }
    return assoc_packed_impl(packed, m, n_samples, row_flip, row_maf, s, xcov, y_rot, u_t, p, sample_indices, n_sel, 0, low, high,
                             max_iter, tol, warm, init_log10_lbd, has_nullml, nullml, out, genetic_model, chain_off, n_chains);
}
static int assoc_packed_impl(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                             const float *row_maf, const double *s, const double *xcov, const double *y_rot,
                             const float *u_t, int p, const int64_t *sample_indices, int n_sel, int model,
                             double low, double high, int max_iter, double tol, int warm, double init_log10_lbd,
                             int has_nullml, double nullml, double *out, int genetic_model, const int64_t *chain_off,
                             int64_t n_chains) {
    if (genetic_model < 0 || genetic_model > 3) return fail("model must be one of: add, dom, rec, het");
    const int cols = model == 2 ? 6 : (has_nullml ? 4 : 3);
    if (model < 0 || model > 2) return fail("model must be 0 (lmm), 1 (fvlmm) or 2 (lmm2)");
    if (model == 2 && !(has_nullml && std::isfinite(nullml))) return fail("nullml must be finite");
    if (n_samples <= 0) return fail("n_samples must be > 0");
    if (model != 1 && low >= high) return fail("low must be < high");
    if (model != 1 && !(isfinite(tol) && tol > 0.0)) return fail("tol must be positive and finite");
    if (m <= 0) return 0;
    SampleSel sel;
    if (make_sample_sel(sample_indices, n_sel, n_samples, sel)) return 1;
    const int n = sel.n;
    DevBuf p32, dcnt;
    if (stage_p32(packed, m, n_samples, sel, p32)) return 1;
    if (dcnt.alloc(sizeof(int32_t) * 3 * (size_t)m)) return 1;
    if (jxg_row_counts_p32(p32.as<uint8_t>(), m, n, dcnt.as<int32_t>(), nullptr)) return 1;
    std::vector<int32_t> cnt((size_t)m * 3);
    JX_HIP(hipMemcpy(cnt.data(), dcnt.p, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<float> lut((size_t)m * 4);
    for (int64_t j = 0; j < m; ++j)
        scan_lut_from_counts(row_maf[j], row_flip[j] != 0, &cnt[(size_t)j * 3], n, &lut[(size_t)j * 4], genetic_model);

    NullDev nd;
    if (nd.upload(s, xcov, y_rot, n, p)) return 1;
    FvDev fv;
    if (model == 1 && fv.prepare(nd, n, p, pow(10.0, low))) return 1;

    const int64_t npad = (int64_t)num_tiles(n) * JXG_TILE;
    DevBuf dut, uhi, ulo, dlut, drot, dout;
    if (dut.alloc(sizeof(float) * (size_t)n * n)) return 1;
    JX_HIP(hipMemcpy(dut.p, u_t, sizeof(float) * (size_t)n * n, hipMemcpyHostToDevice));
    if (uhi.alloc(sizeof(uint16_t) * (size_t)(npad * npad)) || ulo.alloc(sizeof(uint16_t) * (size_t)(npad * npad)))
        return 1;
    const int scale_exp = 10;
    if (jxg_ut_split(dut.as<float>(), n, uhi.as<uint16_t>(), ulo.as<uint16_t>(), scale_exp, nullptr)) return 1;
    DevBuf dusum, dlut16, drowoff;
    if (dusum.alloc(sizeof(float) * (size_t)npad)) return 1;
    if (jxg_ut_rowsum(dut.as<float>(), n, dusum.as<float>(), nullptr)) return 1;
    JX_HIP(hipDeviceSynchronize());
    dut.release();
    if (dlut.alloc(lut.size() * sizeof(float))) return 1;
    JX_HIP(hipMemcpy(dlut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
    // fp16 LUT records once for all rows; rows without missing calls as integer LUT + offset (exact-row rotation)
    if (dlut16.alloc((size_t)16 * (size_t)m) || drowoff.alloc(sizeof(float) * (size_t)m)) return 1;
    const int64_t brows = 8192;
    // From n = 4096 the rows of a block are dealt to the int8 rotation (design rows without a missing call: three int8 planes
    // of U, k_rotate_i8.hip) and the 256-tile fp16 kernel by position lists, as pipeline.scan_rows does (JXGPU_ROT_I8=0: off)
    static const bool q_env = !(getenv("JXGPU_ROT_I8") && atoi(getenv("JXGPU_ROT_I8")) == 0);
    const int fused_mode = getenv("JXGPU_FVLMM_FUSED") ? atoi(getenv("JXGPU_FVLMM_FUSED")) : 1;
    const bool use_q = n >= 4096 && q_env && !(model == 1 && p <= 8 && fused_mode == 2);
    // fixed lambda below that size: the rotation kernel's fused epilogue reduces the tile in place, G~ is never written
    const bool fused = model == 1 && p <= 8 && fused_mode != 0 && !use_q;
    // rows with a few missing calls keep the exact rotation where the int8 kernel runs; their missing-call term is added behind
    // it (jxg_rotate_missing_correct), exactly as pipeline.scan_rows does
    DevBuf drowmiss, dusamp;
    // mean missing calls per row over the rows that can pass a quality filter (<= n / 10 missing calls), as
    // pipeline.Panel.mean_missing does: mostly-missing junk rows must not decide the path of the others
    double miss_sum = 0.0;
    int64_t miss_rows = 0;
    for (int64_t j = 0; j < m; ++j) {
        const double mi = (double)cnt[(size_t)j * 3];
        if (mi <= (double)n / 10.0) {
            miss_sum += mi;
            ++miss_rows;
        }
    }
    const int miss_max = use_q ? jxg_rot_miss_max(n, miss_rows > 0 ? miss_sum / (double)miss_rows : 0.0) : 0;
    bool any_rowmiss = false;
    if (miss_max > 0) {
        if (drowmiss.alloc(sizeof(float) * (size_t)m)) return 1;
        JX_HIP(hipMemset(drowmiss.p, 0, sizeof(float) * (size_t)m));
    }
    if (jxg_lut_split_rows_m(p32.as<uint8_t>(), m, n, nullptr, dlut.as<float>(), m, dlut16.p, drowoff.as<float>(),
                             miss_max > 0 ? drowmiss.as<float>() : nullptr, miss_max, nullptr))
        return 1;
    // beyond n / 300 missing calls per row (limit > 256 = none): the missing-call term as one more int8 product
    // (jxg_rotate_missing_dense) over the rows that have one, instead of the gather form
    const bool miss_dense = miss_max > 256;
    std::vector<float> hm;
    if (miss_max > 0) {
        hm.resize((size_t)m);
        JX_HIP(hipMemcpy(hm.data(), drowmiss.p, sizeof(float) * (size_t)m, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < m && !any_rowmiss; ++i) any_rowmiss = hm[(size_t)i] != 0.0f;
    }
    DevBuf dq, dumax, dsel, dselm;
    std::vector<float> hrowoff;
    std::vector<int32_t> hsel;
    if (use_q) {
        if (dq.alloc((size_t)3 * (size_t)(npad * npad)) || dumax.alloc(sizeof(float) * (size_t)npad)) return 1;
        DevBuf dut2;
        if (dut2.alloc(sizeof(float) * (size_t)n * n)) return 1;
        JX_HIP(hipMemcpy(dut2.p, u_t, sizeof(float) * (size_t)n * n, hipMemcpyHostToDevice));
        if (jxg_ut_quant3(dut2.as<float>(), n, dq.as<int8_t>(), dumax.as<float>(), nullptr)) return 1;
        if (any_rowmiss && !miss_dense) {
            if (dusamp.alloc(sizeof(float) * (size_t)n * n)) return 1;
            if (jxg_transpose_f32(dut2.as<float>(), n, dusamp.as<float>(), nullptr)) return 1;
        }
        JX_HIP(hipDeviceSynchronize());
        hrowoff.resize((size_t)m);
        JX_HIP(hipMemcpy(hrowoff.data(), drowoff.p, sizeof(float) * (size_t)m, hipMemcpyDeviceToHost));
        if (dsel.alloc(sizeof(int32_t) * (size_t)brows)) return 1;
        hsel.resize((size_t)brows);
        if (any_rowmiss && miss_dense && dselm.alloc(sizeof(int32_t) * (size_t)brows)) return 1;
    }
    DevBuf dsums, dachol;
    if (fused) {
        if (dsums.alloc(sizeof(double) * (size_t)num_tiles(n) * brows * (p + 2)) || dachol.alloc(sizeof(double) * (size_t)p * p))
            return 1;
        JX_HIP(hipMemcpy(dachol.p, fv.a_chol.data(), sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice));
    } else if (drot.alloc(sizeof(float) * (size_t)brows * n)) {
        return 1;
    }
    if (dout.alloc(sizeof(double) * (size_t)brows * cols)) return 1;
    DevBuf drows;
    if (drows.alloc(sizeof(int32_t) * (size_t)brows)) return 1;
    std::vector<int32_t> hrows((size_t)brows);
    // ---- warm-start chains (model 0) ------------------------------------------------------------------------------------
    // carry: one state per chain (NaN: none).  With the series form (k_scan_fast.hip) the series of a SUPER-BLOCK of rows are kept
    // and ONE Brent launch walks every chain that touches it -- a block of 8192 rows holds 16 chains of 512 rows, far too few
    // sequential jobs for 256 CUs; without it (wide bounds, many covariates) the chains are walked block by block.  A chain cut
    // by a block or super-block boundary continues from its carry state.
    const bool chain = chain_off != nullptr && model == 0;
    DevBuf dcarry, dchoff, dtab, dscoef, dsssq, dout_sb;
    std::vector<int32_t> hchoff;
    int64_t sd = 0, sb_rows = 0, sb0 = 0;            // series doubles per row; rows per super-block; first row of the open one
    if (chain) {
        if (dcarry.alloc(sizeof(double) * (size_t)n_chains)) return 1;
        std::vector<double> hc((size_t)n_chains, (warm && std::isfinite(init_log10_lbd)) ? init_log10_lbd : std::nan(""));
        JX_HIP(hipMemcpy(dcarry.p, hc.data(), sizeof(double) * (size_t)n_chains, hipMemcpyHostToDevice));
        const int64_t tb = jxg_lmm_tables_bytes(n, p, low, high);
        sd = tb > 0 ? jxg_lmm_series_doubles(p, low, high) : 0;
        if (sd > 0) {
            if (dtab.alloc((size_t)tb)) return 1;
            if (jxg_lmm_tables_build(nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), n, p, low, high, dtab.p, nullptr))
                return 1;
            const int64_t cap = ((int64_t)8 << 30) / (8 * (sd + 1 + cols));
            sb_rows = std::min<int64_t>(m, std::max<int64_t>(brows, cap / brows * brows));
            if (dscoef.alloc(sizeof(double) * (size_t)(sb_rows * sd)) || dsssq.alloc(sizeof(double) * (size_t)sb_rows) ||
                dout_sb.alloc(sizeof(double) * (size_t)(sb_rows * cols)))
                return 1;
        }
        if (dchoff.alloc(sizeof(int32_t) * (size_t)(std::min<int64_t>(n_chains, sd > 0 ? sb_rows : brows) + 2))) return 1;
    }
    // chains touching the rows [a, b): local offsets into hchoff, -> index of the first one
    auto chain_segments = [&](int64_t a, int64_t b, int64_t &c_first) -> int {
        const int64_t *lo = std::upper_bound(chain_off, chain_off + n_chains + 1, a) - 1;       // last offset <= a
        c_first = lo - chain_off;
        if (c_first >= n_chains) c_first = n_chains - 1;
        hchoff.clear();
        int64_t c = c_first;
        for (; c < n_chains && chain_off[c] < b; ++c)
            hchoff.push_back((int32_t)(std::max<int64_t>(chain_off[c], a) - a));
        hchoff.push_back((int32_t)(std::min<int64_t>(chain_off[c], b) - a));
        return (int)(c - c_first);
    };
    auto chain_brent_superblock = [&](int64_t a, int64_t b) -> int {
        int64_t c_first = 0;
        const int nch = chain_segments(a, b, c_first);
        if ((size_t)(nch + 1) * sizeof(int32_t) > dchoff.bytes && dchoff.alloc(sizeof(int32_t) * (size_t)(nch + 1))) return 1;
        JX_HIP(hipMemcpy(dchoff.p, hchoff.data(), sizeof(int32_t) * (size_t)(nch + 1), hipMemcpyHostToDevice));
        if (jxg_lmm_series_brent_tab((int)(b - a), n, nd.s.as<double>(), nd.x.as<double>(), p, low, high, dtab.p, tol, max_iter, 0,
                                     0.0, dscoef.as<double>(), dsssq.as<double>(), dchoff.as<int32_t>(), nch,
                                     dcarry.as<double>() + c_first, has_nullml, nullml, dout_sb.as<double>(), nullptr, nullptr))
            return 1;
        JX_HIP(hipMemcpy(out + (size_t)a * cols, dout_sb.p, sizeof(double) * (size_t)(b - a) * cols, hipMemcpyDeviceToHost));
        return 0;
    };
    ProgressTicker ticker;
    for (int64_t r0 = 0; r0 < m; r0 += brows) {
        const int rows = (int)std::min<int64_t>(brows, m - r0);
        for (int i = 0; i < rows; ++i) hrows[i] = (int32_t)(r0 + i);
        JX_HIP(hipMemcpy(drows.p, hrows.data(), sizeof(int32_t) * (size_t)rows, hipMemcpyHostToDevice));
        if (fused) {
            if (jxg_rotate_packed16x_fused(p32.as<uint8_t>(), m, n, drows.as<int32_t>(), rows,
                                           (const uint8_t *)dlut16.p + (size_t)r0 * 16, drowoff.as<float>() + r0,
                                           dusum.as<float>(), uhi.as<uint16_t>(), ulo.as<uint16_t>(), scale_exp,
                                           fv.w.as<float>(), fv.py.as<float>(), fv.wx.as<float>(), p, dsums.as<double>(),
                                           p + 2, 0, nullptr) ||
                jxg_fvlmm_finish_dev(dsums.as<double>(), num_tiles(n), p + 2, rows, n, p, dachol.as<double>(), fv.sc[0], (int)fv.sc[2],
                                     has_nullml, nullml, fv.sc[1], 0, dout.as<double>(), nullptr))
                return 1;
        } else if (use_q) {
            // positions of the exact rows (finite row offset) first, the others behind them
            int ne = 0;
            for (int i = 0; i < rows; ++i)
                if (!std::isnan(hrowoff[(size_t)r0 + i])) hsel[ne++] = i;
            int nx = ne;
            for (int i = 0; i < rows; ++i)
                if (std::isnan(hrowoff[(size_t)r0 + i])) hsel[nx++] = i;
            JX_HIP(hipMemcpy(dsel.p, hsel.data(), sizeof(int32_t) * (size_t)rows, hipMemcpyHostToDevice));
            if (jxg_rotate_packed16x_q(p32.as<uint8_t>(), m, n, drows.as<int32_t>(), rows,
                                       (const uint8_t *)dlut16.p + (size_t)r0 * 16, drowoff.as<float>() + r0, dusum.as<float>(),
                                       uhi.as<uint16_t>(), ulo.as<uint16_t>(), scale_exp, dq.as<int8_t>(), dumax.as<float>(),
                                       ne > 0 ? dsel.as<int32_t>() : nullptr, ne, rows > ne ? dsel.as<int32_t>() + ne : nullptr,
                                       rows - ne, drot.as<float>(), nullptr))
                return 1;
            if (any_rowmiss && miss_dense) {
                int nm = 0;
                for (int i = 0; i < rows; ++i)
                    if (hm[(size_t)r0 + i] != 0.0f) hsel[nm++] = i;
                if (nm > 0) {
                    JX_HIP(hipMemcpy(dselm.p, hsel.data(), sizeof(int32_t) * (size_t)nm, hipMemcpyHostToDevice));
                    if (jxg_rotate_missing_dense(p32.as<uint8_t>(), m, n, drows.as<int32_t>(), dselm.as<int32_t>(), nm,
                                                 drowmiss.as<float>() + r0, dq.as<int8_t>(), dumax.as<float>(), drot.as<float>(), n,
                                                 nullptr))
                        return 1;
                }
            } else if (any_rowmiss && jxg_rotate_missing_correct(p32.as<uint8_t>(), m, n, drows.as<int32_t>(), rows,
                                                                 drowmiss.as<float>() + r0, dusamp.as<float>(), drot.as<float>(), n,
                                                                 nullptr))
                return 1;
        } else if (jxg_rotate_packed16x(p32.as<uint8_t>(), m, n, drows.as<int32_t>(), rows,
                                        (const uint8_t *)dlut16.p + (size_t)r0 * 16, drowoff.as<float>() + r0,
                                        dusum.as<float>(), uhi.as<uint16_t>(), ulo.as<uint16_t>(), scale_exp,
                                        drot.as<float>(), nullptr)) {
            return 1;
        }
        if (fused) {
        } else if (chain && sd > 0) {
            // series of this block behind those of the open super-block; Brent when the super-block is full or the payload ends
            if (jxg_lmm_series_coef_tab(drot.as<float>(), rows, n, nd.x.as<double>(), p, low, high, dtab.p,
                                        dscoef.as<double>() + (size_t)(r0 - sb0) * sd, dsssq.as<double>() + (r0 - sb0), nullptr))
                return 1;
            if (r0 + rows - sb0 >= sb_rows || r0 + rows >= m) {
                if (chain_brent_superblock(sb0, r0 + rows)) return 1;
                sb0 = r0 + rows;
            }
            if (ticker.tick(r0 + rows, m, brows)) return fail("interrupted by the progress callback");
            continue;
        } else if (chain) {
            int64_t c_first = 0;
            const int nch = chain_segments(r0, r0 + rows, c_first);
            if ((size_t)(nch + 1) * sizeof(int32_t) > dchoff.bytes && dchoff.alloc(sizeof(int32_t) * (size_t)(nch + 1))) return 1;
            JX_HIP(hipMemcpy(dchoff.p, hchoff.data(), sizeof(int32_t) * (size_t)(nch + 1), hipMemcpyHostToDevice));
            if (jxg_lmm_scan_chain(drot.as<float>(), rows, n, nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), p, low, high,
                                   tol, max_iter, dchoff.as<int32_t>(), nch, dcarry.as<double>() + c_first, has_nullml, nullml,
                                   dout.as<double>(), nullptr, nullptr))
                return 1;
        } else if (model == 0) {
            if (jxg_lmm_scan(drot.as<float>(), rows, n, nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), p, low,
                             high, tol, max_iter, warm, init_log10_lbd, has_nullml, nullml, dout.as<double>(), nullptr,
                             nullptr))
                return 1;
        } else if (model == 2) {
            if (jxg_lmm2_scan(drot.as<float>(), rows, n, nd.s.as<double>(), nd.x.as<double>(), nd.y.as<double>(), p, low,
                              high, tol, max_iter, warm, init_log10_lbd, nullml, dout.as<double>(), nullptr))
                return 1;
        } else {
            if (jxg_fvlmm_scan(drot.as<float>(), rows, n, p, fv.w.as<float>(), fv.py.as<float>(), fv.wx.as<float>(),
                               fv.a_chol.data(), fv.sc[0], (int)fv.sc[2], has_nullml, nullml, fv.sc[1],
                               dout.as<double>(), nullptr))
                return 1;
        }
        JX_HIP(hipMemcpy(out + (size_t)r0 * cols, dout.p, sizeof(double) * (size_t)rows * cols, hipMemcpyDeviceToHost));
        if (ticker.tick(r0 + rows, m, brows)) return fail("interrupted by the progress callback");
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------------
// lm_block_assoc_packed (src/stats/glm.rs:3550-3860): the plain LM scan the mixed-model routes fall back to
// ---------------------------------------------------------------------------------------------------
extern "C" int jxg_lm_scan_p32(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                               const float *d_lut, const double *d_xr, int q0, const double *d_ixx, double yy_r,
                               double *d_work, double *d_out, void *stream);

// Host half shared with the resident-panel route (janusx_amd/pipeline.py): r_y = y - X (C X'y), y'M_X y, and [X | r_y]
// rounded through f32 (glm.rs:3635-3672).  xr_out (n, q0 + 1).
extern "C" int jx_lm_residualize(const double *y, const double *x, const double *ixx, int n, int q0, double *xr_out,
                                 double *yy_r_out) {
    if (n <= 0 || q0 < 0) return fail("X.n_rows must equal len(y)");
    if (n <= q0 + 1) return fail("n too small: require n > q0+1, got n=" + std::to_string(n) + ", q0=" + std::to_string(q0));
    std::vector<double> xy((size_t)q0, 0.0), cxy((size_t)q0, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < q0; ++j) xy[j] += x[(size_t)i * q0 + j] * y[i];
    for (int i = 0; i < q0; ++i) {
        double acc = 0.0;
        for (int j = 0; j < q0; ++j) acc += ixx[(size_t)i * q0 + j] * xy[j];
        cxy[i] = acc;
    }
    double yy = 0.0;
    const int ld = q0 + 1;
    for (int i = 0; i < n; ++i) {
        double pred = 0.0;
        for (int j = 0; j < q0; ++j) pred += x[(size_t)i * q0 + j] * cxy[j];
        const double resid = y[i] - pred;
        yy += resid * resid;
        for (int j = 0; j < q0; ++j) xr_out[(size_t)i * ld + j] = (double)(float)x[(size_t)i * q0 + j];
        xr_out[(size_t)i * ld + q0] = (double)(float)resid;
    }
    *yy_r_out = yy;
    return 0;
}

extern "C" int jx_lm_assoc_packed(const double *y, const double *x, const double *ixx, int q0, const uint8_t *packed,
                                  int64_t m, int n_samples, const uint8_t *row_flip, const float *row_maf,
                                  const int64_t *sample_indices, int n_sel, double *out) {
    if (n_samples <= 0) return fail("n_samples must be > 0");
    if (m <= 0) return 0;
    SampleSel sel;
    if (make_sample_sel(sample_indices, n_sel, n_samples, sel)) return 1;
    const int n = sel.n;
    std::vector<double> xr((size_t)n * (q0 + 1));
    double yy_r = 0.0;
    if (jx_lm_residualize(y, x, ixx, n, q0, xr.data(), &yy_r)) return 1;
    DevBuf p32;
    if (stage_p32(packed, m, n_samples, sel, p32)) return 1;
    std::vector<float> lut((size_t)m * 4);
    for (int64_t j = 0; j < m; ++j) {                       // `decode_mean_imputed_additive_packed_block_rows_f32`,
        const float mean_g = std::min(std::max(2.0f * row_maf[j], 0.0f), 2.0f);   // src/math/bedmath.rs:984-989
        float *l = &lut[(size_t)j * 4];
        if (row_flip[j]) l[0] = 2.0f, l[1] = mean_g, l[2] = 1.0f, l[3] = 0.0f;
        else l[0] = 0.0f, l[1] = mean_g, l[2] = 1.0f, l[3] = 2.0f;
    }
    DevBuf dlut, dxr, dixx, dwork, dout;
    if (dlut.alloc(lut.size() * sizeof(float)) || dxr.alloc(xr.size() * sizeof(double)) ||
        dixx.alloc(sizeof(double) * (size_t)std::max(q0 * q0, 1)))
        return 1;
    JX_HIP(hipMemcpy(dlut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
    JX_HIP(hipMemcpy(dxr.p, xr.data(), xr.size() * sizeof(double), hipMemcpyHostToDevice));
    if (q0 > 0) JX_HIP(hipMemcpy(dixx.p, ixx, sizeof(double) * (size_t)q0 * q0, hipMemcpyHostToDevice));
    const int64_t brows = 1 << 20;
    if (dwork.alloc(sizeof(double) * (size_t)std::min(brows, m) * (q0 + 2)) ||
        dout.alloc(sizeof(double) * (size_t)std::min(brows, m) * 4))
        return 1;
    DevBuf drows;
    if (drows.alloc(sizeof(int32_t) * (size_t)std::min(brows, m))) return 1;
    std::vector<int32_t> hrows((size_t)std::min(brows, m));
    ProgressTicker ticker;
    for (int64_t r0 = 0; r0 < m; r0 += brows) {
        const int rows = (int)std::min<int64_t>(brows, m - r0);
        for (int i = 0; i < rows; ++i) hrows[i] = (int32_t)(r0 + i);
        JX_HIP(hipMemcpy(drows.p, hrows.data(), sizeof(int32_t) * (size_t)rows, hipMemcpyHostToDevice));
        if (jxg_lm_scan_p32(p32.as<uint8_t>(), m, n, drows.as<int32_t>(), rows, dlut.as<float>() + (size_t)r0 * 4,
                            dxr.as<double>(), q0, dixx.as<double>(), yy_r, dwork.as<double>(), dout.as<double>(), nullptr))
            return 1;
        JX_HIP(hipMemcpy(out + (size_t)r0 * 4, dout.p, sizeof(double) * (size_t)rows * 4, hipMemcpyDeviceToHost));
        if (ticker.tick(r0 + rows, m, brows)) return fail("interrupted by the progress callback");
    }
    return 0;
}

// `lm_block_assoc_f32` (src/stats/glm.rs:4313-4497): dense SNP-major f32 rows on the host, blocks staged to the device.
extern "C" int jx_lm_assoc_dense(const double *y, const double *x, const double *ixx, int q0, const float *g, int64_t m, int n,
                                 double *out) {
    if (m <= 0) return 0;
    std::vector<double> xr((size_t)n * (q0 + 1));
    double yy_r = 0.0;
    if (jx_lm_residualize(y, x, ixx, n, q0, xr.data(), &yy_r)) return 1;
    DevBuf dxr, dixx, dwork, dout, dg;
    if (dxr.alloc(xr.size() * sizeof(double)) || dixx.alloc(sizeof(double) * (size_t)std::max(q0 * q0, 1))) return 1;
    JX_HIP(hipMemcpy(dxr.p, xr.data(), xr.size() * sizeof(double), hipMemcpyHostToDevice));
    if (q0 > 0) JX_HIP(hipMemcpy(dixx.p, ixx, sizeof(double) * (size_t)q0 * q0, hipMemcpyHostToDevice));
    const int64_t brows = std::max<int64_t>(1, std::min<int64_t>(m, ((int64_t)1 << 30) / ((int64_t)n * 4)));   // <= 1 GiB of rows
    if (dg.alloc(sizeof(float) * (size_t)brows * n) || dwork.alloc(sizeof(double) * (size_t)brows * (q0 + 2)) ||
        dout.alloc(sizeof(double) * (size_t)brows * 4))
        return 1;
    for (int64_t r0 = 0; r0 < m; r0 += brows) {
        const int rows = (int)std::min<int64_t>(brows, m - r0);
        JX_HIP(hipMemcpy(dg.p, g + (size_t)r0 * n, sizeof(float) * (size_t)rows * n, hipMemcpyHostToDevice));
        if (jxg_lm_scan_dense(dg.as<float>(), rows, n, n, dxr.as<double>(), q0, dixx.as<double>(), yy_r, dwork.as<double>(),
                              dout.as<double>(), nullptr))
            return 1;
        JX_HIP(hipMemcpy(out + (size_t)r0 * 4, dout.p, sizeof(double) * (size_t)rows * 4, hipMemcpyDeviceToHost));
    }
    return 0;
}

// ---- association TSV writer (host) ------------------------------------------------------------------------------------
// Native counterpart of the reference's row formatter + writer (src/io/assoc2tsv.rs:430-548, `AsyncTsvWriter`
// src/stats/common.rs:374): the numeric columns of every row are formatted here with Rust's float text (`{:.4}`,
// `{:.4e}` / `{:.6e}` with an unpadded exponent, `NaN`, `inf`), the per-row prefix `chrom\tpos\tsnp\tallele0\tallele1`
// comes from the caller as one byte blob with offsets.  stats: (rows, ncol) f64, ncol = 3 [beta, se, p], 4 [.., plrt] or
// 6 [beta, se, p, lambda, ml, plrt].  Writes `path` directly (the caller renames a temp file).
namespace {
inline char *put_str(char *o, const char *t) {
    while (*t) *o++ = *t++;
    return o;
}
inline char *put_fixed4(char *o, double v) {
    if (v != v) return put_str(o, "NaN");
    if (std::isinf(v)) return put_str(o, v > 0 ? "inf" : "-inf");
    // Rust's {:.4} prints every digit of a huge finite value (up to 309 before the point): the caller reserves
    // kRowReserve bytes per row, enough for four such fields
    const int len = snprintf(o, 336, "%.4f", v);
    return o + (len < 336 ? len : 335);
}
constexpr size_t kRowReserve = 2048;   // 4 fixed-point fields of <= 336 bytes + 5 exponent fields + tabs
inline char *put_exp(char *o, double v, int prec) {
    if (v != v) return put_str(o, "NaN");
    if (std::isinf(v)) return put_str(o, v > 0 ? "inf" : "-inf");
    char tmp[64];
    const int len = snprintf(tmp, sizeof(tmp), "%.*e", prec, v);
    int epos = len - 1;
    while (epos > 0 && tmp[epos] != 'e') --epos;
    memcpy(o, tmp, (size_t)epos + 1);          // mantissa and 'e'
    o += epos + 1;
    const char *x = tmp + epos + 1;
    if (*x == '-') *o++ = '-';
    if (*x == '-' || *x == '+') ++x;
    while (*x == '0' && x[1] != 0) ++x;        // Rust prints the exponent without padding: e-3, e0, e12
    return put_str(o, x);
}
}  // namespace

extern "C" int64_t jx_assoc_tsv_append(const char *path, const char *prefix_blob, const int64_t *prefix_off, int64_t rows,
                                       const float *af, const float *miss, const double *stats, int ncol, int append);

extern "C" int64_t jx_assoc_tsv_write(const char *path, const char *prefix_blob, const int64_t *prefix_off, int64_t rows,
                                      const float *af, const float *miss, const double *stats, int ncol) {
    return jx_assoc_tsv_append(path, prefix_blob, prefix_off, rows, af, miss, stats, ncol, 0);
}

// append bit 0: the rows go behind what `path` holds already, without a header (the block-wise writer of the streaming
// scan: header + first block with 0, every later block with 1); bit 1 (+2): `miss` holds COUNTS of missing samples and is
// printed as an integer (`AssocMissValue::Count`, src/io/assoc2tsv.rs:452-458: the LM routes) instead of a rate `{:.4}`
extern "C" int64_t jx_assoc_tsv_append(const char *path, const char *prefix_blob, const int64_t *prefix_off, int64_t rows,
                                       const float *af, const float *miss, const double *stats, int ncol, int append_flags) {
    const int append = append_flags & 1;
    const bool miss_count = (append_flags & 2) != 0;
    if (ncol != 3 && ncol != 4 && ncol != 6) {
        fail("unsupported GWAS result column count: " + std::to_string(ncol) + " (expected 3, 4, or 6)");
        return -1;
    }
    FILE *fh = fopen(path, append ? "ab" : "wb");
    if (!fh) {
        fail(std::string("cannot open ") + path + " for writing");
        return -1;
    }
    static const char *h3 = "chrom\tpos\tsnp\tallele0\tallele1\taf\tmiss\tbeta\tse\tchisq\tpwald\n";
    static const char *h4 = "chrom\tpos\tsnp\tallele0\tallele1\taf\tmiss\tbeta\tse\tchisq\tpwald\tplrt\n";
    static const char *h6 = "chrom\tpos\tsnp\tallele0\tallele1\taf\tmiss\tbeta\tse\tchisq\tpwald\tlambda\tml\tplrt\n";
    if (!append) fputs(ncol == 6 ? h6 : (ncol == 4 ? h4 : h3), fh);
    std::vector<char> buf((size_t)1 << 20);
    size_t used = 0;
    const double min_pos = 2.2250738585072014e-308;
    for (int64_t i = 0; i < rows; ++i) {
        const int64_t plen = prefix_off[i + 1] - prefix_off[i];
        if (used + (size_t)plen + kRowReserve > buf.size()) {
            if (fwrite(buf.data(), 1, used, fh) != used) {
                fclose(fh);
                fail(std::string("write to ") + path + " failed");
                return -1;
            }
            used = 0;
            if ((size_t)plen + kRowReserve > buf.size()) buf.resize((size_t)plen + 2 * kRowReserve);
        }
        char *o = buf.data() + used;
        memcpy(o, prefix_blob + prefix_off[i], (size_t)plen);
        o += plen;
        const double *r = stats + i * ncol;
        const double beta = r[0], se = r[1], p = r[2];
        double chisq, pv;
        if (std::isfinite(beta) && std::isfinite(se) && se > 0.0) {      // sanitize_assoc_pvalue, src/math/linalg.rs:111-121
            const double z = beta / se;
            chisq = z * z;
            pv = std::isfinite(p) ? std::min(std::max(p, min_pos), 1.0) : 1.0;
        } else {
            chisq = NAN;
            pv = 1.0;
        }
        *o++ = '\t'; o = put_fixed4(o, (double)af[i]);
        *o++ = '\t';
        if (miss_count) o += snprintf(o, 32, "%lld", (long long)llround((double)miss[i]));
        else o = put_fixed4(o, (double)miss[i]);
        *o++ = '\t'; o = put_fixed4(o, beta);
        *o++ = '\t'; o = put_fixed4(o, se);
        *o++ = '\t'; o = put_exp(o, chisq, 4);
        *o++ = '\t'; o = put_exp(o, pv, 4);
        if (ncol == 6) {
            *o++ = '\t'; o = put_exp(o, r[3], 6);
            *o++ = '\t'; o = put_exp(o, r[4], 6);
            *o++ = '\t'; o = put_exp(o, r[5], 4);
        } else if (ncol == 4) {
            *o++ = '\t'; o = put_exp(o, r[3], 4);
        }
        *o++ = '\n';
        used = (size_t)(o - buf.data());
    }
    const bool ok = fwrite(buf.data(), 1, used, fh) == used;
    if (fclose(fh) != 0 || !ok) {
        fail(std::string("write to ") + path + " failed");
        return -1;
    }
    return rows;
}
