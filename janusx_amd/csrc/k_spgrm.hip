// Sparse GRM (first half of `-splmm`): threshold + compaction of the f64 GRM accumulator into the
// lower-triangle CSC image of the reference's `.spgrm` file.
//
// Reference: `compute_spgrm_task_entries` (src/stats/spgrm.rs:3422-3554): per sample-tile pair the f32 GEMM blocks are
// merged in f64, `scaled = acc * inv_scale`, an entry (row >= col) is kept when it is on the diagonal or passes
// `spgrm_keep_value` (:1956-1965: |v| > thr with abs_threshold, everything when thr < 0, else v > thr); a non-finite
// value is an error; entries are ordered by (col, row) (`spgrm_entry_cmp` :1401) and laid out as col_ptr u64 /
// row_indices u32 / values f64 (`coo_lower_to_csc` :3637-3683, `write_sparse_grm_csc` :3745-3767).
//
// Here the accumulator is the one `jxg_grm_accumulate` leaves in HBM (row-major, lower triangle valid, leading
// dimension num_tiles(n) * 128), so the whole matrix is thresholded in place: HBM-bound, two streaming passes over the
// lower triangle (count, fill; 8 n^2 bytes together), 256 columns x 256 rows per workgroup with the wave reading 512
// contiguous bytes of one accumulator row per step; every thread owns one column of its band and walks the rows in
// order, so the compaction is order-preserving without a sort: entry position = col_ptr[col] + (entries of the bands
// above in that column) + running count.
#include <math.h>
#include <stdint.h>

#include "jx_common.h"

namespace jx {

constexpr int SG_T = 256;      // threads = columns per workgroup
constexpr int SG_BAND = 256;   // rows per band

__device__ __forceinline__ bool spgrm_keep(double v, double thr, int abs_thr) {
    if (abs_thr) return fabs(v) > thr;
    if (thr < 0.0) return true;
    return v > thr;
}

// FILL = false: d_cnt[band][col] = kept entries of (band, col); FILL = true: write them at their final position.
template <bool FILL>
__global__ __launch_bounds__(SG_T) void spgrm_band_kernel(const double *__restrict__ acc, int64_t ld, int n,
                                                          double inv_scale, double thr, int abs_thr,
                                                          int32_t *__restrict__ cnt,
                                                          const uint64_t *__restrict__ colptr,
                                                          uint32_t *__restrict__ rows, double *__restrict__ vals,
                                                          int *__restrict__ flag, int band0) {
    // band0: first band of this launch (a row panel of the accumulator; cnt is indexed by band - band0)
    const int c = blockIdx.x * SG_T + threadIdx.x;
    const int band = blockIdx.y + band0;
    if (c >= n) return;
    const int r_lo = band * SG_BAND;
    const int r_hi = (r_lo + SG_BAND < n) ? (r_lo + SG_BAND) : n;
    int kept = 0;
    uint64_t pos = 0;
    if (FILL) pos = colptr[c] + (uint64_t)cnt[(int64_t)(band - band0) * n + c];
    bool bad = false;
    const int r0 = (c > r_lo) ? c : r_lo;   // lower triangle: row >= col
    const double *p = acc + (int64_t)r0 * ld + c;
#pragma unroll 8
    for (int r = r0; r < r_hi; ++r, p += ld) {
        const double v = __builtin_nontemporal_load(p) * inv_scale;
        bad |= !isfinite(v);
        if (r == c || spgrm_keep(v, thr, abs_thr)) {
            if (FILL) {
                rows[pos] = (uint32_t)r;
                vals[pos] = v;
                ++pos;
            } else {
                ++kept;
            }
        }
    }
    if (!FILL) {
        cnt[(int64_t)(band - band0) * n + c] = kept;
        if (bad) atomicOr(flag, 1);
    }
}

// per column: counts of the bands -> exclusive prefix over the bands (in place), column total -> colptr[col + 1]
__global__ void spgrm_band_prefix_kernel(int32_t *__restrict__ cnt, int n, int band0, int nbands,
                                         uint64_t *__restrict__ colptr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    uint64_t run = 0;
    const int bfirst = (c / SG_BAND > band0) ? c / SG_BAND : band0;
    for (int b = bfirst; b < nbands; ++b) {        // bands above the diagonal hold no entry of this column
        const int64_t at = (int64_t)(b - band0) * n + c;
        const int32_t k = cnt[at];
        cnt[at] = (int32_t)run;
        run += (uint64_t)k;
    }
    colptr[c + 1] = run;
    if (c == 0) colptr[0] = 0;
}

// one workgroup: inclusive scan of colptr[1..n] (column totals) in place
__global__ __launch_bounds__(1024) void spgrm_colptr_scan_kernel(uint64_t *__restrict__ colptr, int n) {
    __shared__ uint64_t part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = t * per, hi = (lo + per < n) ? (lo + per) : n;
    uint64_t s = 0;
    for (int c = lo; c < hi; ++c) s += colptr[c + 1];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint64_t add = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint64_t run = (t > 0) ? part[t - 1] : 0;
    for (int c = lo; c < hi; ++c) {
        run += colptr[c + 1];
        colptr[c + 1] = run;
    }
}

// CSC (lower) -> dense symmetric f64, optionally the sub-matrix of a sample selection: map[orig] = new index or -1.
// One wave per column, lanes stride over the column's entries.
__global__ __launch_bounds__(256) void spgrm_densify_kernel(const uint64_t *__restrict__ colptr,
                                                            const uint32_t *__restrict__ rows,
                                                            const double *__restrict__ vals, int n,
                                                            const int32_t *__restrict__ map, int n_out,
                                                            double *__restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n) return;
    const int nc = map ? map[c] : c;
    if (nc < 0) return;
    const uint64_t lo = colptr[c], hi = colptr[c + 1];
    for (uint64_t q = lo + (threadIdx.x & 63); q < hi; q += 64) {
        const int r = (int)rows[q];
        const int nr = map ? map[r] : r;
        if (nr < 0) continue;
        const double v = vals[q];
        out[(int64_t)nr * n_out + nc] = v;
        out[(int64_t)nc * n_out + nr] = v;
    }
}

}  // namespace jx

using namespace jx;

static inline int spgrm_bands(int n) { return (n + SG_BAND - 1) / SG_BAND; }

extern "C" int64_t jxg_spgrm_work_bytes(int n) { return (int64_t)spgrm_bands(n) * (int64_t)n * 4 + 16; }

// Bands [band0, band1) of 256 sample rows (band1 < 0: all): d_acc is then the row panel holding exactly those bands (row 0
// = sample row band0 * 256, full leading dimension), d_work needs (band1 - band0) * n * 4 + 16 bytes and d_colptr / d_rows /
// d_vals describe the entries of these rows only (the caller merges the panels column by column).
extern "C" int jxg_spgrm_count_bands(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold,
                                     int band0, int band1, void *d_work, uint64_t *d_colptr, void *stream) {
    if (n <= 0) return fail("jxg_spgrm_count: n must be > 0");
    hipStream_t st = (hipStream_t)stream;
    const int nb_all = spgrm_bands(n);
    if (band1 < 0) {
        band0 = 0;
        band1 = nb_all;
    }
    if (band0 < 0 || band1 > nb_all || band0 >= band1) return fail("jxg_spgrm_count_bands: band range out of bounds");
    const int nb = band1 - band0;
    const int64_t ld = (int64_t)num_tiles(n) * JXG_TILE;
    const double *acc = d_acc - (int64_t)band0 * SG_BAND * ld;
    int32_t *cnt = (int32_t *)d_work;
    int *flag = (int *)((char *)d_work + (int64_t)nb * n * 4);
    JX_HIP(hipMemsetAsync(flag, 0, 16, st));
    JX_HIP(hipMemsetAsync(cnt, 0, (size_t)((int64_t)nb * n * 4), st));     // columns right of a band's rows stay empty
    hipLaunchKernelGGL(spgrm_band_kernel<false>, dim3((n + SG_T - 1) / SG_T, nb), dim3(SG_T), 0, st, acc, ld, n,
                       inv_scale, threshold, abs_threshold, cnt, (const uint64_t *)nullptr, (uint32_t *)nullptr,
                       (double *)nullptr, flag, band0);
    hipLaunchKernelGGL(spgrm_band_prefix_kernel, dim3((n + 255) / 256), dim3(256), 0, st, cnt, n, band0, band1, d_colptr);
    hipLaunchKernelGGL(spgrm_colptr_scan_kernel, dim3(1), dim3(1024), 0, st, d_colptr, n);
    JX_LAUNCH_CHECK();
    int hflag = 0;
    JX_HIP(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    JX_HIP(hipStreamSynchronize(st));
    if (hflag) return fail("Sparse GRM produced non-finite value");
    return 0;
}

extern "C" int jxg_spgrm_fill_bands(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold,
                                    int band0, int band1, const void *d_work, const uint64_t *d_colptr, uint32_t *d_rows,
                                    double *d_vals, void *stream) {
    if (n <= 0) return fail("jxg_spgrm_fill: n must be > 0");
    const int nb_all = spgrm_bands(n);
    if (band1 < 0) {
        band0 = 0;
        band1 = nb_all;
    }
    if (band0 < 0 || band1 > nb_all || band0 >= band1) return fail("jxg_spgrm_fill_bands: band range out of bounds");
    const int64_t ld = (int64_t)num_tiles(n) * JXG_TILE;
    const double *acc = d_acc - (int64_t)band0 * SG_BAND * ld;
    hipLaunchKernelGGL(spgrm_band_kernel<true>, dim3((n + SG_T - 1) / SG_T, band1 - band0), dim3(SG_T), 0,
                       (hipStream_t)stream, acc, ld, n, inv_scale, threshold, abs_threshold, (int32_t *)d_work, d_colptr, d_rows,
                       d_vals, (int *)nullptr, band0);
    JX_LAUNCH_CHECK();
    return 0;
}

extern "C" int jxg_spgrm_count(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold,
                               void *d_work, uint64_t *d_colptr, void *stream) {
    return jxg_spgrm_count_bands(d_acc, n, inv_scale, threshold, abs_threshold, 0, -1, d_work, d_colptr, stream);
}

extern "C" int jxg_spgrm_fill(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold,
                              const void *d_work, const uint64_t *d_colptr, uint32_t *d_rows, double *d_vals,
                              void *stream) {
    return jxg_spgrm_fill_bands(d_acc, n, inv_scale, threshold, abs_threshold, 0, -1, d_work, d_colptr, d_rows, d_vals, stream);
}

extern "C" int jxg_spgrm_densify(const uint64_t *d_colptr, const uint32_t *d_rows, const double *d_vals, int n,
                                 const int32_t *d_map, int n_out, double *d_out, void *stream) {
    if (n <= 0 || n_out <= 0) return fail("jxg_spgrm_densify: n and n_out must be > 0");
    hipStream_t st = (hipStream_t)stream;
    JX_HIP(hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)n_out * (size_t)n_out, st));
    hipLaunchKernelGGL(spgrm_densify_kernel, dim3((n + 3) / 4), dim3(256), 0, st, d_colptr, d_rows, d_vals, n, d_map,
                       n_out, d_out);
    JX_LAUNCH_CHECK();
    return 0;
}
