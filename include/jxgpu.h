/* jxgpu.h -- C ABI of libjxgpu.so: the MI355X (gfx950) implementation of JanusX's mixed-model hot path.
 *
 * Drop-in boundary: the reference exposes this path as PyO3 functions of the `janusx.janusx` extension
 * module (/root/reference/src/lib.rs:691-1005).  Each entry point below names the reference function it
 * replaces (file:line).  Plain pointers and sizes only; no Python/torch types.
 *
 * Two layers:
 *   jxg_*  device layer : every pointer is a DEVICE pointer (HBM), `stream` is a hipStream_t (may be NULL).
 *                         Used when inputs are already resident in HBM (bench.py, multi-GPU pipeline).
 *   jx_*   host layer   : every pointer is a HOST pointer; the call stages through HBM itself.  These have
 *                         the argument meaning of the reference's PyO3 functions (numpy arrays -> C arrays).
 *
 * Return value: 0 on success, non-zero on failure; `jx_last_error()` returns a thread-local message
 * (the reference raises RuntimeError(msg), e.g. src/stats/grm.rs:3079-3086).
 * Per-SNP numerical failures are not errors: rows are (NaN, NaN, 1.0) for the exact scan
 * (src/stats/lmm.rs:74-81) and (NaN, NaN, NaN) for the fixed-lambda scan (src/stats/fvlmm.rs:1753-1761).
 */
#ifndef JXGPU_H
#define JXGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JXG_TILE 128          /* sample tile of the internal packed layout ("P32": 128 samples = 32 bytes) */
#define JXG_MAX_COV 15        /* max fixed-effect columns (incl. intercept) handled by the scan kernels */

const char *jx_last_error(void);
int jx_version(void);

/* Progress hook of the host layer's row-block loops (jx_assoc_packed): `fn(done, total, user)` is called every `every` rows
 * (every <= 0: once per internal block of 8192 rows) and at the end; a nonzero return stops the call, which then fails with
 * "interrupted by the progress callback".  NULL clears the hook.  Replaces the `progress_callback` / `progress_every`
 * arguments of the reference's PyO3 entry points (src/stats/lmm.rs:3059-3083, 3214-3330; src/stats/grm.rs:3485-3495). */
typedef int (*jx_progress_fn)(int64_t done, int64_t total, void *user);
int jx_set_progress(jx_progress_fn fn, void *user, int64_t every);
int jxg_device_count(void);
int jxg_set_device(int dev);
/* device properties: out[0]=CU count, out[1]=clock kHz, out[2]=total HBM MiB, out[3]=LDS bytes/CU */
int jxg_device_info(int64_t *out4);

/* ------------------------------------------------------------------------------------------------
 * Device layer
 * ---------------------------------------------------------------------------------------------- */

/* number of 128-sample tiles and padded sample count for n samples */
int jxg_num_tiles(int n);

/* A1. Re-tile a PLINK SNP-major 2-bit payload (m, bps) into the internal P32 layout
 *   p32[(tile * m_out + j) * 32 + b]  =  samples tile*128 + 4b .. 4b+3 of output SNP j,
 * optionally gathering a sample subset (sample_idx, n_sel) and a row subset (row_idx, m_out).
 * Samples beyond n_sel are encoded 01 (missing).  Replaces the subset/gather plans of
 * src/math/bedmath.rs:1359-1441 and `SubsetDecodePlan`. */
int jxg_repack_p32(const uint8_t *d_packed, int64_t bps, int n_src, int64_t m_src,
                   const int32_t *d_sample_idx, int n_sel, const int64_t *d_row_idx, int64_t m_out,
                   uint8_t *d_p32, void *stream);

/* A1. per-SNP (missing, het, hom_alt) counts over the n_sel real samples of a P32 buffer
 * -> d_counts (m,3) int32.  src/io/gfreader.rs:1378-1395 `count_packed_row_counts`. */
int jxg_row_counts_p32(const uint8_t *d_p32, int64_t m, int n_sel, int32_t *d_counts, void *stream);
/* The same counts over a duplicate-free sample subset straight from a device-resident PLINK payload (m rows of bps bytes), without a
 * P32 image: d_mask (bps bytes) has both bits set at every selected sample.  jx_row_counts takes this route for device payloads. */
int jxg_row_counts_raw_masked(const uint8_t *d_packed, int64_t bps, int64_t m, const uint8_t *d_mask, int32_t *d_counts,
                              void *stream);

/* A3+A4. acc(lower tiles) += Z Z^T over the SNPs rows[k], k in [0, mk), where
 * z = lut[k][code] (4 f32 values per SNP indexed by the 2-bit code; lut[k][1] must be 0).
 * d_rows may be NULL (identity).  d_acc is (n_pad, n_pad) f64 row-major, n_pad = 128*tiles; only tiles
 * (ti >= tj) are written.  Numerics: value LUT split into fp16 hi+lo, three MFMA products with f32
 * accumulation over <= kchunk SNPs, f64 merge -- the counterpart of the reference's f32 SSYRK per block +
 * f64 merge (src/stats/grm.rs:1638-1667, 1700-1772).  SNPs whose values are b + {0, 1, 2} (either orientation) run as
 * an exact integer Gram on the int8 matrix pipes with f64 affine terms; with missing calls they additionally get the sparse
 * correction of k_grm_miss.hip (whole-triangle calls; f64, independent of scheduling).  precision: 0 = default,
 * 2 = rows with missing calls stay on the fp16 split kernel (what row-panel calls always do; the sparse-GRM builder
 * passes it so that its file does not depend on the memory plan), 1 = exact f32 MFMA (not built).  kchunk <= 0 selects
 * the default (8192). */
int jxg_grm_accumulate(const uint8_t *d_p32, int64_t m_total, int n_sel, const int32_t *d_rows,
                       const float *d_lut, int64_t mk, double *d_acc, int kchunk, int precision,
                       void *stream);
/* Tile rows [tile_row_begin, tile_row_end) of the lower triangle only (128-row tiles; end < 0: all); d_acc is then the
 * panel buffer: (end - begin) * 128 rows of the full leading dimension, row 0 = sample row begin * 128. */
int jxg_grm_accumulate_rows(const uint8_t *d_p32, int64_t m_total, int n_sel, const int32_t *d_rows, const float *d_lut,
                            int64_t mk, double *d_acc, int kchunk, int precision, int tile_row_begin, int tile_row_end,
                            void *stream);

/* A5. K = acc * inv_scale, mirrored lower -> upper, written as (n,n) f32 or f64 row-major.
 * src/stats/grm.rs:2771-2785 `grm_scale_and_symmetrize_raw_f64`. */
int jxg_grm_finalize(const double *d_acc, int n, double inv_scale, void *d_out, int out_is_f64,
                     void *stream);

/* A6 (next row 8f-3). Sparse GRM: threshold + order-preserving compaction of the accumulator into the lower-triangle
 * CSC image of the reference's `.spgrm` file (`compute_spgrm_task_entries`, src/stats/spgrm.rs:3422-3554; keep rule
 * `spgrm_keep_value` :1956-1965 — diagonal always, |v| > thr with abs_threshold, everything for thr < 0, else v > thr;
 * order (col, row) as `spgrm_entry_cmp` :1401; layout `coo_lower_to_csc` :3637-3683).
 * jxg_spgrm_count fills d_colptr (n + 1, u64; d_colptr[n] = nnz) and the per-band offsets in d_work
 * (jxg_spgrm_work_bytes(n) bytes), synchronises, and fails on a non-finite value as the reference does;
 * jxg_spgrm_fill writes d_rows (nnz, u32) and d_vals (nnz, f64 = acc * inv_scale). */
int64_t jxg_spgrm_work_bytes(int n);
int jxg_spgrm_count(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold, void *d_work,
                    uint64_t *d_colptr, void *stream);
int jxg_spgrm_fill(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold,
                   const void *d_work, const uint64_t *d_colptr, uint32_t *d_rows, double *d_vals, void *stream);
/* Row-panel forms for a sparse GRM whose n x n f64 accumulator does not fit in HBM (`jx_spgrm_packed_to_jxgrm` switches to
 * them above JXGPU_SPGRM_ACC_GB = 96 GB, or when JXGPU_SPGRM_PANEL_ROWS is set): bands [band0, band1) of 256 sample rows
 * (band1 < 0: all); d_acc is the panel holding exactly those rows (row 0 = sample row band0 * 256, leading dimension
 * num_tiles(n) * 128), d_work needs (band1 - band0) * n * 4 + 16 bytes, and d_colptr / d_rows / d_vals describe the entries of
 * these rows only (merged column by column by the caller). */
int jxg_spgrm_count_bands(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold, int band0,
                          int band1, void *d_work, uint64_t *d_colptr, void *stream);
int jxg_spgrm_fill_bands(const double *d_acc, int n, double inv_scale, double threshold, int abs_threshold, int band0,
                         int band1, const void *d_work, const uint64_t *d_colptr, uint32_t *d_rows, double *d_vals,
                         void *stream);

/* Dense symmetric image (n_out, n_out) f64 of a lower-triangle CSC sparse GRM in HBM, optionally restricted / reordered
 * by d_map (n int32: new index of every original sample, -1 = dropped; NULL = identity, n_out = n) — the operand of the
 * sparse REML null model (`subset_sparse_grm_csc`, src/math/cholesky.rs:618-690, then K + lambda I, src/stats/spreml.rs:
 * 384-512), which this library evaluates spectrally (jxg_eigh_f64 of the dense image) instead of by a sparse LLT. */
int jxg_spgrm_densify(const uint64_t *d_colptr, const uint32_t *d_rows, const double *d_vals, int n,
                      const int32_t *d_map, int n_out, double *d_out, void *stream);

/* B1. symmetric eigendecomposition, f64, ascending.  d_a (n,n) is overwritten with U^T row-major
 * (row j = eigenvector j); d_w receives the n eigenvalues.  Replaces LAPACK dsyevd/dsyevr behind
 * src/math/eigh.rs:1422-1528: own Householder tridiagonalisation (k_sytrd.hip), divide and conquer (k_stedc.hip) and
 * compact-WY back-transformation (k_ormtr.hip); rocSOLVER dsyevd only below n = 256.  `ridge` is added to the diagonal first
 * (python/janusx/assoc/workflow.py:5639-5641). */
int jxg_eigh_f64(double *d_a, int n, double ridge, double *d_w, void *stream);
/* 1 when every n-dependent launch grid of jxg_eigh_f64 is inside the HIP limits for an n-row problem (no GPU needed). */
int jxg_eigh_grid_check(int n);

/* Node-level distribution of B1 (no counterpart in the reference, whose LAPACK call is single-process): every rank
 * calls jxg_eigh_f64 on the same matrix; for n >= min_n (<= 0: keep the default 16384) each rank streams 1 / world of
 * the trailing-matrix tiles in every column of the tridiagonalisation and `allreduce(user)` has to sum the
 * `jxg_eigh_dist_staging_doubles(n)` doubles at d_staging over the ranks, ordered on the stream given to jxg_eigh_f64
 * (RCCL through torch.distributed in janusx_amd/pipeline.py).  The replicated part of every column is computed without
 * floating-point atomics in this mode, so the ranks' copies stay bit-identical.  world = 1 or allreduce = NULL: off. */
int64_t jxg_eigh_dist_staging_doubles(int n);
int jxg_eigh_set_dist(int rank, int world, int (*allreduce)(void *), void *user, double *d_staging,
                      int64_t staging_doubles, int min_n);
/* on != 0: the following jxg_eigh_f64 calls of this process decompose matrices of its own (no collective, no sharding, the
 * one-rank thresholds) although a distribution is registered; on = 0 restores it.  Used where a multi-rank job deals whole
 * small problems over the ranks instead of sharding one (the diagonal blocks of the sparse GRM, janusx._SpectralSparseReml). */
int jxg_eigh_set_local(int on);
/* Node-level distribution of the two-stage path of B1 (one rank: n >= 1500; several ranks: n >= 10000): with rank / world set by jxg_eigh_set_dist and a
 * gather callback registered here, every rank runs the (bit-reproducible) reduction stages and the divide and conquer on
 * the same matrix, back-transforms only the eigenvectors [n r / world, n (r + 1) / world) -- two thirds of the flops of
 * the decomposition shard this way -- and gather(user) must then deliver the other ranks' rows of the row-major result
 * into d_a (janusx_amd/pipeline.py: one broadcast per rank over RCCL).  NULL: off (the rank-sharded one-stage
 * tridiagonalisation of jxg_eigh_set_dist is used instead; JXGPU_DIST_EIGH_ONESTAGE=1 forces that too). */
int jxg_eigh_set_gather(int (*gather)(void *), void *user);
/* Guard of that mode: before the sharded back-transformations every rank hashes its replicated intermediate results
 * (eigenvalues of the tridiagonal matrix, the divide and conquer's permutation, the first row of its eigenvector matrix)
 * and agree(user, checksum) must return 1 when every rank reports the same value, 0 when they differ (every rank then
 * back-transforms ALL its eigenvectors itself: slower, self-consistent), < 0 on failure.  NULL: no check.
 * jxg_eigh_last_dist_agree: -1 not checked, 1 the replicas agreed, 0 they differed, for the last decomposition. */
int jxg_eigh_set_agree(int (*agree)(void *, uint64_t checksum), void *user);
int jxg_eigh_last_dist_agree(void);
/* Band reduction of the two-stage path with the trailing matrix SHARDED over the ranks (B1 on several GPUs; no counterpart in
 * the reference, which factors on one host: src/math/eigh.rs:1422-1528): ownership by block rows of `block` samples (<= 0:
 * 2048) dealt cyclically; per panel two collectives through allreduce(user, count) = the sum over the ranks of the first
 * `count` doubles of d_staging (jxg_eigh_band_staging_doubles(n) doubles) on the stream passed to jxg_eigh_f64: the partial
 * symmetric products Z = A22 V, and the gather of the next panel's block column.  From min_n rows on (<= 0: 8192), on the
 * multi-rank two-stage path only.  jxg_eigh_last_band_sharded: 1 when the last decomposition took it. */
int jxg_eigh_set_band_dist(int rank, int world, int (*allreduce)(void *, int64_t count), void *user, double *d_staging,
                           int64_t staging_doubles, int min_n, int block);
int64_t jxg_eigh_band_staging_doubles(int n);
int jxg_eigh_last_band_sharded(void);
/* 1 when the last multi-rank decomposition's divide and conquer formed only this rank's eigenvector columns in its top-level
 * merge (three quarters of its product flops; JXGPU_DIST_DC_WINDOW=0 switches that off). */
int jxg_eigh_last_dc_windowed(void);

/* Building blocks of B1's two-stage reduction, exported for tests and timing scripts (no counterpart in the reference,
 * which calls LAPACK dsyevd, src/math/eigh.rs:1320-1400).  All matrices column-major f64 in HBM.
 *   jxg_dgemm_f64:            C = alpha op(A) op(B) + beta C on the f64 matrix pipes (ksplit <= 0: automatic split over K)
 *   jxg_oz_dgemm_f64:         the same product on the int8 matrix pipes (csrc/k_ozgemm.hip: operands sliced into jxg_oz_planes()
 *                             signed base-254 digit planes, exact i32 digit products, f64 combination; ~254^-planes relative
 *                             to max|row of op(A)| max|column of op(B)|); h_ms (optional, 3 floats): slice A / slice B / product ms
 *   jxg_dsymm_lower_f64:      C = alpha A B + beta C, A (m,m) symmetric with only its lower triangle stored
 *   jxg_dsyr2k_lower_nt_f64:  lower tiles of C (m,m) = alpha A B' + beta C, A, B (m,k)
 *   jxg_sy2st_f64:            dense symmetric -> band (half bandwidth 64) -> tridiagonal (d_d, d_e); d_ab_out (optional,
 *                             128 x n) = the band after stage 1; h_flags[0] / [1] = stage-1 failure / stage-2 abort flag */
int jxg_dgemm_f64(int ta, int tb, int m, int n, int k, double alpha, const double *d_a, int64_t lda, const double *d_b,
                  int64_t ldb, double beta, double *d_c, int64_t ldc, int ksplit, void *stream);
int jxg_oz_dgemm_f64(int ta, int tb, int m, int n, int k, double alpha, const double *d_a, int64_t lda, const double *d_b,
                     int64_t ldb, double beta, double *d_c, int64_t ldc, float *h_ms, void *stream);
int jxg_oz_planes(void);
/* Digit planes per operand of the sliced products for the calls that follow (4 .. 6; anything else: back to the default of
 * JXGPU_OZ_PLANES / 6); returns the previous override.  pipeline.eigh_from_grm(f32_consumer=True) runs Q1 and the
 * divide-and-conquer merges with 5 planes (15 int8 products instead of 21) when the eigenvectors are only kept as the f32
 * U^T the reference's scan consumes (src/stats/reml.rs:109-198): orthogonality 4e-10 instead of 2e-12, far inside f32. */
int jxg_oz_set_planes(int planes);
int jxg_dsymm_lower_f64(int m, int n, double alpha, const double *d_a, int64_t lda, const double *d_b, int64_t ldb,
                        double beta, double *d_c, int64_t ldc, void *stream);
int jxg_dsyr2k_lower_nt_f64(int m, int k, double alpha, const double *d_a, int64_t lda, const double *d_b, int64_t ldb,
                            double beta, double *d_c, int64_t ldc, void *stream);
int jxg_sy2st_f64(double *d_a, int n, double *d_d, double *d_e, double *d_ab_out, int *h_flags, void *stream);

/* Multi-GPU reduce of the partial GRMs (no counterpart in the single-process reference): the lower-triangle
 * tiles (ti >= tj, 128 x 128) of the (npad, npad) f64 accumulator <-> a packed buffer of jxg_tri_tiles_doubles(npad)
 * doubles, so the all-reduce over xGMI moves n (n + 1) / 2 values instead of the n^2 square. */
int64_t jxg_tri_tiles_doubles(int npad);
int jxg_tri_tiles_pack_f64(double *d_acc, int npad, double *d_buf, int unpack, void *stream);

/* a <- (a + a^T)/2 (src/math/eigh.rs:179-207); dst = src^T; dst (k,k) f64 = src[idx, idx] of an (n,n)
 * f32/f64 matrix (trait-sample subset, python/janusx/assoc/workflow.py:5509-5560). */
int jxg_symmetrize_f64(double *d_a, int n, void *stream);
int jxg_transpose_f64(const double *d_src, double *d_dst, int n, void *stream);
int jxg_gather_sub_f64(const void *d_src, int src_is_f64, int n, const int32_t *d_idx, int k, double *d_dst,
                       void *stream);

/* f64 U^T -> f32 U^T (row-major), python/janusx/pyBLUP/assoc.py:1818 (`Dh`). */
int jxg_cast_f64_to_f32(const double *d_src, float *d_dst, int64_t count, void *stream);

/* C1. X~ = U^T X, y~ = U^T y; U^T f32, f64 accumulation.  d_xy (n, q) f64 row-major (covariates and y
 * side by side) -> d_out (n, q).  src/stats/reml.rs:109-198. */
int jxg_rotate_xy(const float *d_ut, int n, const double *d_xy, int q, double *d_out, void *stream);

/* C2-C4. null REML: Brent on log10(lambda) in [low, high].  d_out3 (DEVICE) = (lbd, ml, reml).
 * src/stats/reml.rs:572-616 `lmm_reml_null_f32`. */
int jxg_lmm_reml_null(const double *d_s, const double *d_xcov, const double *d_y, int n, int p, double low,
                      double high, int max_iter, double tol, double *d_out3, void *stream);

/* D2 (prep). split f32 U^T (n,n) into two fp16 planes (n_pad, n_pad), scaled by 2^scale_exp, zero padded. */
int jxg_ut_split(const float *d_ut, int n, uint16_t *d_hi, uint16_t *d_lo, int scale_exp, void *stream);

/* D1+D2. rotate a block of SNP rows straight from the packed payload:
 *   out[r, j] = sum_i lut[r][code(rows[r], i)] * u_t[j, i],   r in [0, nrows), j in [0, n)
 * lut (nrows,4) f32 holds the already centred design values (src/decode/decode.rs:192-271);
 * d_out (nrows, n) f32 row-major.  src/stats/lmm.rs:728-784 `rotate_snp_block_with_ut_blas`. */
int jxg_rotate_packed(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                      const float *d_lut, const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp,
                      float *d_out, void *stream);

/* D1+D2 split form for block loops: `jxg_lut_split` converts (mk,4) f32 design LUTs into the 16-byte fp16 hi/lo
 * records once (range-checked, synchronises once); `jxg_rotate_packed16` then only launches the MFMA kernel
 * (no allocation, no synchronisation) on rows[r] / lut16[r], r in [0, nrows). */
int jxg_lut_split(const float *d_lut, int64_t mk, void *d_lut16, void *stream);
int jxg_rotate_packed16(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                        const void *d_lut16, const uint16_t *d_uhi, const uint16_t *d_ulo, int scale_exp,
                        float *d_out, void *stream);

/* D1+D2, exact-row form.  A design row whose three genotype values are beta + {0,1,2} (the scan design g - row mean,
 * src/decode/decode.rs:192-271, either allele orientation) and that has no missing call among the n selected samples
 * is rotated as  (c U) + beta * usum:  `jxg_lut_split_rows` stores its integer LUT (fp16-exact, lo plane zero) and
 * beta in d_rowoff[k]; every other row keeps the hi/lo split and gets d_rowoff[k] = NaN.  `jxg_ut_rowsum` gives
 * usum[j] = sum_i u_t[j][i] (n_pad floats).  `jxg_rotate_packed16x` = `jxg_rotate_packed16` plus the affine term; a
 * 128-row tile whose rows all qualify skips the lo plane of the design (two MFMA products instead of three). */
int jxg_lut_split_rows(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const float *d_lut,
                       int64_t mk, void *d_lut16, float *d_rowoff, void *stream);
/* jxg_lut_split_rows with a tolerance for rows that hold 1 .. miss_max missing calls (jxg_rot_miss_max(n, mean number of
 * missing calls per row) is the default limit: 256 up to n / 300 missing calls per row on average, else 0 -- one decision per
 * scan, a block is not split between the kernels for a handful of rows): they keep the exact path (finite d_rowoff) and d_rowmiss[k] = lut[k][missing] - (value of the int8 rotation's clean
 * form at a missing call = offset + 2 [flipped]) is the weight of their
 * missing-call term, which jxg_rotate_missing_correct adds behind the rotation: out[r][j] += d_r * sum over the row's missing
 * samples i of U[i][j].  d_usamp (n, n) f32 = U with one row per sample (jxg_transpose_f32 of u_t).  Rows with more missing
 * calls stay general (NaN offset, fp16 kernel) as before.  Beyond n / 300 missing calls per row on average jxg_rot_miss_max
 * returns a value > 256 ("no limit"): every affine row keeps the exact path and its missing-call term is ONE MORE int8 product
 * with the indicator of the missing calls as the integer operand, jxg_rotate_missing_dense (d_sel: positions, inside the block,
 * of the rows with d_rowmiss != 0; d_q / d_umax: the three planes of jxg_ut_quant3) -- the cost of the fp16 kernel those rows
 * took before, but exact; JXGPU_ROT_MISS_DENSE=0 restores the fp16 kernel.  src/stats/lmm.rs:728-784 with the decode of
 * src/decode/decode.rs:192-271. */
int jxg_rot_miss_max(int n, double mean_missing_per_row);
int jxg_rotate_missing_dense(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const int32_t *d_sel, int nsel,
                             const float *d_rowmiss, const int8_t *d_q, const float *d_umax, float *d_out, int64_t ld_out,
                             void *stream);
int jxg_lut_split_rows_m(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, const float *d_lut,
                         int64_t mk, void *d_lut16, float *d_rowoff, float *d_rowmiss, int miss_max, void *stream);
int jxg_transpose_f32(const float *d_src, int n, float *d_dst, void *stream);
int jxg_rotate_missing_correct(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                               const float *d_rowmiss, const float *d_usamp, float *d_out, int64_t ld_out, void *stream);

int jxg_ut_rowsum(const float *d_ut, int n, float *d_usum, void *stream);
int jxg_rotate_packed16x(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                         const void *d_lut16, const float *d_rowoff, const float *d_usum, const uint16_t *d_uhi,
                         const uint16_t *d_ulo, int scale_exp, float *d_out, void *stream);
/* The same with a row pitch `ld_out` >= n (floats) of d_out: writes a block of eigenvector columns of a wider rotated-row
 * buffer (block-diagonal rotation of the sparse-GRM routes, one call per diagonal block). */
int jxg_rotate_packed16x_ld(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                            const void *d_lut16, const float *d_rowoff, const float *d_usum, const uint16_t *d_uhi,
                            const uint16_t *d_ulo, int scale_exp, float *d_out, int64_t ld_out, void *stream);
/* D2 on the int8 matrix pipes for exact design rows (no counterpart in the reference, which runs one f32 SGEMM,
 * src/stats/lmm.rs:728-784).  jxg_ut_quant3: U^T (n x n f32, row j = eigenvector j) -> three int8 planes d_q (3 x npad x npad
 * bytes, npad = 128 ceil(n / 128)) + one scale per eigenvector d_umax (npad f32): U_ij = umax_j (q1 / 127 + q2 / (127 254) +
 * q3 / (127 254^2)) to 2^-24 umax_j.  jxg_rotate_packed16x_q = jxg_rotate_packed16x with those planes and the block's rows
 * split into two lists of positions: d_sel_exact (rows that factor as beta + {0,1,2} without missing calls, i.e. finite
 * d_rowoff: three exact v_mfma_i32_32x32x32_i8 products combined in f64) and d_sel_rest (fp16 hi / lo kernel); together
 * they must cover 0 .. nrows-1.  A row's path, and its bits, do not depend on the block it is scanned in. */
int jxg_ut_quant3(const float *d_ut, int n, int8_t *d_q, float *d_umax, void *stream);
int jxg_rotate_packed16x_q(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows, const void *d_lut16,
                           const float *d_rowoff, const float *d_usum, const uint16_t *d_uhi, const uint16_t *d_ulo,
                           int scale_exp, const int8_t *d_q, const float *d_umax, const int32_t *d_sel_exact, int n_exact,
                           const int32_t *d_sel_rest, int n_rest, float *d_out, void *stream);

/* Rotation with the fused fixed-lambda reduction (src/stats/fvlmm.rs:1691-1805 consumes G~ only through three weighted sums):
 * G~ is not written; column tile t of this call (128 eigenvector columns, t < jxg_num_tiles(n)) writes its share of
 * (sum_j w_j g~_rj^2, sum_j g~_rj py_j, sum_j g~_rj wx_jk) to d_part[tile0 + t][r][0 .. p+1]; d_w, d_py (n) f32 and d_wx (n, p)
 * f32 are the state of jxg_fvlmm_prepare on this call's columns; d_part (tiles of all calls, nrows, lds_sums >= p + 2) f64.
 * Successive calls with tile0 advancing serve the diagonal blocks of the block route; p <= 8.  jxg_fvlmm_finish_dev adds the
 * ntiles tiles in index order (no atomics: bit-reproducible) and turns the sums into (beta, se, p[, plrt]) -- score_mode != 0:
 * the SparseLMM exact scan (src/stats/splmm.rs:2517-2538, df = n - p). */
int jxg_rotate_packed16x_fused(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                               const void *d_lut16, const float *d_rowoff, const float *d_usum, const uint16_t *d_uhi,
                               const uint16_t *d_ulo, int scale_exp, const float *d_w, const float *d_py,
                               const float *d_wx, int p, double *d_part, int lds_sums, int tile0, void *stream);
int jxg_fvlmm_finish_dev(const double *d_part, int ntiles, int lds, int nrows, int n, int p, const double *d_a_chol,
                         double ypy, int df, int with_plrt, double nullml, double log_det_v, int score_mode,
                         double *d_out, void *stream);

/* D2 (dense input). out[r, j] = sum_i g[r, i] * u_t[j, i] in exact f32 (f32 MFMA).
 * src/stats/lmm.rs:520-552 `rotate_snp_block_with_ut`. */
int jxg_rotate_dense_f32(const float *d_g, int nrows, int n, const float *d_ut, float *d_out, void *stream);

/* D3-D5. exact per-SNP scan of an already rotated block: Brent over -REML per SNP, GLS beta/SE, Wald p.
 * warm = 0: no warm start (reference core API, src/stats/lmm.rs:1577-1579); warm = 1: every SNP starts at
 * init_log10_lbd.  d_out (nrows, 3 or 4) f64 = [beta, se, p(, plrt)]; d_evals (nrows) int32 or NULL.
 * src/stats/lmm.rs:94-199. */
int jxg_lmm_scan(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                 const double *d_y, int p, double low, double high, double tol, int max_iter, int warm,
                 double init_log10_lbd, int with_plrt, double nullml, double *d_out, int32_t *d_evals,
                 void *stream);

/* D3 split form for block loops: the lambda-only REML sums (sum ln v, X'V^-1X, X'V^-1y, y'V^-1y) are tabulated
 * once per (model, [low, high]) as 32-term Chebyshev series per width-2 segment in a caller-provided device
 * workspace of jxg_lmm_tables_bytes() bytes (0 = this configuration needs jxg_lmm_scan_exact); jxg_lmm_scan_tab
 * then scans rotated blocks with one pass over the n samples per objective evaluation, without allocating or
 * synchronising.  jxg_lmm_scan = build + scan. */
int64_t jxg_lmm_tables_bytes(int n, int p, double low, double high);
int jxg_lmm_tables_build(const double *d_s, const double *d_xcov, const double *d_y, int n, int p, double low,
                         double high, void *d_work, void *stream);
int jxg_lmm_scan_tab(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov, int p, double low,
                     double high, const void *d_work, double tol, int max_iter, int warm, double init_log10_lbd,
                     int with_plrt, double nullml, double *d_out, int32_t *d_evals, void *stream);

/* Same contract as jxg_lmm_scan, evaluated with the reference's two-pass formulas (normal equations + explicit
 * residual quadratic form per evaluation, src/stats/reml.rs:286-344) instead of the tabulated lambda-only sums;
 * selected automatically when high - low > 16 or when JXGPU_SCAN_EXACT is set.  Used to validate the fast path. */
int jxg_lmm_scan_exact(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                       const double *d_y, int p, double low, double high, double tol, int max_iter, int warm,
                       double init_log10_lbd, int with_plrt, double nullml, double *d_out, int32_t *d_evals,
                       void *stream);

/* D3 along the reference's warm-start chains (`carry_warm_start`, src/stats/lmm.rs:134-161: each SNP's Brent starts from the
 * optimum of the valid SNP before it; the default of `lmm_reml_assoc_packed_f32`, src/stats/lmm.rs:3244-3245, and of the BED
 * route, :2627).  d_chain_off: nchains + 1 ascending int32 row offsets into this block (device); d_carry: nchains doubles
 * (device): the log10 lambda a chain starts from on entry (NaN: none -- the interval midpoint), the optimum of its last valid
 * SNP on return, so that a chain cut by the caller's blocking continues in the next call.  One wave per chain, the chains in
 * parallel.  _tab: tables from jxg_lmm_tables_build; jxg_lmm_scan_chain = build + scan (and jxg_lmm_scan_exact_chain, the
 * reference-formulation kernel, for configurations outside the tables). */
int jxg_lmm_scan_chain_tab(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov, int p, double low,
                           double high, const void *d_work, double tol, int max_iter, const int32_t *d_chain_off, int nchains,
                           double *d_carry, int with_plrt, double nullml, double *d_out, int32_t *d_evals, void *stream);
int jxg_lmm_scan_chain(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov, const double *d_y, int p,
                       double low, double high, double tol, int max_iter, const int32_t *d_chain_off, int nchains,
                       double *d_carry, int with_plrt, double nullml, double *d_out, int32_t *d_evals, void *stream);
int jxg_lmm_scan_exact_chain(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov, const double *d_y,
                             int p, double low, double high, double tol, int max_iter, const int32_t *d_chain_off, int nchains,
                             double *d_carry, int with_plrt, double nullml, double *d_out, int32_t *d_evals, void *stream);
/* The series form of D3 in two steps (csrc/k_scan_fast.hip): jxg_lmm_series_coef_tab turns a block of rotated rows into every
 * SNP's Chebyshev series of its SNP-specific sums (one pass over the rows on the f64 matrix pipes; d_scoef: nrows x
 * jxg_lmm_series_doubles(p, low, high) doubles, d_ssq: nrows), jxg_lmm_series_brent_tab runs the Brent searches on stored series
 * -- one per row (d_chain_off NULL; warm / init as jxg_lmm_scan_tab) or along chains (as jxg_lmm_scan_chain_tab).  Chain scans of
 * a whole payload keep the series of many blocks and walk all chains in ONE launch.  jxg_lmm_series_doubles = 0: this
 * (p, low, high) has no series form (more than two width-2 segments, p > 14). */
int64_t jxg_lmm_series_doubles(int p, double low, double high);
int jxg_lmm_series_coef_tab(const float *d_grot, int nrows, int n, const double *d_xcov, int p, double low, double high,
                            const void *d_work, double *d_scoef, double *d_ssq, void *stream);
int jxg_lmm_series_brent_tab(int nrows, int n, const double *d_s, const double *d_xcov, int p, double low, double high,
                             const void *d_work, double tol, int max_iter, int warm, double init_log10_lbd, const double *d_scoef,
                             const double *d_ssq, const int32_t *d_chain_off, int nchains, double *d_carry, int with_plrt,
                             double nullml, double *d_out, int32_t *d_evals, void *stream);

/* E1. fixed-lambda cache (device vectors): w f32(n), py f32(n), wx f32(n,p); scalars to HOST out:
 * a_chol (p*p), ypy, log_det_v, df.  src/stats/fvlmm.rs:1484-1563. */
int jxg_fvlmm_prepare(const double *d_s, const double *d_xcov, const double *d_y, int n, int p, double lbd,
                      float *d_w, float *d_py, float *d_wx, double *h_a_chol, double *h_scalars3);

/* E2. fixed-lambda scan of a rotated block -> d_out (nrows, 3 or 4) f64 = [beta, se, p(, plrt)]; with_plrt adds
 * the ML likelihood-ratio column against nullml (log_det_v from jxg_fvlmm_prepare).  src/stats/fvlmm.rs:1691-1805. */
int jxg_fvlmm_scan(const float *d_grot, int nrows, int n, int p, const float *d_w, const float *d_py,
                   const float *d_wx, const double *h_a_chol, double ypy, int df, int with_plrt, double nullml,
                   double log_det_v, double *d_out, void *stream);

/* LMM2 scan of a rotated block ("next" row 8f-2): per SNP a REML Brent, `final_beta_se`, then a second Brent on
 * -ml_loglike seeded with the REML optimum and the LRT against nullml.  d_out6 (nrows, 6) f64 =
 * [beta, se, pwald, lambda_reml, ml_alt, plrt]; invalid rows (NaN, NaN, 1, NaN, NaN, 1).  warm as jxg_lmm_scan.
 * `run_rotated_lmm2_assoc_block_f32`, src/stats/lmm.rs:202-330. */
int jxg_lmm2_scan(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                  const double *d_y, int p, double low, double high, double tol, int max_iter, int warm,
                  double init_log10_lbd, double nullml, double *d_out6, void *stream);

/* The same scan through the reference-formulation kernel (one wave per SNP, every evaluation a pass over the n samples);
 * jxg_lmm2_scan takes the tabulated tiled kernel whenever the lambda range and p fit its tables, this one otherwise. */
int jxg_lmm2_scan_exact(const float *d_grot, int nrows, int n, const double *d_s, const double *d_xcov,
                        const double *d_y, int p, double low, double high, double tol, int max_iter, int warm,
                        double init_log10_lbd, double nullml, double *d_out6, void *stream);

/* Null ML for the LMM2 scan: Brent on -ml_loglike without a SNP column -> d_out2 = (log10 lambda, ml0).
 * src/stats/lmm.rs:2902-2921. */
int jxg_lmm2_null_ml(const double *d_s, const double *d_xcov, const double *d_y, int n, int p, double low,
                     double high, int max_iter, double tol, int has_init, double init, double *d_out2,
                     void *stream);

/* C2. `ml_loglike` and `reml_loglike` of the null model at log10 lambda -> d_out2 = (ml, reml); -1e8 on failure.
 * src/stats/reml.rs:255-470. */
int jxg_lmm_loglike_null(const double *d_s, const double *d_xcov, const double *d_y, int n, int p,
                         double log10_lbd, double *d_out2, void *stream);

/* Duration (ms, HIP events on the launch stream) of the MFMA kernel(s) issued by the most recent
 * jxg_grm_accumulate (which = 0) or jxg_rotate_packed (which = 1) call of this process; which = 2: mean duration
 * of the symv launches sampled by the most recent jxg_eigh_f64 (one per 64-column panel), which = 3: the mean
 * algorithmic megabytes (lower triangle of the trailing matrix, f64) of those launches; which = 4 / 5: duration (ms) and algorithmic
 * GFLOP of the Q2 back-transformation of the most recent two-stage decomposition, 6 - 9: band reduction, bulge chasing, divide and
 * conquer, Q1 (ms), 10: 1 when the two-stage path ran, 16: apply launches of that Q2, 17: its form (0 three waves per unit, 1 one
 * wave per unit, 2 two sweep groups per pass, 3 balanced five / four units per CU); 18 - 21: the last jx_rrblup_pcg_packed solve --
 * wall ms of its iteration loop, its iterations, the summed HIP-event ms of its two streaming operator kernels (Z'p and Z (Z'p):
 * 2 x n_train x m / 4 payload bytes per application), wall ms of its set-up (images + pre-pass); 22 / 23: summed operator kernel
 * ms and operator applications of the last jx_he_traces_packed call.  Counterpart of the
 * reference's JX_GRM_*_STAGE_TIMING / JX_LMM_*_STAGE_TIMING stage timers (src/stats/grm.rs:3521-3568). */
float jxg_last_kernel_ms(int which);

/* G1 ("next" row). GBLUP on a training GRM: eigh + intercept-only REML (Brent on log10 lambda,
 * v floored at 1e-12) + alpha = U (V^-1 r).  d_k (n,n) f64 is overwritten with U^T; d_yc = y - mean(y).
 * h_out = (lambda, beta_rot, r'V^-1r, ml, reml, mean eigenvalue).  src/stats/gblup.rs:1105-1240. */
int jxg_gblup_fit(double *d_k, int n, double ridge, const double *d_yc, double low, double high, double tol,
                  int max_iter, double *d_alpha, double *h_out, void *stream);
/* d_out[i] = beta0 + sum_j K[rows[i], cols[j]] alpha[j]; K (n_full, n_full) f32/f64 row-major on the device
 * (`square_matrix_subset_cross_dot_f64`, src/stats/gblup.rs:1446-1460). */
int jxg_cross_dot(const void *d_k, int k_is_f64, int64_t n_full, const int32_t *d_rows, int nrows,
                  const int32_t *d_cols, int ncols, const double *d_alpha, double beta0, double *d_out,
                  void *stream);

/* Decoded rows of a P32 image: d_out[r][i] = d_lut[r][code(rows[r], i)] for the n selected samples, d_out (nrows, ld >= n)
 * f32, d_rows = SNP records or NULL for 0 .. nrows-1.  `bed_packed_decode_rows_f32`, src/stats/packed.rs:577-672 (the decode
 * the reference's Python layer asks for when it wants genotype rows as numbers). */
int jxg_decode_rows_p32(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                        const float *d_lut, float *d_out, int64_t ld, void *stream);

/* Matrix-free products with the 2-bit genotype matrix of a P32 image (n selected samples, rows = SNP records or NULL
 * for all m_total); the decoded value of (SNP r, sample i) is d_lut[r][code], d_lut (nrows, 4) f32:
 *   jxg_packed_tdot: d_out[r] = sum_i lut[r][code(r,i)] alpha[i]   (nrows)  `compute_malpha_from_meta_stream`,
 *                    src/stats/gblup.rs:859-925 (and the Z'v half of src/math/pcg.rs:578-640);
 *   jxg_packed_dot : d_out[i] = sum_r lut[r][code(r,i)] beta[r]    (n)      `predict_from_effect_stream`,
 *                    src/stats/gblup.rs:1037-1103 (the Z v half). f64 accumulation; outputs are overwritten. */
int jxg_packed_tdot(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                    const float *d_lut, const double *d_alpha, double *d_out, void *stream);
int jxg_packed_dot(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                   const float *d_lut, const double *d_beta, double *d_out, void *stream);
/* jxg_packed_tdot with the vector rounded to f32 (the Z u half of the PCG operator of `rrblup_pcg_bed`, whose vectors are
 * f32: src/stats/rrblup.rs:1220-1372).  From 1024 samples and SNPs on: the bit planes of the codes against a four-digit int8
 * image of the vector on v_mfma_i32_16x16x64_i8, exact sums (csrc/k_pcg_i8.hip); below that, or with JXGPU_PCG_I8=0: the
 * bit-plane table form (three LDS lookups per four genotypes, f32 partial sums inside a 128-sample tile). */
int jxg_packed_tdot_f32(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                        const float *d_lut, const double *d_u, double *d_out, void *stream);
/* The Z'p half of the same operator from a sample-major image of the payload: `jxg_p32_transpose` builds
 * t32[snp_tile][sample][32 B] (128 consecutive SNPs of the row list per record; `jxg_t32_bytes` bytes) once per solve,
 * `jxg_packed_dot_t32` then evaluates d_out[i] = sum_r f32(lut[r][code] * f32(beta[r])) -- the same int8 form (the three
 * per-SNP weight vectors as digit planes) or, below 1024 / with JXGPU_PCG_I8=0, the same bit-plane tables; d_work needs
 * 16 * nrows + 16 bytes. */
int64_t jxg_t32_bytes(int n, int nrows);
int jxg_p32_transpose(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows, uint8_t *d_t32,
                      void *stream);
int jxg_packed_dot_t32(const uint8_t *d_t32, int n, int nrows, const float *d_lut, const double *d_beta, void *d_work,
                       double *d_out, void *stream);

/* SparseLMM exact scan on rotated rows (`exact_scan_blocks_core`, src/stats/splmm.rs:2567-2880, with V = K + lambda I
 * handled spectrally): per row g~ = U'g the sums g~'Wg~, g~'(W X~), g~.(Py)~ of E2, then the score-form Wald test of
 * `splmm_wald_from_score_denom` (:2517-2538) with the NULL model's sigma2 = ypy / df (df = n - p): out (nrows, 3) =
 * (beta, se, p), rows with g'Pg <= 1e-30 are (NaN, NaN, 1).  d_w, d_py, d_wx, d_a_chol, ypy as produced by
 * jxg_fvlmm_prepare at the null lambda. */
int jxg_splmm_exact_scan_dev(const float *d_grot, int nrows, int n, int p, const float *d_w, const float *d_py,
                             const float *d_wx, const double *d_a_chol, double ypy, int df, double *d_out,
                             void *stream);

/* E2 with the Cholesky factor already on the device: launch only (no allocation, no synchronisation). */
int jxg_fvlmm_scan_dev(const float *d_grot, int nrows, int n, int p, const float *d_w, const float *d_py,
                       const float *d_wx, const double *d_a_chol, double ypy, int df, int with_plrt, double nullml,
                       double log_det_v, double *d_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Host layer (reference PyO3 signatures with C arrays)
 * ---------------------------------------------------------------------------------------------- */

/* per-SNP (missing, het, hom_alt) counts over the selected samples -> out_counts (m,3) int32
 * (src/io/gfreader.rs:1378-1395 `count_packed_row_counts`, the QC input of src/stats/lmm.rs:1258-1320). */
int jx_row_counts(const uint8_t *packed, int64_t m, int n_samples, const int64_t *sample_indices, int n_sel,
                  int32_t *out_counts);

/* `grm_packed_f32` / `grm_packed_f64_with_stats` (src/stats/grm.rs:3053-3066, 5611-5651).
 * packed (m, ceil(n_samples/4)) u8; row_flip (m) u8 0/1; row_maf (m) f32; sample_indices (n_sel) i64 or
 * NULL.  out_k (n,n) f32 or f64; out_row_sum (m) f64 or NULL; out_varsum (1) f64 or NULL. */
int jx_grm_packed(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                  const float *row_maf, const int64_t *sample_indices, int n_sel, int method,
                  void *out_k, int out_is_f64, double *out_row_sum, double *out_varsum);

/* `grm_stream_bed_f32` (src/stats/grm.rs:4676-4703) on an in-memory payload: QC pass + GRM.
 * out_k (n,n) f32; out_eff_m (1); out_keep (m) u8 or NULL. */
int jx_grm_stream_payload_f32(const uint8_t *packed, int64_t m, int n_samples, int method,
                              float maf_threshold, float max_missing_rate, float het_threshold,
                              float *out_k, int64_t *out_eff_m, uint8_t *out_keep);

/* `rust_eigh_from_array_f64` (src/math/eigh.rs:1621-1703): a (n,n) f64 row-major (symmetrised
 * (A+A^T)/2 like eigh.rs:179) -> evals (n) ascending, evecs (n,n) row-major, columns = eigenvectors
 * (NULL = values only still computes vectors internally). */
int jx_eigh_f64(const double *a, int n, double diag_shift, double *evals, double *evecs);

/* `lmm_rotate_x_y_with_ut_f64` (src/stats/reml.rs:107-198). */
int jx_lmm_rotate_x_y_with_ut_f64(const float *u_t, int n, const double *x, int q, const double *y,
                                  double *out_x, double *out_y);

/* `lmm_reml_null_f32` (src/stats/reml.rs:570-616) -> out3 = (lbd, ml, reml). */
int jx_lmm_reml_null(const double *s, const double *xcov, const double *y_rot, int n, int p, double low,
                     double high, int max_iter, double tol, double *out3);

/* `lmm_reml_chunk_f32` (src/stats/lmm.rs:333-335; rotated input) and `lmm_reml_chunk_from_snp_f32`
 * (src/stats/lmm.rs:1479-1630; u_t != NULL -> rotate first).  has_nullml -> 4 output columns. */
int jx_lmm_reml_chunk(const double *s, const double *xcov, const double *y_rot, int n, int p, double low,
                      double high, const float *snp_chunk, int64_t m_chunk, const float *u_t, int max_iter,
                      double tol, int has_nullml, double nullml, double *out);

/* `ml_loglike_null_f32` (src/stats/reml.rs:618-646) -> *ml. */
int jx_ml_loglike_null(const double *s, const double *xcov, const double *y_rot, int n, int p, double log10_lbd,
                       double *ml);

/* `lmm_reml_lmm2_chunk_from_snp_f32` (src/stats/lmm.rs:1632-1760; u_t == NULL -> snp_chunk is already rotated)
 * -> out (m_chunk, 6) = [beta, se, pwald, lambda_reml, ml_alt, plrt]. */
int jx_lmm2_chunk(const double *s, const double *xcov, const double *y_rot, int n, int p, double low, double high,
                  const float *snp_chunk, int64_t m_chunk, const float *u_t, double nullml, int max_iter,
                  double tol, double *out);

/* Null ML of `lmm_reml_lmm2_assoc_bed_to_tsv_f32` when the caller passes no nullml (src/stats/lmm.rs:2902-2921)
 * -> out2 = (log10 lambda, ml0). */
int jx_lmm2_null_ml(const double *s, const double *xcov, const double *y_rot, int n, int p, double low,
                    double high, int max_iter, double tol, int has_init, double init, double *out2);

/* `fvlmm_assoc_chunk_f32` / `fvlmm_assoc_chunk_from_snp_f32` (src/stats/fvlmm.rs:1941-1994, 2114-2262).
 * has_nullml -> 4 output columns. */
int jx_fvlmm_assoc_chunk(const double *s, const double *xcov, const double *y_rot, int n, int p,
                         double log10_lbd, const float *snp_chunk, int64_t m_chunk, const float *u_t,
                         int has_nullml, double nullml, double *out);

/* `lmm_reml_assoc_packed_f32` (src/stats/lmm.rs:3040-3362) and its fixed-lambda sibling
 * (`fvlmm_assoc_packed` core of src/stats/fvlmm.rs:4958-5190 with a caller-rotated null model).
 * model: 0 = exact per-SNP REML (lmm), 1 = fixed lambda (fvlmm; `low` carries log10 lambda), 2 = LMM2
 * (`lmm_reml_lmm2_assoc_bed_to_tsv_f32` core, src/stats/lmm.rs:2779-3037; needs nullml; out (m, 6)).
 * warm: 0 none (parity contract), 1 seed with init_log10_lbd.  out (m, 3), or (m, 4) with has_nullml (plrt column,
 * src/stats/lmm.rs:202-330). */
int jx_assoc_packed(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                    const float *row_maf, const double *s, const double *xcov, const double *y_rot,
                    const float *u_t, int p, const int64_t *sample_indices, int n_sel, int model, double low,
                    double high, int max_iter, double tol, int warm, double init_log10_lbd, int has_nullml,
                    double nullml, double *out);
/* The same with the `model` / `genetic_model` argument of the reference's entry points (`PackedGeneticModel`,
 * src/decode/decode.rs:100-147: 0 add, 1 dom (g > 0), 2 rec (g = 2), 3 het (g = 1) applied to the decode table
 * [0 | 2, 2 maf, 1, 2 | 0] INCLUDING its imputed entry (:163-178), then the row is centred by its own mean (:181-189)).
 * jx_assoc_packed is genetic_model = 0.  Non-additive rows are not affine in the allele count: they take the general
 * (fp16 hi / lo) rotation. */
int jx_assoc_packed_gm(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                    const float *row_maf, const double *s, const double *xcov, const double *y_rot,
                    const float *u_t, int p, const int64_t *sample_indices, int n_sel, int model, double low,
                    double high, int max_iter, double tol, int warm, double init_log10_lbd, int has_nullml,
                    double nullml, double *out, int genetic_model);
/* lmm_reml_assoc_packed_f32 with the reference's default warm-start chains (src/stats/lmm.rs:3244-3245 `true, true`; the BED
 * route's `use_warm_start`, :2627): the exact scan (model 0) where chain c = rows [chain_off[c], chain_off[c + 1]) of the payload
 * in order (host int64 offsets, chain_off[0] = 0, ascending, chain_off[n_chains] = m; the reference's chains are its blocks of
 * `rotate_block_rows` rows, cut further by rayon's work splitting).  The first valid SNP of a chain starts from init_log10_lbd
 * (warm != 0) or the interval midpoint, every later one from the optimum of the valid SNP before it (:134-161). */
int jx_assoc_packed_chain(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip, const float *row_maf,
                          const double *s, const double *xcov, const double *y_rot, const float *u_t, int p,
                          const int64_t *sample_indices, int n_sel, double low, double high, int max_iter, double tol, int warm,
                          double init_log10_lbd, int has_nullml, double nullml, double *out, int genetic_model,
                          const int64_t *chain_off, int64_t n_chains);

/* `lm_block_assoc_packed` (src/stats/glm.rs:3550-3860): the plain LM scan `jx gwas -lmm / -fvlmm` switches to when the null
 * likelihood-ratio test (jx_gwas_lmm_lm_null_lrt_decision) finds no polygenic variance
 * (python/janusx/assoc/workflow_model_stream.py:930-963).  x (n, q0) row-major INCLUDING the intercept column, ixx (q0, q0) =
 * (X'X)^-1 (`_lm_precompute_ixx_qr`, python/janusx/pyBLUP/assoc.py:453-480), decode = mean-imputed additive
 * ([0, clamp(2 maf, 0, 2), 1, 2] or flipped, src/math/bedmath.rs:984-989).  out (m, 4) = beta, se, pwald (two-sided Student t,
 * df = n - q0 - 1), plrt. */
int jx_lm_assoc_packed(const double *y, const double *x, const double *ixx, int q0, const uint8_t *packed, int64_t m,
                       int n_samples, const uint8_t *row_flip, const float *row_maf, const int64_t *sample_indices,
                       int n_sel, double *out);
/* Host half of the LM scan (glm.rs:3635-3672): r_y = y - X (ixx X'y), *yy_r = r_y'r_y, xr_out (n, q0 + 1) = [X | r_y]
 * rounded through f32 as the reference does before its sgemm. */
int jx_lm_residualize(const double *y, const double *x, const double *ixx, int n, int q0, double *xr_out,
                      double *yy_r_out);
/* Device half on a resident P32 image: d_lut (nrows, 4) f32, d_xr (n, q0 + 1) f64, d_ixx (q0, q0) f64, d_work
 * (nrows * (q0 + 2)) f64 scratch, d_out (nrows, 4) f64. */
int jxg_lm_scan_p32(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows, const float *d_lut,
                    const double *d_xr, int q0, const double *d_ixx, double yy_r, double *d_work, double *d_out,
                    void *stream);
/* `lm_block_assoc_f32` (src/stats/glm.rs:4313-4497): the same LM formulas on an already decoded SNP-major f32 block
 * g (m, n) on the host; out (m, 4).  Row rules of that entry point (s must exceed 1e-12; a non-finite variance or standard
 * error voids the row, glm.rs:4457-4479). */
int jx_lm_assoc_dense(const double *y, const double *x, const double *ixx, int q0, const float *g, int64_t m, int n,
                      double *out);
/* Device half for dense rows: d_g (nrows, ld >= n) f32 on the device, the other arguments as jxg_lm_scan_p32. */
int jxg_lm_scan_dense(const float *d_g, int nrows, int n, int64_t ld, const double *d_xr, int q0, const double *d_ixx,
                      double yy_r, double *d_work, double *d_out, void *stream);

/* SparseLMM approximate (GRAMMAR-gamma) scan = `grammar_scan_blocks_core`, additive model (src/stats/splmm.rs:2935-3316; the
 * scan of `scan_with_py_and_rhat`, :3318-3363, behind `splmm_assoc_pcg_bed[_to_tsv]` in scan_mode "approx", :4641, 4814):
 * d_lut (nrows, 4) f32 = mean-imputed additive value by 2-bit code (not centred), d_xr (n, p + 1) f64 = [X | score vector]
 * already rounded through f32 (`pack_score_design_rhs_f32`), d_ixx (p, p) f64 = (X'X)^-1, d_work (nrows * (p + 2)) f64
 * scratch, d_out (nrows, 3) f64 = beta, se, p; score = score_scale g.score_vec, denominator = denom_scale g'M_X g. */
int jxg_splmm_grammar_scan_p32(const uint8_t *d_p32, int64_t m_total, int n, const int32_t *d_rows, int nrows,
                               const float *d_lut, const double *d_xr, int p, const double *d_ixx, double score_scale,
                               double denom_scale, double sigma2, double *d_work, double *d_out, void *stream);

/* Sums behind `estimate_gamma_from_markers` (src/stats/splmm_approx.rs:921-1068) over rotated marker rows g~ = U'g:
 * d_out (nrows, 3 + 2 p) = [g~'g~, g~'Wg~, g~'a~, X~'g~ (p), X~'Wg~ (p)]; d_w, d_a (n) f64, d_x (n, p) f64 row-major. */
int jxg_splmm_gamma_sums(const float *d_grot, int nrows, int n, int64_t ld, int p, const double *d_w, const double *d_a,
                         const double *d_x, double *d_out, void *stream);

/* `gblup_reml_npy_grm` (src/stats/gblup.rs:1242-1516) on an in-memory GRM: fit on K[train,train] + g_eps I, predict
 * K[*,train] alpha + beta0.  out_scalars = (pve, lambda, ml, reml, sigma_g2, sigma_e2, beta0). */
int jx_gblup_reml_grm(const void *k_full, int k_is_f64, int64_t n_full, const int64_t *train_idx, int n_train,
                      const double *y_train, const int64_t *test_idx, int n_test, double g_eps, double low,
                      double high, int max_iter, double tol, int estimate_only, double *out_pred_train,
                      double *out_pred_test, double *out_scalars);

/* `rrblup_pcg_bed` on a resident packed payload (src/stats/rrblup.rs:3494-4307): marker effects by
 * Jacobi-preconditioned conjugate gradients, (Z_c Z_c' + lambda I) beta = Z y_c, Z the standardised genotypes of the
 * training samples streamed from the 2-bit payload (never materialised).  value_lut (eff_m,4) f32 = design values by
 * 2-bit code [00, 01 (missing, must be 0), 10, 11] (src/math/bedmath.rs:1199-1214); row_indices (eff_m) selects the
 * kept SNP rows (NULL = all m_total).  out_beta (eff_m) f32; out_pred_train (n_train) and out_pred_test (n_test) f64
 * (either may be NULL); out_scalars = (converged, iterations, relative residual, sum of centred row sums of squares,
 * intercept).  The iteration follows `pcg_solve_into` for f32 (src/math/pcg.rs:870-949).  `packed` is a host pointer
 * (uploaded once) or a DEVICE pointer (a payload already resident in HBM is used in place -- BASELINE configs[4]: 50 GB);
 * jx_he_traces_packed likewise. */
int jx_rrblup_pcg_packed(const uint8_t *packed, int64_t m_total, int n_samples, const int64_t *row_indices,
                         int64_t eff_m, const float *value_lut, const int64_t *train_idx, int n_train,
                         const double *y_train, const int64_t *test_idx, int n_test, double lambda_value, double tol,
                         int max_iter, float *out_beta, double *out_pred_train, double *out_pred_test,
                         double *out_scalars);
/* Marker-sharded form of that solve over `world` ranks (last row: SNP-range shards, one all-reduce of an
 * n_train-vector per iteration; the reference has no distributed layer, src/math/pcg.rs:870-949 is the iteration every rank
 * runs).  After jx_pcg_set_dist every rank calls jx_rrblup_pcg_packed with ITS range of the kept rows (payload rows, value_lut
 * and out_beta of the shard; samples and y_train replicated): Z'p, mu'p and the three scalars of an iteration are summed over
 * the ranks through allreduce(user), which must add the first jx_pcg_dist_count() doubles of d_staging in place on every
 * rank; predictions and out_scalars are complete on every rank, out_beta is the shard's.  world <= 1 or allreduce NULL: off. */
int jx_pcg_set_dist(int rank, int world, int (*allreduce)(void *), void *user, double *d_staging, int64_t staging_doubles);
/* Image scope of the PCG / Haseman-Elston routes: between jx_pcg_image_scope(1) and jx_pcg_image_scope(0) the two images of the
 * training payload (SNP-major P32, sample-major T32) that jx_he_traces_packed / jx_rrblup_pcg_packed build stay in HBM, keyed by
 * (payload pointer, row list, training samples), and a second call on the same inputs reuses them (`jx gs -rrBLUP -rr-solver pcg`:
 * lambda by HE, then the solve).  The caller guarantees that the payload does not change inside the scope.  (0) frees them. */
int jx_pcg_image_scope(int on);

/* Releases the scratch blocks the library keeps between calls (the eigensolver's workspaces, at most 6 GB each; the reference's
 * CPU path has no counterpart: faer / LAPACK workspaces are freed per call, src/math/linalg.rs:120-184).  Blocks leased by a running call stay.
 * Returns the number of bytes handed back to the driver. */
int64_t jxg_scratch_trim(void);
int64_t jx_pcg_dist_count(void);
/* Exact marker-space rrBLUP on a resident payload: `rrblup_exact_snp_packed` (src/stats/rrblup.rs:3179-3490; cache
 * :1613-1899, fit :1951-2430).  A* = Z Z' - rs rs' / n_train over the training samples (f64), eigendecomposition, Brent on the
 * REML cost of the spectrum (:1568-1611) over log10 lambda in [low, high], beta = V diag(1 / (s + lambda)) V' Z y_c,
 * predictions alpha + Z' beta.  value_lut / row_indices / indices as for jx_rrblup_pcg_packed.  out_scalars (8):
 * pve_trainvar, lambda, REML, var_g, sigma_e2, rank, intercept alpha, mean of y_train. */
int jx_rrblup_exact_snp_packed(const uint8_t *packed, int64_t m_total, int n_samples, const int64_t *row_indices,
                               int64_t eff_m, const float *value_lut, const int64_t *train_idx, int n_train,
                               const double *y_train, const int64_t *test_idx, int n_test, double log10_lambda_low,
                               double log10_lambda_high, double reml_tol, int reml_max_iter, float *out_beta,
                               double *out_pred_train, double *out_pred_test, double *out_scalars);

/* `spgrm_packed_to_jxgrm` (src/stats/spgrm.rs:5201-5278 -> `spgrm_packed_to_jxgrm_core` :3769-3908) and, with
 * stream_denominator != 0, the stream core behind `spgrm_bed_to_jxgrm` (:3910-4264: denominator = sum of 2p(1-p) in
 * f64 whatever the sample selection, :3973-3989): GRM accumulation as jx_grm_packed, then A6; writes the file
 * (`write_sparse_grm_csc` :3745-3767: u64 n, u64 nnz, col_ptr u64 (n+1), row_indices u32 (nnz), zero padding to an
 * 8-byte boundary, values f64 (nnz), little endian) at out_path (already normalised, `normalize_spgrm_path` :450-469
 * is done by the caller).  Errors carry the reference's messages (`validate_spgrm_inputs` :2858-2915). */
int jx_spgrm_packed_to_jxgrm(const uint8_t *packed, int64_t m, int n_samples, const uint8_t *row_flip,
                             const float *row_maf, const int64_t *sample_indices, int n_sel, int method,
                             double threshold, int abs_threshold, int stream_denominator, const char *out_path,
                             int64_t *out_n, int64_t *out_nnz);

/* Several processes (one per GPU) build ONE sparse GRM (BASELINE.json configs[5]; the reference has no counterpart: its
 * tile plan runs on the threads of one process, src/stats/spgrm.rs:3769-3908).  jx_spgrm_set_part(part, nparts) makes the
 * next jx_spgrm_packed_to_jxgrm call of this process compute only the row panels dealt to `part` (back and forth over the
 * parts, so that the cost of the lower triangle's rows is even) and leave them in `<out_path>.part<part>` (out_nnz = the
 * entries of the part); after a barrier one process calls jx_spgrm_merge_parts, which joins the nparts files into the
 * `.spgrm` file of `write_sparse_grm_csc` (:3745-3767) -- byte-identical to the file of a single process -- and removes
 * them.  (0, 1) switches the mode off. */
int jx_spgrm_set_part(int part, int nparts);
int jx_spgrm_merge_parts(const char *out_path, int n, int nparts, int64_t *out_nnz);

/* Haseman-Elston sufficient statistics over the same matrix-free operator (`he_pcg_bed`, src/stats/he.rs:1633-2070,
 * 2101-2636): K = Z'Z / m_scale on the training samples, P the projector off [1, x_cov] (x_cov (n_train, p_cov)
 * row-major or NULL).  out5 = (y'PKPy, y'Py, tr(PKP), tr((PKP)^2), tr(P)); traces by `trace_samples` Rademacher probes
 * generated as the reference does (splitmix64 chains keyed by `seed`), or exactly (one probe per sample) when
 * exact_trace != 0.  The 2x2 HE normal equations are solved by the caller (janusx_amd/janusx.py `he_pcg_bed`). */
int jx_he_traces_packed(const uint8_t *packed, int64_t m_total, int n_samples, const int64_t *row_indices,
                        int64_t eff_m, const float *value_lut, const int64_t *train_idx, int n_train,
                        const double *y_train, const double *x_cov, int p_cov, int trace_samples, uint64_t seed,
                        int exact_trace, double m_scale, double *out5);

/* F2. Association TSV writer (host): replaces `append_assoc_block_from_arrays` / `append_assoc_row_from_fields` and the
 * writer thread behind them (src/io/assoc2tsv.rs:430-548, 604; src/stats/common.rs:374).  Row i = prefix i (the bytes
 * prefix_blob[prefix_off[i] .. prefix_off[i+1]) = `chrom\tpos\tsnp\tallele0\tallele1`) + af, miss, beta, se as `{:.4}`,
 * chisq = (beta/se)^2 and p as `{:.4e}` (Rust float text: unpadded exponent, `NaN`, `inf`; invalid beta / se -> chisq NaN,
 * p = 1, src/math/linalg.rs:111-121) [+ plrt | lambda, ml `{:.6e}`, plrt].  stats (rows, ncol) f64, ncol in {3, 4, 6}.
 * Returns the number of rows written, -1 on error (jx_last_error). */
int64_t jx_assoc_tsv_write(const char *path, const char *prefix_blob, const int64_t *prefix_off, int64_t rows,
                           const float *af, const float *miss, const double *stats, int ncol);
/* Same rows appended behind the present content of `path` (append bit 0 set: no header): the block-wise writer of the
 * streaming scan, counterpart of the reference's AsyncTsvWriter (src/stats/common.rs:374).  append bit 1 (+2): `miss` holds
 * counts of missing samples and is printed as an integer (`AssocMissValue::Count`, src/io/assoc2tsv.rs:452-458: the LM
 * routes) instead of a rate. */
int64_t jx_assoc_tsv_append(const char *path, const char *prefix_blob, const int64_t *prefix_off, int64_t rows,
                           const float *af, const float *miss, const double *stats, int ncol, int append);

/* ---- sparse symmetric solves on the device (csrc/k_spsolve.hip): SparseLMM on a relatedness graph with a connected component
 * beyond one dense eigenproblem.  The reference factorises K + lambda I sparsely on the host (src/math/cholesky.rs:733,
 * 1018-1183) and solves one system per SNP (`exact_scan_blocks_core`, src/stats/splmm.rs:2567-2880); here a block of decoded
 * SNP rows is the right-hand side of ONE Jacobi-preconditioned multi-vector CG over the CSR image (full symmetric pattern) of K.
 * Vectors are (n, ldr) row-major f64, ldr = jxg_sps_ldr(nrhs) (a multiple of 64): row i = entry i of every right-hand side.
 *   jxg_sps_rows_to_cols_f64: decoded rows (nrhs, ld) f32 -> that layout;
 *   jxg_sps_solve_multi:      X = (K + lambda I)^-1 B; d_dinv[i] = 1 / (K_ii + lambda); stops at |r| <= tol |b| for every
 *                             right-hand side; h_info (host) = [iterations, max |r| / |b|]; fails when max_iter is reached;
 *   jxg_sps_scan_sums:        per right-hand side [g'V^-1 g, g.Py, g.(V^-1 X)[:, k]] -> d_sums (nrhs, p + 2): what
 *                             jxg_fvlmm_finish_dev takes with ntiles = 1, score_mode = 1;
 *   d_work: jxg_sps_work_doubles(n, ldr) doubles, shared by the two calls. */
int jxg_sps_ldr(int nrhs);
int64_t jxg_sps_work_doubles(int n, int ldr);
int jxg_sps_rows_to_cols_f64(const float *d_rows, int nrhs, int n, int64_t ld, double *d_out, int ldr, void *stream);
int jxg_sps_solve_multi(int n, const int64_t *d_rowptr, const int32_t *d_col, const double *d_val, double lambda,
                        const double *d_dinv, const double *d_b, int nrhs, int ldr, double tol, int max_iter, double *d_x,
                        double *d_work, double *h_info, void *stream);
int jxg_sps_scan_sums(int n, const double *d_g, const double *d_z, int nrhs, int ldr, const double *d_py, const double *d_vinvx,
                      int p, double *d_sums, double *d_work, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* JXGPU_H */
