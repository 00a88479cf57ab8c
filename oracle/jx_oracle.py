"""CPU restatement (numpy) of the JanusX mixed-model hot path -- TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: a plain restatement of what the reference's Rust kernels compute,
written from the reference sources under /root/reference (cited per function as file:line).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``janusx_amd``) never does: it calls the HIP library through the C ABI and
fails loudly when that library is missing.

Parity pinning status -- which function is pinned by what (tests/test_oracle_golden.py runs every line of this table on CPU;
DESIGN.md section 4 "Oracle and parity" has the long form).  "Reference-produced" = computed HERE by the reference's own
importable Python (pyBLUP/assoc.py, blup.py, mlm.py with a stub in place of its native module; tests/golden/gen_fixtures.py,
gen_reference_model_fixtures.py, gen_cli_fixtures.py) and stored as data in tests/golden/:

  function(s) of this file                       pinned by
  ---------------------------------------------  ---------------------------------------------------------------------------
  pack_codes / unpack_codes / decode LUTs        the reference's own unit vectors, src/math/bedmath.rs:1537-1660
  genetic_model_apply, decode tables             src/decode/decode.rs:107-178 tables (test_genetic_model_tables)
  chi2_sf_df1, normal_sf                         src/math/linalg.rs:369-398 closed forms
  eigh_sym                                       src/math/eigh.rs:1982-1998 (2 x 2, eigenvalues {1, 3})
  reml_loglike / ml_loglike, lmm_reml_null       reference-produced: LMM._NULLREML values (panel_small.npz: nullreml_*),
                                                 blup.REML at 5 lambda, mlm.BLUP._REML incl. its v_floor branch
  spectral_null_model (sigma_g2, sigma_e2, PVE,  reference-produced: _lmm_profile_exact_vc; the whole model layer
    bounds), fvlmm cache, LM helpers             (LMM.from_spectral, FvLMM, LM) replayed over a recording stub
                                                 (reference_model.npz)
  CLI helpers (chunk sizes, names, k-fold)       reference-produced: cli_helpers.json
  StdRng / splitmix64 probes                     published ChaCha12 / splitmix64 vectors (test_stdrng_published_vectors)
  sparse REML (spreml_*)                         the reference's vectors in src/stats/spreml.rs tests
  brent_minimize                                 closed-form minima (test_brent_known_function); restated line by line
  ---------------------------------------------  ---------------------------------------------------------------------------
  PINNED BY READING ONLY (the reference ships no numeric test for them and its Rust crate cannot be built here -- no cargo /
  rustc): the per-SNP Brent trajectory inside `lmm_scan_rotated_block` (brent.rs is restated statement for statement),
  `final_beta_se`, the fixed-lambda formulas of fvlmm.rs, the QC threshold arithmetic of the row statistics, the GRM's f32 SYRK /
  f64 merge order, the TSV float formatting (`{:.4}` / `{:.4e}`: Python's formatter is IEEE-correct like Rust's), and the
  PCG32 seed expansion behind `splmm_choose_rhat_rows`.  For these the restatement follows the cited code with the same
  dtypes, constants and loop order, and the C twin (jx_oracle.c) is held equal to this file on the same inputs
  (test_python_vs_c_oracle).

All functions take/return numpy arrays; dtypes follow the reference (f32 where it uses f32).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import numpy as np

F32 = np.float32
F64 = np.float64
MIN_POSITIVE = np.finfo(np.float64).tiny  # f64::MIN_POSITIVE
EPS64 = np.finfo(np.float64).eps  # f64::EPSILON

# --------------------------------------------------------------------------------------------
# A.1  BED payload (src/math/bedmath.rs:20-27; src/io/gfreader.rs:1378-1395)
# --------------------------------------------------------------------------------------------

BED_MAGIC = bytes([0x6C, 0x1B, 0x01])


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """codes (m, n) uint8 in {0,1,2,3} -> packed (m, ceil(n/4)) uint8, sample j at bits 2*(j&3)
    of byte j>>2 (src/math/bedmath.rs:1528-1534 test helper `pack_codes`)."""
    codes = np.asarray(codes, dtype=np.uint8)
    if codes.ndim == 1:
        codes = codes[None, :]
    m, n = codes.shape
    bps = (n + 3) // 4
    pad = np.zeros((m, bps * 4), dtype=np.uint8)
    pad[:, :n] = codes & 3
    pad = pad.reshape(m, bps, 4)
    return (pad[:, :, 0] | (pad[:, :, 1] << 2) | (pad[:, :, 2] << 4) | (pad[:, :, 3] << 6)).astype(np.uint8)


def unpack_codes(packed: np.ndarray, n: int) -> np.ndarray:
    """packed (m, bps) uint8 -> 2-bit codes (m, n) uint8. 00->g=0, 10->g=1, 11->g=2, 01->missing
    (src/math/bedmath.rs:20-27)."""
    packed = np.asarray(packed, dtype=np.uint8)
    if packed.ndim == 1:
        packed = packed[None, :]
    m = packed.shape[0]
    shifts = np.array([0, 2, 4, 6], dtype=np.uint8)
    return ((packed[:, :, None] >> shifts) & 3).reshape(m, -1)[:, :n]


def genotypes_to_codes(g: np.ndarray) -> np.ndarray:
    """dosage (m,n) in {0,1,2,-9/negative=missing} -> BED 2-bit codes."""
    g = np.asarray(g)
    codes = np.full(g.shape, 1, dtype=np.uint8)  # 01 = missing
    codes[g == 0] = 0
    codes[g == 1] = 2
    codes[g == 2] = 3
    return codes


def row_counts(packed: np.ndarray, n_samples: int, sample_idx=None):
    """(missing, het, hom_alt) per SNP over the selected samples; pad bits ignored
    (src/io/gfreader.rs:1378-1395 `count_packed_row_counts`)."""
    codes = unpack_codes(packed, n_samples)
    if sample_idx is not None:
        codes = codes[:, np.asarray(sample_idx, dtype=np.int64)]
    missing = (codes == 1).sum(axis=1).astype(np.int64)
    het = (codes == 2).sum(axis=1).astype(np.int64)
    hom_alt = (codes == 3).sum(axis=1).astype(np.int64)
    return missing, het, hom_alt


# --------------------------------------------------------------------------------------------
# A.2  per-SNP stats and filters
# --------------------------------------------------------------------------------------------

def gwas_scan_row_stats(missing, het, hom_alt, n, maf_thr, miss_thr, het_thr):
    """GWAS streaming-scan QC (src/stats/lmm.rs:1258-1320): all compares in **f32**.
    Returns keep(bool), maf(f32 = alt_freq, the TSV `af`), miss_rate(f32), flip(all False)."""
    m = len(missing)
    maf_thr = F32(maf_thr)
    miss_thr = F32(miss_thr)
    het_thr = F32(het_thr)
    keep = np.zeros(m, dtype=bool)
    maf = np.zeros(m, dtype=np.float32)
    miss_rate = np.zeros(m, dtype=np.float32)
    for j in range(m):
        mis = int(missing[j])
        nm = max(n - mis, 0)
        mr = F32(mis) / F32(n) if n > 0 else F32(1.0)
        miss_rate[j] = mr
        if mr > miss_thr:
            continue
        if nm == 0:
            if maf_thr > F32(0.0):
                continue
            keep[j] = True
            maf[j] = F32(0.0)
            continue
        if het_thr > F32(0.0):
            if F32(int(het[j])) / F32(nm) > het_thr:
                continue
        alt_sum = int(het[j]) + 2 * int(hom_alt[j])
        alt_freq = F32(alt_sum) / (F32(2.0) * F32(nm))
        maf_v = min(alt_freq, F32(1.0) - alt_freq)
        if maf_v < maf_thr:
            continue
        keep[j] = True
        maf[j] = alt_freq
    return keep, maf, miss_rate, np.zeros(m, dtype=bool)


def packed_prep_row_stats(missing, het, hom_alt, n, maf_thr, miss_thr, het_thr):
    """Packed-workflow QC (`prepare_bed_2bit_packed`: src/io/gfreader.rs:5380-5420 keep rule,
    `packed_row_stats_from_counts` :1911-1929 and the subset loader :6340-6400 for the kept rows' columns).
    Returns keep, miss_rate (f32), maf (f32: the *alt allele frequency*, clamped to [0,1]), std_denom (f32),
    row_flip (all False)."""
    m = len(missing)
    maf_thr, miss_thr, het_thr = F32(maf_thr), F32(miss_thr), F32(het_thr)
    apply_het = het_thr > F32(0.0)
    keep = np.zeros(m, dtype=bool)
    miss = np.zeros(m, dtype=np.float32)
    maf = np.zeros(m, dtype=np.float32)
    std = np.zeros(m, dtype=np.float32)
    for j in range(m):
        nm = max(n - int(missing[j]), 0)
        alt_sum = int(het[j]) + 2 * int(hom_alt[j])
        mr = F32(n - nm) / F32(n) if n > 0 else F32(0.0)
        miss[j] = mr
        if nm > 0:
            p = alt_sum / (2.0 * nm)
            d = F32(math.sqrt(2.0 * p * (1.0 - p)))
            std[j] = d if math.isfinite(float(d)) else F32(0.0)
            af = F32(alt_sum) / (F32(2.0) * F32(nm))
        else:
            af = F32(0.0)
        maf[j] = min(max(af, F32(0.0)), F32(1.0))
        if mr > miss_thr:
            ok = False
        elif nm == 0:
            ok = bool(maf_thr <= F32(0.0))
        elif apply_het and (int(het[j]) / float(nm)) > float(het_thr):
            ok = False
        else:
            ok = bool(min(af, F32(1.0) - af) >= maf_thr)
        keep[j] = ok
    return keep, miss, maf, std, np.zeros(m, dtype=bool)


def stream_grm_row_prepare(missing, het, hom_alt, n_samples, method, maf_thr, miss_thr, het_thr,
                           eps=1e-12):
    """Stream-GRM per-row preparation, all in **f64** with f32 thresholds widened
    (src/stats/grm.rs:1465-1536 `grm_stream_row_prepare_from_counts_f32`).
    Thresholds are clamped by the caller (grm.rs:4709-4711).
    Returns keep, mean_g(f32), std_scale(f32), flip(bool), var(f64)."""
    m = len(missing)
    maf_thr64 = float(F32(maf_thr))
    miss_thr64 = float(F32(miss_thr))
    het_thr32 = F32(het_thr)
    eps64 = float(F32(eps))
    keep = np.zeros(m, dtype=bool)
    mean_g = np.zeros(m, dtype=np.float32)
    scale = np.zeros(m, dtype=np.float32)
    flip = np.zeros(m, dtype=bool)
    var = np.zeros(m, dtype=np.float64)
    for j in range(m):
        nm = n_samples - int(missing[j])
        if het_thr32 > F32(0.0) and nm > 0:
            if float(int(het[j])) / float(nm) > float(het_thr32):
                continue
        if n_samples == 0:
            continue
        missing_rate = 1.0 - (float(nm) / float(n_samples))
        if missing_rate > miss_thr64:
            continue
        if nm == 0:
            if F32(maf_thr) > F32(0.0):
                continue
            keep[j] = True
            mean_g[j] = 0.0
            scale[j] = 0.0 if method == 2 else 1.0
            continue
        alt_sum = float(int(het[j]) + 2 * int(hom_alt[j]))
        alt_freq = alt_sum / (2.0 * float(nm))
        fl = alt_freq > 0.5
        if fl:
            alt_sum = 2.0 * float(nm) - alt_sum
            alt_freq = alt_sum / (2.0 * float(nm))
        maf = min(alt_freq, 1.0 - alt_freq)
        if maf < maf_thr64:
            continue
        keep[j] = True
        flip[j] = fl
        mean_g[j] = F32(alt_sum / float(nm))
        v = max(2.0 * alt_freq * (1.0 - alt_freq), 0.0)
        var[j] = v
        if method == 2:
            scale[j] = F32(1.0 / math.sqrt(v)) if v > eps64 else F32(0.0)
        else:
            scale[j] = 1.0
    return keep, mean_g, scale, flip, var


# --------------------------------------------------------------------------------------------
# A.3  GRM
# --------------------------------------------------------------------------------------------

def grm_value_lut_f32(row_maf_f32, flip: bool, method: int, eps=1e-12):
    """4-entry f32 LUT indexed by the 2-bit code [00, 01(missing), 10, 11]
    (src/decode/decode.rs:813-839; src/math/bedmath.rs:1208-1224; decode.rs:558-566)."""
    p = F32(min(max(F32(row_maf_f32), F32(0.0)), F32(1.0)))
    mean_g = F32(2.0) * p
    var = F32(2.0) * p * (F32(1.0) - p)
    if method == 2:
        s = F32(1.0) / F32(np.sqrt(var)) if var > F32(eps) else F32(0.0)
    else:
        s = F32(1.0)
    g = (F32(2.0), F32(1.0), F32(0.0)) if flip else (F32(0.0), F32(1.0), F32(2.0))
    return np.array([(g[0] - mean_g) * s, F32(0.0), (g[1] - mean_g) * s, (g[2] - mean_g) * s],
                    dtype=np.float32)


def decode_grm_block_f32(packed, n_samples, row_flip, row_maf, sample_idx, method, r0, r1):
    """Design block Z (rows, n_out) f32 (src/decode/decode.rs:728-886). Subset+method 1 centres
    by the *passed* maf (src/math/bedmath.rs:1359-1441, test :1630)."""
    codes = unpack_codes(packed[r0:r1], n_samples)
    if sample_idx is not None:
        codes = codes[:, np.asarray(sample_idx, dtype=np.int64)]
    out = np.empty(codes.shape, dtype=np.float32)
    for k in range(r1 - r0):
        lut = grm_value_lut_f32(row_maf[r0 + k], bool(row_flip[r0 + k]), method)
        out[k] = lut[codes[k]]
    return out


def grm_varsum(row_maf, method, full_sample, n_out=None):
    """Denominator D (src/stats/grm.rs:91-111 for the full-sample centred case; the subset route
    accumulates `var_global_centered` computed in f32 per row, bedmath.rs:1411-1412, 1437-1438)."""
    row_maf = np.asarray(row_maf, dtype=np.float32)
    if method != 1:
        return float(len(row_maf))
    if full_sample:
        acc = 0.0
        for maf in row_maf:
            p = float(maf)
            v = 2.0 * p * (1.0 - p)
            if math.isfinite(v) and v > 0.0:
                acc += v
        return acc
    acc = 0.0
    for maf in row_maf:
        p0 = F32(min(max(F32(maf), F32(0.0)), F32(1.0)))
        mean_g = F32(2.0) * p0
        pg = F32(min(max(F32(0.5) * mean_g, F32(0.0)), F32(1.0)))
        v = max(F32(2.0) * pg * (F32(1.0) - pg), F32(0.0))
        acc += float(v)
    return acc


def grm_packed(packed, n_samples, row_flip, row_maf, sample_idx=None, method=1, block_rows=65536,
               out_dtype=np.float32, exact_f64=False):
    """`grm_packed_f32` / `_f64` restatement (src/stats/grm.rs:204-360, 3066):
    per SNP block T_b = Z_b^T Z_b by an **f32** SYRK (numpy sgemm here), blocks merged in **f64**
    (grm.rs:1638-1667), K = acc * (1/D), mirrored (grm.rs:2771-2785), cast to `out_dtype`.
    exact_f64=True accumulates everything in f64 (the `_f64` API, grm.rs:3013)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    m = packed.shape[0]
    if method not in (1, 2):
        raise RuntimeError(f"unsupported method={method}; expected 1 (centered) or 2 (standardized)")
    if m == 0:
        raise RuntimeError("packed must contain at least one SNP row")
    full = sample_idx is None or (len(sample_idx) == n_samples and
                                  np.array_equal(np.asarray(sample_idx), np.arange(n_samples)))
    n = n_samples if sample_idx is None else len(sample_idx)
    D = grm_varsum(row_maf, method, full, n)
    if not (math.isfinite(D) and D > 0.0):
        raise RuntimeError("invalid centered GRM denominator: sum(2p(1-p)) <= 0")
    acc = np.zeros((n, n), dtype=np.float64)
    step = max(1, int(block_rows))
    for r0 in range(0, m, step):
        r1 = min(m, r0 + step)
        z = decode_grm_block_f32(packed, n_samples, row_flip, row_maf, None if full else sample_idx,
                                 method, r0, r1)
        if exact_f64:
            z64 = z.astype(np.float64)
            acc += z64.T @ z64
        else:
            acc += (z.T @ z).astype(np.float64)  # f32 GEMM, f64 merge
    inv = 1.0 / D
    k = acc * inv
    k = np.tril(k) + np.tril(k, -1).T  # mirror lower -> upper
    return k.astype(out_dtype), D


def spgrm_keep_value(value: float, threshold: float, abs_threshold: bool) -> bool:
    """Off-diagonal keep rule of the sparse GRM (src/stats/spgrm.rs:1956-1965; tests :6618-6630)."""
    if abs_threshold:
        return abs(value) > threshold
    if threshold < 0.0:
        return True
    return value > threshold


def sparse_grm_csc_from_packed(packed, n_samples, row_flip, row_maf, sample_idx=None, method=1, threshold=0.05,
                               abs_threshold=False, stream_denominator=False, block_rows=65536):
    """Sparse GRM of `spgrm_packed_to_jxgrm` / `sparse_grm_coo_from_packed` + `coo_lower_to_csc`
    (src/stats/spgrm.rs:3769-3908, 3556-3683): validation `validate_spgrm_inputs` (:2858-2915); denominator
    `centered_varsum_from_packed` (:2917-2979: full sample -> sum of 2p(1-p) in f64, subset -> the decode's f32
    `var_global_centered` sums) or, for the stream core behind `spgrm_bed_to_jxgrm`, always the f64 sum (:3973-3989);
    per sample-tile pair f32 GEMM blocks merged in f64, `scaled = acc * (1 / denom)`, entry kept on the diagonal or by
    `spgrm_keep_value`, non-finite -> error (`compute_spgrm_task_entries` :3422-3554); (col, row) order (:1401).
    Returns (col_ptr u64 (n+1), row_indices u32 (nnz), values f64 (nnz))."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    if n_samples == 0:
        raise RuntimeError("Sparse GRM requires n_samples > 0")
    if method not in (1, 2):
        raise RuntimeError(f"Sparse GRM method must be 1 (centered) or 2 (standardized); got {method}")
    if not math.isfinite(threshold):
        raise RuntimeError("Sparse GRM threshold must be finite")
    if sample_idx is not None and len(sample_idx) == 0:
        raise RuntimeError("Sparse GRM sample_indices must not be empty")
    m = len(row_flip)
    if m == 0:
        raise RuntimeError("Sparse GRM requires at least one SNP row")
    if len(row_maf) != m:
        raise RuntimeError(f"Sparse GRM row_maf length mismatch: got {len(row_maf)}, expected {m}")
    full = sample_idx is None or (len(sample_idx) == n_samples and
                                  np.array_equal(np.asarray(sample_idx), np.arange(n_samples)))
    n = n_samples if sample_idx is None else len(sample_idx)
    denom = grm_varsum(row_maf, method, full or stream_denominator, n)
    if not (math.isfinite(denom) and denom > 0.0):
        raise RuntimeError("Sparse GRM centered denominator is not positive" if method == 1
                           else "Sparse GRM denominator is not positive")
    acc = np.zeros((n, n), dtype=np.float64)
    for r0 in range(0, m, max(1, int(block_rows))):
        r1 = min(m, r0 + max(1, int(block_rows)))
        z = decode_grm_block_f32(packed, n_samples, row_flip, row_maf, None if full else sample_idx, method, r0, r1)
        acc += (z.T @ z).astype(np.float64)        # f32 GEMM per SNP block, f64 merge
    scaled = acc * (1.0 / denom)
    if not np.isfinite(scaled).all():
        raise RuntimeError("Sparse GRM produced non-finite value")
    col_ptr = np.zeros(n + 1, dtype=np.uint64)
    rows_out, vals_out = [], []
    for c in range(n):
        col = scaled[c:, c]
        if abs_threshold:
            keep = np.abs(col) > threshold
        elif threshold < 0.0:
            keep = np.ones(col.shape, dtype=bool)
        else:
            keep = col > threshold
        keep[0] = True                             # the diagonal is always stored
        r = np.nonzero(keep)[0]
        rows_out.append((r + c).astype(np.uint32))
        vals_out.append(col[r])
        col_ptr[c + 1] = col_ptr[c] + np.uint64(len(r))
    return col_ptr, np.concatenate(rows_out), np.concatenate(vals_out).astype(np.float64)


def sparse_grm_csc_from_dense(k, threshold=0.05, abs_threshold=False):
    """`spgrm_dense_f32_to_jxgrm_core` (src/stats/spgrm.rs:5027-5101): lower triangle of the stored dense matrix
    (entry [row, col], row >= col), widened to f64, diagonal always kept, `spgrm_keep_value` otherwise; a non-finite
    entry is an error.  -> (col_ptr u64, row_indices u32, values f64)."""
    if not math.isfinite(threshold):
        raise RuntimeError("Sparse GRM threshold must be finite")
    k = np.asarray(k)
    n = k.shape[0]
    if n == 0:
        raise RuntimeError("Sparse GRM dense writer requires n_samples > 0")
    col_ptr = np.zeros(n + 1, dtype=np.uint64)
    rows_out, vals_out = [], []
    for c in range(n):
        col = k[c:, c].astype(np.float64)
        if not np.isfinite(col).all():
            r = int(np.nonzero(~np.isfinite(col))[0][0]) + c
            raise RuntimeError(f"Sparse GRM dense writer found non-finite value at pair ({r}, {c})")
        if abs_threshold:
            keep = np.abs(col) > threshold
        elif threshold < 0.0:
            keep = np.ones(col.shape, dtype=bool)
        else:
            keep = col > threshold
        keep[0] = True
        r = np.nonzero(keep)[0]
        rows_out.append((r + c).astype(np.uint32))
        vals_out.append(col[r])
        col_ptr[c + 1] = col_ptr[c] + np.uint64(len(r))
    return col_ptr, np.concatenate(rows_out), np.concatenate(vals_out)


def normalize_spgrm_path(prefix: str) -> str:
    """Output path of the sparse GRM (src/stats/spgrm.rs:450-469)."""
    import os
    t = prefix.strip()
    if not t:
        return ""
    if t.lower().endswith(".spgrm") or t.lower().endswith(".jxgrm"):
        return t
    if os.path.exists(t + ".jxgrm") and not os.path.exists(t + ".spgrm"):
        return t + ".jxgrm"
    return t + ".spgrm"


def write_sparse_grm_csc(path, n_samples, col_ptr, row_indices, values):
    """`.spgrm` layout (src/stats/spgrm.rs:3745-3767; padding :810-820): u64 n, u64 nnz, col_ptr u64 (n+1),
    row_indices u32 (nnz), zero padding up to a multiple of 8 bytes of the row payload, values f64 (nnz); LE."""
    nnz = len(values)
    pad = (-(nnz * 4)) % 8
    with open(path, "wb") as fh:
        fh.write(np.array([n_samples, nnz], dtype="<u8").tobytes())
        fh.write(np.asarray(col_ptr, dtype="<u8").tobytes())
        fh.write(np.asarray(row_indices, dtype="<u4").tobytes())
        fh.write(b"\0" * pad)
        fh.write(np.asarray(values, dtype="<f8").tobytes())


def read_sparse_grm_csc(path):
    """Inverse of `write_sparse_grm_csc` -> (n, col_ptr, row_indices, values); checks the total length."""
    raw = open(path, "rb").read()
    n, nnz = (int(v) for v in np.frombuffer(raw[:16], dtype="<u8"))
    at = 16
    col_ptr = np.frombuffer(raw[at:at + 8 * (n + 1)], dtype="<u8")
    at += 8 * (n + 1)
    rows = np.frombuffer(raw[at:at + 4 * nnz], dtype="<u4")
    at += 4 * nnz
    pad = (-(nnz * 4)) % 8
    if raw[at:at + pad] != b"\0" * pad:
        raise RuntimeError("sparse GRM padding is not zero")
    at += pad
    vals = np.frombuffer(raw[at:at + 8 * nnz], dtype="<f8")
    if at + 8 * nnz != len(raw):
        raise RuntimeError(f"sparse GRM file length mismatch: {len(raw)} != {at + 8 * nnz}")
    return n, col_ptr, rows, vals


# --------------------------------------------------------------------------------------------
# Sparse REML null model over a `.spgrm` (src/stats/spreml.rs; first consumer of the sparse GRM in `-splmm`)
# --------------------------------------------------------------------------------------------

def sparse_grm_dense_subset(n, col_ptr, row_indices, values, sample_idx=None):
    """Dense symmetric K of the lower-triangle CSC, optionally K[idx][:, idx] in the given order
    (`subset_sparse_grm_csc`, src/math/cholesky.rs:618-690: empty / out-of-range / duplicated indices are errors;
    reference vector :1656-1669)."""
    col_ptr = np.asarray(col_ptr, dtype=np.int64)
    k = np.zeros((n, n), dtype=np.float64)
    for c in range(n):
        for q in range(int(col_ptr[c]), int(col_ptr[c + 1])):
            r = int(row_indices[q])
            k[r, c] = values[q]
            k[c, r] = values[q]
    if sample_idx is None:
        return k
    idx = np.asarray(sample_idx, dtype=np.int64)
    if idx.size == 0:
        raise RuntimeError("Sparse GRM subset requires at least one sample")
    if (idx < 0).any() or (idx >= n).any():
        raise RuntimeError(f"Sparse GRM subset index out of range for n_samples={n}")
    seen = set()
    for v in idx.tolist():
        if v in seen:
            raise RuntimeError(f"Sparse GRM subset contains duplicated sample index: {v}")
        seen.add(v)
    return np.ascontiguousarray(k[np.ix_(idx, idx)])


def spreml_design_matrix(x_cov, n):
    """[1 | x_cov] row-major (src/stats/spreml.rs:296-322)."""
    if n == 0:
        raise RuntimeError("SPREML requires n > 0")
    if x_cov is None:
        return np.ones((n, 1), dtype=np.float64)
    x_cov = np.asarray(x_cov, dtype=np.float64)
    if x_cov.ndim != 2 or x_cov.shape[0] != n:
        raise RuntimeError(f"x_cov shape mismatch: got {list(x_cov.shape)}, expected ({n}, p)")
    return np.concatenate([np.ones((n, 1)), x_cov], axis=1)


def spd_cholesky_with_jitter(mat, label):
    """src/stats/spreml.rs:324-351: plain Cholesky, then up to eight diagonal jitters base * 10^k,
    base = max(mean |diag|, 1) * 1e-10."""
    dim = mat.shape[0]
    chol = np.array(mat, dtype=np.float64)
    if cholesky_inplace(chol):
        return chol
    base = max(float(np.abs(np.diag(mat)).sum()) / max(dim, 1), 1.0) * 1e-10
    for k in range(8):
        chol = np.array(mat, dtype=np.float64)
        chol[np.diag_indices(dim)] += base * 10.0 ** k
        if cholesky_inplace(chol):
            return chol
    raise RuntimeError(f"{label} is not SPD even after diagonal jitter")


class SpdFactor:
    """K + lambda I factorised once, with `solve` and `logdet`.  The reference holds a sparse supernodal LLT (faer 0.18.2,
    src/math/cholesky.rs:776-1075); restated as a dense Cholesky for a dense K (what the reference's own tests compare
    against, src/stats/spreml.rs:1205-1329) and, for a scipy.sparse K, as a sparse LU without pivoting of the same SPD matrix
    (identical solve / log-determinant, O(nnz) memory: the checker of sample counts where no dense image fits the host)."""

    def __init__(self, k, lam):
        import scipy.sparse as sp
        self.sparse = sp.issparse(k)
        if self.sparse:
            from scipy.sparse.linalg import splu
            n = k.shape[0]
            m = (k + lam * sp.identity(n, format="csc")).tocsc()
            self.lu = splu(m, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
            piv = self.lu.U.diagonal()
            if not (np.all(np.isfinite(piv)) and np.all(piv > 0.0)):
                raise np.linalg.LinAlgError("K + lambda I is not positive definite")
            self.logdet = float(np.log(piv).sum())
        else:
            n = np.asarray(k).shape[0]
            self.lfac = np.linalg.cholesky(np.array(k, dtype=np.float64) + lam * np.eye(n))
            self.logdet = 2.0 * float(np.log(np.diag(self.lfac)).sum())

    def solve(self, b):
        if self.sparse:
            return self.lu.solve(np.ascontiguousarray(b, dtype=np.float64))
        import scipy.linalg as sla
        return sla.cho_solve((self.lfac, True), b)


def spreml_evaluate(k_dense, x_design, y, log10_lambda, vp_fixed=None):
    """`evaluate_sparse_reml_at_lambda` (src/stats/spreml.rs:384-512) with the sparse LLT of K + lambda I restated as a
    dense Cholesky (the reference's own tests compare against exactly that, :1205-1262, 1264-1329).
    vp_fixed=None: profile objective; else the fastGWA fixed-Vp objective.
    -> dict(log10_lambda, lambda, sigma_g2, sigma_e2, ml, reml)."""
    lam = 10.0 ** log10_lambda
    if not (math.isfinite(lam) and lam > 0.0):
        raise RuntimeError(f"SPREML lambda is invalid at log10(lambda)={log10_lambda}")
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[0]
    p = x_design.shape[1]
    if p == 0 or n <= p:
        raise RuntimeError(f"SPREML requires n > p, got n={n}, p={p}")
    try:
        fac = SpdFactor(k_dense, lam)
    except np.linalg.LinAlgError:
        raise RuntimeError(f"sparse Cholesky of K + lambda I failed at lambda={lam}")
    rhs = np.concatenate([y[:, None], x_design], axis=1)
    sol = fac.solve(rhs)
    y_vinv, x_vinv = sol[:, 0], sol[:, 1:]
    y_vinv_y = float(y @ y_vinv)
    xt_vinv_y = x_design.T @ y_vinv
    xt_vinv_x = x_design.T @ x_vinv
    cx = spd_cholesky_with_jitter(xt_vinv_x, "SPREML XtVinvX")
    beta = cholesky_solve(cx, xt_vinv_y)
    ypy = y_vinv_y - float(xt_vinv_y @ beta)
    if not math.isfinite(ypy) or ypy <= 1e-30:
        raise RuntimeError(f"SPREML profiled residual quadratic form is invalid at lambda={lam}: yPy={ypy}")
    df = float(n - p)
    log_det_m = fac.logdet
    log_det_x = 2.0 * float(np.log(np.diag(cx)).sum())
    if vp_fixed is None:
        sigma_g2 = ypy / df
        if not math.isfinite(sigma_g2) or sigma_g2 <= 0.0:
            raise RuntimeError(f"SPREML sigma_g2 is invalid at lambda={lam}: sigma_g2={sigma_g2}")
        sigma_e2 = lam * sigma_g2
        reml = df * (math.log(df) - 1.0 - math.log(2.0 * math.pi)) * 0.5 - 0.5 * (df * math.log(ypy) + log_det_m + log_det_x)
        nf = float(n)
        ml = nf * (math.log(nf) - 1.0 - math.log(2.0 * math.pi)) * 0.5 - 0.5 * (nf * math.log(ypy) + log_det_m)
    else:
        if not (math.isfinite(vp_fixed) and vp_fixed > 0.0):
            raise RuntimeError(f"SPREML fastGWA fixed-Vp objective requires finite vp_fixed > 0, got {vp_fixed}")
        sigma_g2 = vp_fixed / (1.0 + lam)
        sigma_e2 = lam * sigma_g2
        reml = -0.5 * (df * math.log(sigma_g2) + log_det_m + log_det_x + ypy / sigma_g2)
        ml = float("nan")
    if not math.isfinite(reml) or not (math.isfinite(ml) or math.isnan(ml)):
        raise RuntimeError(f"SPREML likelihood is invalid at lambda={lam}: ml={ml}, reml={reml}")
    return dict(log10_lambda=log10_lambda, lam=lam, sigma_g2=sigma_g2, sigma_e2=sigma_e2, ml=ml, reml=reml)


def refine_monotone_valid_lower_bound(is_valid, invalid_log10, valid_log10, tol, max_iter):
    """src/stats/spreml.rs:152-186 (reference vector :1197-1207)."""
    if not (math.isfinite(invalid_log10) and math.isfinite(valid_log10) and invalid_log10 < valid_log10):
        return valid_log10
    if not is_valid(valid_log10):
        return valid_log10
    if is_valid(invalid_log10):
        return invalid_log10
    tol_use = max(abs(tol), 1e-6)
    lo, hi = invalid_log10, valid_log10
    for _ in range(max(int(max_iter), 1)):
        if abs(hi - lo) <= tol_use:
            break
        mid = 0.5 * (lo + hi)
        if is_valid(mid):
            hi = mid
        else:
            lo = mid
    return hi


def spreml_grid_search(evaluate, low, high, grid_size):
    """`sparse_reml_grid_search_core` (src/stats/spreml.rs:514-589): failed points are skipped, strict > keeps the
    first best. `evaluate(log10_lambda)` returns the evaluation dict or raises. -> (best, grid list)."""
    if not (math.isfinite(low) and math.isfinite(high)) or low >= high:
        raise RuntimeError(f"SPREML grid search requires finite low < high, got low={low}, high={high}")
    grid_n = max(int(grid_size), 2)
    evals, best, first_err = [], None, None
    for idx in range(grid_n):
        x = low + (high - low) * (idx / (grid_n - 1))
        try:
            ev = evaluate(x)
        except RuntimeError as e:
            if first_err is None:
                first_err = str(e)
            continue
        if best is None or ev["reml"] > best["reml"]:
            best = ev
        evals.append(ev)
    if best is None:
        tail = f"; first failure: {first_err}" if first_err else ""
        raise RuntimeError(f"SPREML sparse grid search found no valid lambda in [{low}, {high}]{tail}")
    return best, evals


def spreml_brent_search(evaluate, is_factorizable, low, high, grid_size, tol, max_iter):
    """`sparse_reml_brent_search_with_progress` (src/stats/spreml.rs:591-757): grid, then Brent on -reml between the
    grid neighbours of the best point started from it; when the best point is the first valid one the lower end is
    refined towards the factorisability edge (24 bisections, tolerance min(tol, 1e-2)); failed evaluations cost 1e300;
    the result is re-evaluated at the Brent minimiser. -> (best, grid list)."""
    if not (math.isfinite(tol) and tol > 0.0):
        raise RuntimeError(f"SPREML Brent tol must be finite and > 0, got {tol}")
    if int(max_iter) == 0:
        raise RuntimeError("SPREML Brent max_iter must be > 0")
    grid_n = max(int(grid_size), 2)
    best, grid = spreml_grid_search(evaluate, low, high, grid_size)
    best_idx = next((i for i, ev in enumerate(grid) if ev["log10_lambda"] == best["log10_lambda"]), 0)
    b_low = grid[best_idx - 1]["log10_lambda"] if best_idx > 0 else low
    b_high = grid[best_idx + 1]["log10_lambda"] if best_idx + 1 < len(grid) else high
    if best_idx == 0 and grid and grid_n > 1:
        raw_step = (high - low) / (grid_n - 1)
        first_valid = grid[0]["log10_lambda"]
        prev_raw = max(first_valid - raw_step, low)
        if prev_raw < first_valid:
            b_low = refine_monotone_valid_lower_bound(
                lambda x: (math.isfinite(10.0 ** x) and 10.0 ** x > 0.0 and is_factorizable(10.0 ** x)),
                prev_raw, first_valid, min(tol, 1e-2), 24)
    if not (math.isfinite(b_low) and math.isfinite(b_high) and b_low < b_high):
        b_low, b_high = low, high

    def cost(x):
        try:
            return -evaluate(x)["reml"]
        except RuntimeError:
            return 1e300

    x_best, _, _ = brent_minimize(cost, b_low, b_high, tol, max_iter, init_x=best["log10_lambda"])
    return evaluate(x_best), grid


def spreml_sparse_reml_brent(n, col_ptr, row_indices, values, y, x_cov=None, sample_idx=None, low=-5.0, high=5.0,
                             grid_size=9, tol=1e-3, max_iter=20, vp_fixed=None, grid_only=False):
    """`spreml_sparse_reml_brent_from_jxgrm` / `_grid_from_jxgrm` / `spreml_sparse_fastgwa_fixed_vp_brent_from_jxgrm`
    (src/stats/spreml.rs:839-1160) on an in-memory CSC -> the reference's 10-tuple
    (lambda, sigma_g2, sigma_e2, ml, reml, log10_lambda, grid_log10, grid_reml, grid_sigma_g2, grid_sigma_e2)."""
    y = np.asarray(y, dtype=np.float64).ravel()
    k = sparse_grm_dense_subset(n, col_ptr, row_indices, values, sample_idx)
    if k.shape[0] != y.shape[0]:
        raise RuntimeError(f"SPREML subset sample size mismatch: sparse n={k.shape[0]}, phenotype n={y.shape[0]}")
    x = spreml_design_matrix(x_cov, y.shape[0])

    def evaluate(v):
        return spreml_evaluate(k, x, y, v, vp_fixed)

    def is_fact(lam):
        try:
            np.linalg.cholesky(k + lam * np.eye(k.shape[0]))
            return True
        except np.linalg.LinAlgError:
            return False

    if grid_only:
        best, grid = spreml_grid_search(evaluate, low, high, grid_size)
    else:
        best, grid = spreml_brent_search(evaluate, is_fact, low, high, grid_size, tol, max_iter)
    return (best["lam"], best["sigma_g2"], best["sigma_e2"], best["ml"], best["reml"], best["log10_lambda"],
            [g["log10_lambda"] for g in grid], [g["reml"] for g in grid], [g["sigma_g2"] for g in grid],
            [g["sigma_e2"] for g in grid])


def splmm_wald_from_score_denom(score, denom, sigma2):
    """src/stats/splmm.rs:2517-2538 -> (beta, se, p) or None."""
    if not (math.isfinite(score) and math.isfinite(denom) and denom > 1e-30 and math.isfinite(sigma2) and sigma2 > 0.0):
        return None
    beta = score / denom
    var_beta = sigma2 / denom
    if not (math.isfinite(beta) and math.isfinite(var_beta) and var_beta > 0.0):
        return None
    se = math.sqrt(var_beta)
    chisq = (score * score) / (sigma2 * denom)
    if not (math.isfinite(se) and se > 0.0 and math.isfinite(chisq) and chisq >= 0.0):
        return None
    return beta, se, chi2_sf_df1(chisq)


def splmm_exact_scan(k_dense, lam, x_design, y, packed, n_samples, maf, row_flip, sample_idx=None, rows=None,
                     dense_rows=None):
    """SparseLMM exact scan (`exact_scan_blocks_core`, src/stats/splmm.rs:2567-2880) with the sparse factor of
    V = K + lambda I restated as a dense Cholesky: null state Py = V^-1 (y - X b), yPy, sigma2 = yPy / (n - p),
    chol(X'V^-1 X); per SNP the mean-imputed additive f32 decode ([0, 2 maf, 1, 2] or flipped, NOT centred:
    `decode_mean_imputed_additive_packed_block_rows_f32`, src/math/bedmath.rs:940-1010), score = f32 GEMV g . f32(Py),
    z = V^-1 g in f64, g'Pg = max(g'z - c'A^-1 c, 0) with c = X'z, then `splmm_wald_from_score_denom`;
    failed rows are (NaN, NaN, 1).  -> (m, 3) f64."""
    y = np.asarray(y, dtype=np.float64)
    n, p = x_design.shape
    fac = SpdFactor(k_dense, lam)                      # dense Cholesky, or a sparse factor for a scipy.sparse K
    sol = fac.solve(np.concatenate([y[:, None], x_design], axis=1))
    xt_vinv_x = x_design.T @ sol[:, 1:]
    cx = spd_cholesky_with_jitter(xt_vinv_x, "SparseLMM XtWX")
    beta0 = cholesky_solve(cx, x_design.T @ sol[:, 0])
    py = sol[:, 0] - sol[:, 1:] @ beta0
    ypy = float(y @ py)
    df = float(n - p)
    if not (math.isfinite(ypy) and ypy > 0.0):
        raise RuntimeError(f"SparseLMM exact scan requires finite positive yPy on K + lambda I scale, got {ypy}")
    sigma2 = ypy / df
    py32 = py.astype(np.float32)
    if dense_rows is not None:
        # `splmm_assoc_pcg_dense_f32` (src/stats/splmm.rs:5464-5650): already decoded f32 rows, taken as they are
        dense_rows = np.asarray(dense_rows, dtype=np.float32)
        codes = None
        rows = np.arange(dense_rows.shape[0]) if rows is None else np.asarray(rows, dtype=np.int64)
    else:
        codes = unpack_codes(np.ascontiguousarray(packed, dtype=np.uint8), n_samples)
        if sample_idx is not None:
            codes = codes[:, np.asarray(sample_idx, dtype=np.int64)]
        rows = np.arange(codes.shape[0]) if rows is None else np.asarray(rows, dtype=np.int64)
    out = np.empty((len(rows), 3), dtype=np.float64)
    for k, r in enumerate(rows):
        if codes is None:
            g32 = dense_rows[r]
        else:
            mean_g = F32(min(max(F32(2.0) * F32(maf[r]), F32(0.0)), F32(2.0)))
            lut = np.array([2.0, mean_g, 1.0, 0.0] if row_flip[r] else [0.0, mean_g, 1.0, 2.0], dtype=np.float32)
            g32 = lut[codes[r]]
        score = float(np.dot(g32, py32))                      # f32 GEMV output
        g = g32.astype(np.float64)
        z = fac.solve(g)
        c = x_design.T @ z
        x_quad = float(c @ cholesky_solve(cx, c))
        res = splmm_wald_from_score_denom(score, max(float(g @ z) - x_quad, 0.0), sigma2)
        out[k] = res if res is not None else (float("nan"), float("nan"), 1.0)
    return out


# ---- SparseLMM approximate (GRAMMAR-gamma / fastGWA-style) route: `jx gwas -splmm` -----------------------------------------

def _chacha_block(key_words, counter, rounds, stream=(0, 0)):
    """One 64-byte ChaCha block (D. J. Bernstein's original layout: 64-bit block counter in words 12-13, 64-bit stream id
    in words 14-15, 0 for `StdRng`), `rounds` rounds -> 16 u32 words."""
    def rotl(v, c):
        return ((v << c) & 0xffffffff) | (v >> (32 - c))
    st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(key_words) + [counter & 0xffffffff, (counter >> 32) & 0xffffffff, stream[0], stream[1]]
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 7)
    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + st[i]) & 0xffffffff for i in range(16)]


class StdRngU32:
    """`StdRng::seed_from_u64(seed)` of the rand crate 0.9.2 (Cargo.toml:42; the crate itself is NOT in /root/reference and
    there is no Cargo.lock): ChaCha12 keyed by 32 bytes that rand_core's default `seed_from_u64` draws from a PCG32 stream
    (multiplier 6364136223846793005, increment 11634580027462260723, state advanced before every output, XSH-RR output,
    little-endian), words handed out in block order.  Restated from the published algorithm; **parity unpinned** (no
    golden vector of the reference depends on it, and the crate cannot be run here).  The ChaCha core itself is pinned to
    RFC 7539 section 2.3.2 with 20 rounds in tests/test_oracle_golden.py."""

    def __init__(self, seed):
        state = int(seed) & 0xffffffffffffffff
        key = []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & 0xffffffffffffffff
            xorshifted = (((state >> 18) ^ state) >> 27) & 0xffffffff
            rot = state >> 59
            key.append(((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & 0xffffffff)
        self.key, self.counter, self.buf = key, 0, []

    def next_u32(self):
        if not self.buf:
            self.buf = _chacha_block(self.key, self.counter, 12)
            self.counter += 1
        return self.buf.pop(0)

    def next_u64(self):
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)

    def random_range(self, m):
        """`rng.random_range(0..m)` for usize (rand 0.9 `UniformUsize::sample_single`: ranges that fit 32 bits are sampled
        as u32 for portability, others as u64) with Canon's biased single-sample method: widening multiply, one more draw
        only when the low half could carry."""
        if m <= 0:
            raise ValueError("empty range")
        if m - 1 > 0xffffffff:
            bits, draw = 64, self.next_u64
        else:
            bits, draw = 32, self.next_u32
        mask = (1 << bits) - 1
        prod = draw() * m
        result, lo = prod >> bits, prod & mask
        if lo > ((-m) & mask):
            new_hi = (draw() * m) >> bits
            if lo + new_hi > mask:
                result += 1
        return result


def choose_rhat_rows(m, count, seed):
    """src/stats/splmm.rs:1493-1507: all rows when count >= m, else 2 count draws with replacement, sorted, deduplicated."""
    soft_cap = min(int(count), int(m))
    if soft_cap == m:
        return np.arange(m, dtype=np.int64)
    rng = StdRngU32(seed)
    return np.unique(np.array([rng.random_range(m) for _ in range(max(2 * soft_cap, soft_cap))], dtype=np.int64))


def splmm_xtx_chol(x_design):
    """`xtx_chol_from_design` (src/stats/splmm_approx.rs:310-328)."""
    return spd_cholesky_with_jitter(x_design.T @ x_design, "SparseLMM approx XtX")


def splmm_residualize(x_design, cx, v):
    """`residualize_vector_with_chol` (src/stats/splmm_approx.rs:400-428): v - X (X'X)^-1 X'v."""
    return v - x_design @ cholesky_solve(cx, x_design.T @ v)


def splmm_additive_row_f64(codes_row, maf32, flip, model="add"):
    """`decode_packed_row_model_into_f64` (src/decode/decode.rs:307-364): missing -> max(2 maf, 0) in f64, flipped calls
    2 - g, then the genetic model applied to EVERY entry, the imputed one included (`gm.apply`, :383); not centred."""
    mean_g = max(2.0 * float(F32(maf32)), 0.0)
    vals = [2.0, mean_g, 1.0, 0.0] if flip else [0.0, mean_g, 1.0, 2.0]
    lut = np.array([genetic_model_apply(model, v) for v in vals], dtype=np.float64)
    return lut[codes_row]


def splmm_approx_null(k, x_design, y, lam):
    """`build_residualized_approx_scan_null_from_lambda_and_factor` (src/stats/splmm_approx.rs:612-699): y_r = M_X y,
    sigma2 = y_r'y_r / ((n - p)(1 + lambda)), a = (K + lambda I)^-1 y_r / sigma2, gamma scale 1 / sigma2.
    -> (factor, y_resid, a_vec, sigma2_scan)."""
    y = np.asarray(y, dtype=np.float64)
    n, p = x_design.shape
    if n <= p:
        raise RuntimeError(f"SparseLMM residualized approx requires n > p, got n={n}, p={p}")
    cx = splmm_xtx_chol(x_design)
    y_resid = splmm_residualize(x_design, cx, y)
    rss = float(y_resid @ y_resid)
    if not (math.isfinite(rss) and rss > 1e-30):
        raise RuntimeError(f"SparseLMM residualized approx produced invalid residualized RSS: {rss}")
    sigma2 = rss / (float(n - p) * (1.0 + lam))
    fac = SpdFactor(k, lam)
    return fac, y_resid, fac.solve(y_resid) / sigma2, sigma2


def splmm_estimate_gamma(fac, x_design, markers, a_vec, soft_cap, gamma_scale):
    """`estimate_gamma_from_markers` (src/stats/splmm_approx.rs:921-1068): per sampled marker g_r = M_X g, ratio =
    g_r'V^-1 g_r / g_r'g_r; markers whose score chi-square (g_r.a)^2 / g_r'V^-1 g_r is below 5 are "nulls"; gamma = mean
    ratio over the nulls when there are at least 100 of them (stopping once soft_cap markers were looked at), else over all
    valid markers; times the scale.  markers: (n_markers, n) f64.  -> (gamma, markers used)."""
    cx = splmm_xtx_chol(x_design)
    fast_sum = res_sum = 0.0
    n_used = res_used = 0
    for idx, g in enumerate(markers):
        g_sq = float(g @ g)
        g_r = splmm_residualize(x_design, cx, g)
        s_ms = float(g_r @ g_r)
        if splmm_residual_sumsq_is_effectively_zero(s_ms, g_sq):
            continue
        svs = float(g_r @ fac.solve(g_r))
        if not (math.isfinite(svs) and svs > 1e-30):
            continue
        ratio = svs / s_ms
        if not (math.isfinite(ratio) and ratio > 0.0):
            continue
        res_sum += ratio
        res_used += 1
        score = float(g_r @ a_vec)
        chisq = score * score / svs
        if math.isfinite(chisq) and chisq < 5.0:
            fast_sum += ratio
            n_used += 1
        if idx + 1 >= soft_cap and n_used >= 100:
            break
    if n_used == 0 and res_used == 0:
        raise RuntimeError("SparseLMM residualized approx gamma estimation found no valid sampled markers")
    if n_used >= 100:
        return fast_sum / n_used * gamma_scale, n_used
    return res_sum / res_used * gamma_scale, res_used


def splmm_residual_sumsq_is_effectively_zero(resid, raw):
    """src/stats/splmm.rs:1710-1717."""
    if not (math.isfinite(resid) and math.isfinite(raw)):
        return True
    return resid <= max(1e-10, 1e-12 * max(abs(raw), 1.0))


def splmm_grammar_scan(packed, n_samples, maf, row_flip, x_design, score_vec, r_hat, sample_idx=None, rows=None,
                       score_scale=1.0, wald_sigma2=1.0, exact_dots=False, model="add"):
    """`grammar_scan_blocks_core`, additive model (src/stats/splmm.rs:2935-3316) as `scan_with_py_and_rhat` calls it (:3318):
    mean-imputed additive f32 decode (not centred), [score | X'g] = f32 GEMM of the block with f32([score_vec | X]), row sum
    of squares from the counts (`additive_row_sumsq_from_counts`, :1801-1820), g'M g = max(g'g - (X'g)'(X'X)^-1 (X'g), 0),
    denominator r_hat g'M g, `splmm_wald_from_score_denom` with sigma2 = 1; rows whose residual sum of squares is
    effectively zero and failed rows are (NaN, NaN, 1).  -> (m, 3) f64.
    exact_dots: accumulate the products of the f32 operands in f64 and round the sum to f32 once (what a correctly rounded
    sgemm would return); the default accumulates in f32 like a BLAS sgemm, whose summation order is its own -- at n = 200 000
    that order alone moves beta by 1e-5 of its standard error."""
    n, p = x_design.shape
    cx = splmm_xtx_chol(x_design)
    rhs32 = np.concatenate([np.asarray(score_vec, dtype=np.float64)[:, None], x_design], axis=1).astype(np.float32)
    codes = unpack_codes(np.ascontiguousarray(packed, dtype=np.uint8), n_samples)
    if sample_idx is not None:
        codes = codes[:, np.asarray(sample_idx, dtype=np.int64)]
    rows = np.arange(codes.shape[0]) if rows is None else np.asarray(rows, dtype=np.int64)
    out = np.empty((len(rows), 3), dtype=np.float64)
    if str(model).lower() != "add":
        # the non-additive branch (src/stats/splmm.rs:3211-3262): the row decoded in f64 with the model applied, score, X'g and
        # g'g as f64 sums over the samples, g'M g = max(g'g - (X'g)'(X'X)^-1 (X'g), 0), NO effectively-zero test -- a row the
        # model makes constant fails inside `splmm_wald_from_score_denom` (denominator 0) -> (NaN, NaN, 1)
        sv = np.asarray(score_vec, dtype=np.float64)
        for k, r in enumerate(rows):
            g = splmm_additive_row_f64(codes[r], maf[r], bool(row_flip[r]), model)
            xts = x_design.T @ g
            s_m_s = max(float(g @ g) - float(xts @ cholesky_solve(cx, xts)), 0.0)
            res = splmm_wald_from_score_denom(score_scale * float(g @ sv), r_hat * s_m_s, wald_sigma2)
            out[k] = res if res is not None else (float("nan"), float("nan"), 1.0)
        return out
    for k, r in enumerate(rows):
        mean32 = F32(min(max(F32(2.0) * F32(maf[r]), F32(0.0)), F32(2.0)))
        lut = np.array([2.0, mean32, 1.0, 0.0] if row_flip[r] else [0.0, mean32, 1.0, 2.0], dtype=np.float32)
        c = codes[r]
        g32 = lut[c]
        if exact_dots:
            dots = (g32.astype(np.float64) @ rhs32.astype(np.float64)).astype(np.float32)
        else:
            dots = (g32 @ rhs32).astype(np.float32)           # sgemm output, f32
        missing, het, hom_alt = int(np.sum(c == 1)), int(np.sum(c == 2)), int(np.sum(c == 3))
        hom_two = (n - missing - het - hom_alt) if row_flip[r] else hom_alt
        mean_g = min(max(2.0 * float(F32(maf[r])), 0.0), 2.0)
        s_sq = 4.0 * hom_two + (het + missing * mean_g * mean_g)
        xts = dots[1:].astype(np.float64)
        s_m_s = max(s_sq - float(xts @ cholesky_solve(cx, xts)), 0.0)
        res = None
        if not splmm_residual_sumsq_is_effectively_zero(s_m_s, s_sq):
            res = splmm_wald_from_score_denom(score_scale * float(dots[0]), r_hat * s_m_s, wald_sigma2)
        out[k] = res if res is not None else (float("nan"), float("nan"), 1.0)
    return out


def splmm_approx_assoc(k, lam, x_design, y, packed, n_samples, maf, row_flip, rhat_markers=30, rhat_seed=20260527,
                       sample_idx=None, rhat_rows=None, model="add"):
    """`estimate_residualized_approx_scan_sparse` (src/stats/splmm_approx.rs:701-795) = the `-splmm` default of the
    reference's workflow (scan_mode "approx"): null from lambda, gamma from sampled markers, scan model a_r = M_X a,
    GRAMMAR scan.  `rhat_rows` overrides the seeded choice (see StdRngU32).  -> (gamma, (m, 3) f64, markers used, rows)."""
    fac, _y_resid, a_vec, sigma2 = splmm_approx_null(k, x_design, y, lam)
    m = np.asarray(packed).shape[0]
    rr = choose_rhat_rows(m, rhat_markers, rhat_seed) if rhat_rows is None else np.asarray(rhat_rows, dtype=np.int64)
    codes = unpack_codes(np.ascontiguousarray(np.asarray(packed)[rr], dtype=np.uint8), n_samples)
    if sample_idx is not None:
        codes = codes[:, np.asarray(sample_idx, dtype=np.int64)]
    markers = np.stack([splmm_additive_row_f64(codes[i], maf[r], bool(row_flip[r]), model) for i, r in enumerate(rr)])
    gamma, n_used = splmm_estimate_gamma(fac, x_design, markers, a_vec, rhat_markers, 1.0 / sigma2)
    if not (math.isfinite(gamma) and gamma > 0.0):
        raise RuntimeError(f"SparseLMM residualized approx gamma must be finite and > 0, got {gamma}")
    a_resid = splmm_residualize(x_design, splmm_xtx_chol(x_design), a_vec)
    out = splmm_grammar_scan(packed, n_samples, maf, row_flip, x_design, a_resid, gamma, sample_idx, model=model)
    return gamma, out, n_used, rr


def grm_stream_bed(packed, n_samples, method=1, maf_threshold=0.02, max_missing_rate=0.05,
                   het_threshold=0.0, block_rows=65536, exact_syrk=False):
    """`grm_stream_bed_f32` restatement (src/stats/grm.rs:4690-5455): prestat pass -> keep/flip/mean/
    scale -> f32 block SYRK + f64 merge -> scale by sum(var) (method 1) or eff_m (method 2).
    Returns (K f32, eff_m, keep mask).
    exact_syrk: the block product of the same f32 operands in exact (f64) arithmetic instead of an f32 SSYRK -- what the
    reference's `cblas_ssyrk` (src/stats/grm.rs:1638-1667) would return without the rounding of its own f32 accumulation,
    whose size and sign depend on the BLAS at hand (blocking, threads, FMA); everything else unchanged, the result is cast to
    f32 as the reference's API does."""
    maf_thr = min(max(float(maf_threshold), 0.0), 0.5)
    miss_thr = min(max(float(max_missing_rate), 0.0), 1.0)
    het_thr = min(max(float(het_threshold), 0.0), 1.0)
    missing, het, hom = row_counts(packed, n_samples)
    keep, mean_g, scale, flip, var = stream_grm_row_prepare(
        missing, het, hom, n_samples, method, maf_thr, miss_thr, het_thr)
    idx = np.nonzero(keep)[0]
    eff_m = len(idx)
    if eff_m == 0:
        raise RuntimeError("No SNPs remained after filtering; GRM is empty.")
    acc = np.zeros((n_samples, n_samples), dtype=np.float64)
    varsum = 0.0
    step = max(1, int(block_rows))
    for b0 in range(0, eff_m, step):
        rows = idx[b0:b0 + step]
        codes = unpack_codes(packed[rows], n_samples)
        z = np.empty(codes.shape, dtype=np.float32)
        for k, j in enumerate(rows):
            g = (F32(2.0), F32(1.0), F32(0.0)) if flip[j] else (F32(0.0), F32(1.0), F32(2.0))
            lut = np.array([(g[0] - mean_g[j]) * scale[j], F32(0.0), (g[1] - mean_g[j]) * scale[j],
                            (g[2] - mean_g[j]) * scale[j]], dtype=np.float32)
            z[k] = lut[codes[k]]
            varsum += var[j]
        if exact_syrk:
            z64 = z.astype(np.float64)
            acc += z64.T @ z64
        else:
            acc += (z.T @ z).astype(np.float64)
    D = varsum if method == 1 else float(eff_m)
    if not (math.isfinite(D) and D > 0.0):
        raise RuntimeError("invalid centered GRM denominator: sum(2p(1-p)) <= 0")
    k = acc * (1.0 / D)
    k = np.tril(k) + np.tril(k, -1).T
    return k.astype(np.float32), eff_m, keep


# --------------------------------------------------------------------------------------------
# B  eigendecomposition (src/math/eigh.rs:179-207, 1422-1470; python/janusx/assoc/workflow.py:5639)
# --------------------------------------------------------------------------------------------

def eigh_sym(a: np.ndarray, driver="evd"):
    """symmetrise ((A+A^T)/2) then LAPACK dsyevd/dsyevr, ascending eigenvalues, columns = vectors."""
    import scipy.linalg as sla
    a = np.asarray(a, dtype=np.float64)
    a = 0.5 * (a + a.T)
    s, u = sla.eigh(a, driver=driver)
    return s, u


def gwas_eigh_from_grm(k: np.ndarray, diag_ridge=1e-6, subset_idx=None):
    k = np.asarray(k, dtype=np.float64)
    if subset_idx is not None:
        ix = np.asarray(subset_idx, dtype=np.int64)
        k = k[np.ix_(ix, ix)]
    k = k.copy()
    k[np.diag_indices_from(k)] += diag_ridge
    return eigh_sym(k)


# --------------------------------------------------------------------------------------------
# C  rotation of X, y and likelihoods
# --------------------------------------------------------------------------------------------

def lmm_rotate_x_y_with_ut(u_t_f32, x, y):
    """X~ = U^T X, y~ = U^T y with U^T stored f32, f64 accumulation (src/stats/reml.rs:157-170)."""
    ut = np.asarray(u_t_f32, dtype=np.float32).astype(np.float64)
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).ravel()
    return ut @ x, (ut @ y).reshape(-1, 1)


def cholesky_inplace(a: np.ndarray) -> bool:
    """src/math/linalg.rs:341-363 (lower factor, pivot floor 1e-18, upper zeroed)."""
    dim = a.shape[0]
    for i in range(dim):
        for j in range(i + 1):
            s = a[i, j]
            for k in range(j):
                s -= a[i, k] * a[j, k]
            if i == j:
                if s <= 1e-18:
                    return False
                a[i, j] = math.sqrt(s)
            else:
                a[i, j] = s / a[j, j]
        for j in range(i + 1, dim):
            a[i, j] = 0.0
    return True


def cholesky_solve(l: np.ndarray, b: np.ndarray) -> np.ndarray:
    """src/stats/reml.rs:46-66."""
    dim = l.shape[0]
    yv = np.zeros(dim)
    for i in range(dim):
        s = b[i]
        for k in range(i):
            s -= l[i, k] * yv[k]
        yv[i] = s / l[i, i]
    x = np.zeros(dim)
    for i in range(dim - 1, -1, -1):
        s = yv[i]
        for k in range(i + 1, dim):
            s -= l[k, i] * x[k]
        x[i] = s / l[i, i]
    return x


def _moments(lbd, s, xcov, y, snp):
    """A (lower, before ridge), b, and the design matrix used by reml/ml/final_beta_se
    (src/stats/reml.rs:286-319). numpy reductions (pairwise) instead of the reference's sequential
    loop: differences are O(1e-16) relative; the C oracle (jx_oracle.c) keeps the sequential order."""
    v = s + lbd
    if np.any(v <= 0.0):
        return None
    vinv = 1.0 / v
    x = xcov if snp is None else np.concatenate([xcov, snp[:, None]], axis=1)
    xw = x * vinv[:, None]
    a = xw.T @ x
    b = xw.T @ y
    return v, vinv, x, a, b


def reml_loglike(log10_lbd, s, xcov, y, snp=None):
    """src/stats/reml.rs:255-362."""
    lbd = 10.0 ** log10_lbd
    if not math.isfinite(lbd) or lbd <= 0.0:
        return -1e8
    n, p_cov = xcov.shape
    dim = p_cov + (0 if snp is None else 1)
    if n <= dim:
        return -1e8
    mo = _moments(lbd, s, xcov, y, snp)
    if mo is None:
        return -1e8
    v, vinv, x, a, b = mo
    a = np.tril(a)
    a[np.diag_indices(dim)] += 1e-6
    a = a + np.tril(a, -1).T
    if not cholesky_inplace(a):
        return -1e8
    beta = cholesky_solve(a, b)
    r = y - x @ beta
    q = float(np.sum(vinv * r * r))
    log_det_v = float(np.sum(np.log(v)))
    log_det_xtv = 2.0 * float(np.sum(np.log(np.diag(a))))
    with np.errstate(all="ignore"):
        total_log = (n - dim) * (math.log(q) if q > 0 else float("nan")) + log_det_v + log_det_xtv
    c = (n - dim) * (math.log(n - dim) - 1.0 - math.log(2.0 * math.pi)) / 2.0
    reml = c - 0.5 * total_log
    return reml if math.isfinite(reml) else -1e8


def ml_loglike(log10_lbd, s, xcov, y, snp=None):
    """src/stats/reml.rs:364-470."""
    lbd = 10.0 ** log10_lbd
    if not math.isfinite(lbd) or lbd <= 0.0:
        return -1e8
    n, p_cov = xcov.shape
    dim = p_cov + (0 if snp is None else 1)
    if n <= dim:
        return -1e8
    mo = _moments(lbd, s, xcov, y, snp)
    if mo is None:
        return -1e8
    v, vinv, x, a, b = mo
    a = np.tril(a)
    a[np.diag_indices(dim)] += 1e-6
    a = a + np.tril(a, -1).T
    if not cholesky_inplace(a):
        return -1e8
    beta = cholesky_solve(a, b)
    r = y - x @ beta
    q = float(np.sum(vinv * r * r))
    if not math.isfinite(q) or q <= 0.0:
        return -1e8
    total_log = n * math.log(q) + float(np.sum(np.log(v)))
    c = n * (math.log(n) - 1.0 - math.log(2.0 * math.pi)) / 2.0
    ml = c - 0.5 * total_log
    return ml if math.isfinite(ml) else -1e8


def final_beta_se(log10_lbd, s, xcov, y, snp):
    """src/stats/reml.rs:472-568 -> (beta_k, se_k, lbd)."""
    lbd = 10.0 ** log10_lbd
    nan = float("nan")
    if not math.isfinite(lbd) or lbd <= 0.0:
        return nan, nan, nan
    n, p_cov = xcov.shape
    dim = p_cov + 1
    if n <= dim:
        return nan, nan, lbd
    mo = _moments(lbd, s, xcov, y, snp)
    if mo is None:
        return nan, nan, lbd
    v, vinv, x, a, b = mo
    a = np.tril(a)
    a[np.diag_indices(dim)] += 1e-6
    a = a + np.tril(a, -1).T
    if not cholesky_inplace(a):
        return nan, nan, lbd
    beta = cholesky_solve(a, b)
    r = y - x @ beta
    q = float(np.sum(vinv * r * r))
    sigma2 = q / (n - dim)
    e = np.zeros(dim)
    e[dim - 1] = 1.0
    xk = cholesky_solve(a, e)
    var_k = sigma2 * xk[dim - 1]
    if var_k <= 0.0 or not math.isfinite(var_k):
        return nan, nan, lbd
    return float(beta[dim - 1]), math.sqrt(var_k), lbd


def brent_minimize(f, low, high, tol, max_iter, init_x=None):
    """src/math/brent.rs:1-136 verbatim semantics (incl. `e` not updated on parabolic steps).
    Returns (x, fx, n_evals)."""
    a, c = low, high
    if not (a < c):
        a, c = c, a
    eps = EPS64
    tol = max(abs(tol), 1e-12)
    if init_x is not None and math.isfinite(init_x) and a <= init_x <= c:
        x = init_x
    else:
        x = 0.5 * (a + c)
    w = v = x
    fx = f(x)
    fw = fv = fx
    d = 0.0
    e = 0.0
    evals = 1
    for _ in range(int(max_iter)):
        m = 0.5 * (a + c)
        tol1 = tol * abs(x) + eps
        tol2 = 2.0 * tol1
        if abs(x - m) <= tol2 - 0.5 * (c - a):
            break
        use_parabolic = False
        if abs(e) > tol1:
            p = (x - v) * ((x - w) * (fx - fv)) - (x - w) * ((x - v) * (fx - fw))
            q = 2.0 * (((x - v) * (fx - fw)) - ((x - w) * (fx - fv)))
            if q > 0.0:
                p = -p
            else:
                q = -q
            ok = False
            if abs(q) > eps:
                sstep = p / q
                u = x + sstep
                if (u - a) >= tol2 and (c - u) >= tol2 and abs(sstep) < 0.5 * abs(e):
                    ok = True
            if ok:
                d = p / q
                u = x + d
                if (u - a) < tol2 or (c - u) < tol2:
                    d = tol1 if x < m else -tol1
                use_parabolic = True
        if not use_parabolic:
            e = (c - x) if x < m else (a - x)
            d = 0.3819660 * e
        if abs(d) < tol1:
            d = tol1 if d >= 0.0 else -tol1
        u = x + d
        fu = f(u)
        evals += 1
        if fu <= fx:
            if u >= x:
                a = x
            else:
                c = x
            v, fv = w, fw
            w, fw = x, fx
            x, fx = u, fu
        else:
            if u >= x:
                c = u
            else:
                a = u
            if fu <= fw or w == x:
                v, fv = w, fw
                w, fw = u, fu
            elif fu <= fv or v == x or v == w:
                v, fv = u, fu
    return x, fx, evals


def lmm_reml_null(s, xcov, y_rot, low, high, max_iter=50, tol=1e-2):
    """`lmm_reml_null_f32` (src/stats/reml.rs:572-616) -> (lbd, ml, reml)."""
    s = np.asarray(s, dtype=np.float64).ravel()
    xcov = np.asarray(xcov, dtype=np.float64)
    y = np.asarray(y_rot, dtype=np.float64).ravel()
    if low >= high:
        raise RuntimeError("low must be < high")
    x, cost, _ = brent_minimize(lambda t: -reml_loglike(t, s, xcov, y, None), low, high, tol, max_iter)
    ml = ml_loglike(x, s, xcov, y, None)
    return 10.0 ** x, ml, -cost


def normal_sf(z):
    """src/math/linalg.rs:2-5."""
    return 0.5 * math.erfc(z / math.sqrt(2.0))


def chi2_sf_df1(stat):
    """src/math/linalg.rs:7-17."""
    if not math.isfinite(stat) or stat <= 0.0:
        return 1.0
    p = math.erfc(math.sqrt(0.5 * stat))
    return min(max(p, MIN_POSITIVE), 1.0) if math.isfinite(p) else 1.0


# --------------------------------------------------------------------------------------------
# A.4  spectral null model (python/janusx/pyBLUP/assoc.py:1726-1876)
# --------------------------------------------------------------------------------------------

@dataclass
class NullModel:
    S: np.ndarray
    Dh: np.ndarray  # f32 U^T
    Xcov: np.ndarray
    y: np.ndarray
    lbd_null: float
    ML0: float
    LL0: float
    sigma_g2: float
    sigma_e2: float
    pve: float
    bounds: tuple
    trace_mean: float


def lmm_profile_exact_vc(S, Xcov, y_rot, lbd):
    """python/janusx/pyBLUP/assoc.py:907-951."""
    s = np.maximum(np.asarray(S, dtype=np.float64).ravel(), 0.0)
    x = np.asarray(Xcov, dtype=np.float64)
    y = np.asarray(y_rot, dtype=np.float64).ravel()
    n, p = x.shape
    if not math.isfinite(lbd) or lbd <= 0.0 or n - p <= 0:
        return float("nan"), float("nan")
    v_inv = 1.0 / np.maximum(s + lbd, 1e-30)
    a = (x.T * v_inv) @ x
    b = (x.T * v_inv) @ y
    beta = np.linalg.solve(a, b)
    r = y - x @ beta
    q = float(np.dot(v_inv, r * r))
    if not math.isfinite(q) or q <= 0.0:
        return float("nan"), float("nan")
    sg2 = q / float(n - p)
    return sg2, lbd * sg2


def spectral_null_model(y, X, S, U) -> NullModel:
    """`LMM._initialize_from_spectral` full-rank branch (assoc.py:1816-1876). X includes the
    intercept column already (assoc.py:1710-1714)."""
    S = np.ascontiguousarray(S, dtype=np.float64).ravel()
    U = np.asarray(U, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).ravel()
    X = np.asarray(X, dtype=np.float64)
    n = y.shape[0]
    trace_mean = float(np.sum(np.clip(S, 0.0, None)) / float(max(1, n)))
    Dh = np.ascontiguousarray(U.T.astype(np.float32))
    xr, yr = lmm_rotate_x_y_with_ut(Dh, X, y)
    lbd, ml0, reml = lmm_reml_null(S, xr, yr, -5.0, 5.0, 50, 1e-3)
    sg2, se2 = lmm_profile_exact_vc(S, xr, yr, lbd)
    ssum = sg2 + se2
    if math.isfinite(ssum) and ssum > 0.0:
        vg = sg2 * max(trace_mean, 0.0)
        den = vg + se2
        pve = vg / den if (math.isfinite(den) and den > 0.0) else sg2 / ssum
    else:
        vg_null = float(np.mean(np.clip(S, 0.0, None)))
        pve = vg_null / (vg_null + lbd) if (vg_null + lbd) > 0 else float("nan")
    if pve > 0.95 or pve < 0.05 or (not math.isfinite(lbd)) or lbd <= 0.0:
        bounds = (-5.0, 5.0)
    else:
        bounds = (math.log10(lbd) - 2.0, math.log10(lbd) + 2.0)
    return NullModel(S, Dh, xr, yr.ravel(), lbd, ml0, reml, sg2, se2, pve, bounds, trace_mean)


# --------------------------------------------------------------------------------------------
# D  scan design decode + rotation + exact per-SNP scan
# --------------------------------------------------------------------------------------------

def genetic_model_apply(model: str, g: float) -> float:
    """`PackedGeneticModel::parse` + `apply` (src/decode/decode.rs:107-160): add g; dom 1 if g > 0; rec 1 if |g - 2| < 1e-6;
    het 1 if |g - 1| < 1e-6; else 0."""
    m = str(model).lower()
    if m == "add":
        return g
    if m == "dom":
        return 1.0 if g > 0.0 else 0.0
    if m == "rec":
        return 1.0 if abs(g - 2.0) < 1e-6 else 0.0
    if m == "het":
        return 1.0 if abs(g - 1.0) < 1e-6 else 0.0
    raise RuntimeError("model must be one of: add, dom, rec, het")


def scan_value_lut_f32(row_maf_f32, flip: bool, model: str = "add"):
    """The genetic model applied to [0, mu, 1, 2] (or flipped) with mu = f32(max(2*f64(maf), 0)) -- the imputed entry
    included (`packed_model_value_lut_f32`, src/decode/decode.rs:163-178, 218)."""
    mu = F32(max(2.0 * float(F32(row_maf_f32)), 0.0))
    raw = [2.0, float(mu), 1.0, 0.0] if flip else [0.0, float(mu), 1.0, 2.0]
    return np.array([F32(genetic_model_apply(model, float(F32(v)))) for v in raw], dtype=np.float32)


def decode_centered_block_f32(packed, n_samples, row_flip, row_maf, sample_idx=None, rows=None, model: str = "add"):
    """`decode_centered_block_packed_f32` (src/decode/decode.rs:192-271): mean-impute from the passed
    maf, apply the genetic model to the table, then subtract the *actual* row mean (f64 sum -> f32, decode.rs:181-189)."""
    if rows is None:
        rows = np.arange(packed.shape[0])
    codes = unpack_codes(packed[rows], n_samples)
    if sample_idx is not None:
        codes = codes[:, np.asarray(sample_idx, dtype=np.int64)]
    out = np.empty(codes.shape, dtype=np.float32)
    for k, j in enumerate(rows):
        lut = scan_value_lut_f32(row_maf[j], bool(row_flip[j]), model)
        g = lut[codes[k]]
        mean = F32(np.sum(g.astype(np.float64)) / float(g.shape[0]))
        out[k] = g - mean
    return out


def rotate_block_f32(g_block, u_t):
    """out[r,j] = sum_i g[r,i]*u_t[j,i], f32 GEMM (src/stats/lmm.rs:520-552, 728-784)."""
    return np.asarray(g_block, dtype=np.float32) @ np.asarray(u_t, dtype=np.float32).T


def lmm_scan_rotated_block(g_rot, s, xcov, y, low, high, max_iter, tol, nullml=None, init=None,
                           count_evals=False):
    """`run_rotated_reml_assoc_block_f32` without warm start (src/stats/lmm.rs:94-199)."""
    g_rot = np.asarray(g_rot, dtype=np.float32)
    rows = g_rot.shape[0]
    ncol = 4 if nullml is not None else 3
    out = np.zeros((rows, ncol), dtype=np.float64)
    evals = np.zeros(rows, dtype=np.int64)
    for r in range(rows):
        snp = g_rot[r].astype(np.float64)
        ssq = float(np.sum(snp * snp))
        if not math.isfinite(ssq) or ssq <= 1e-12:
            out[r, 0] = out[r, 1] = np.nan
            out[r, 2] = 1.0
            if nullml is not None:
                out[r, 3] = 1.0
            continue
        x, _, ne = brent_minimize(lambda t: -reml_loglike(t, s, xcov, y, snp), low, high, tol, max_iter,
                                  init)
        evals[r] = ne
        beta, se, _ = final_beta_se(x, s, xcov, y, snp)
        if math.isfinite(beta) and math.isfinite(se) and se > 0.0:
            z = beta / se
            p = min(max(2.0 * normal_sf(abs(z)), MIN_POSITIVE), 1.0)
            out[r, 0], out[r, 1], out[r, 2] = beta, se, (p if math.isfinite(p) else 1.0)
            if nullml is not None:
                ml = ml_loglike(x, s, xcov, y, snp)
                if math.isfinite(ml):
                    stat = 2.0 * (ml - nullml)
                    if not math.isfinite(stat) or stat < 0.0:
                        stat = 0.0
                    out[r, 3] = chi2_sf_df1(stat)
                else:
                    out[r, 3] = 1.0
        else:
            out[r, 0] = out[r, 1] = np.nan
            out[r, 2] = 1.0
            if nullml is not None:
                out[r, 3] = 1.0
    return (out, evals) if count_evals else out


def lmm2_scan_rotated_block(g_rot, s, xcov, y, low, high, max_iter, tol, nullml, init_reml=None, init_ml=None):
    """`run_rotated_lmm2_assoc_block_f32` with use_warm_start = false (src/stats/lmm.rs:202-330):
    (rows, 6) = [beta, se, pwald, lambda_reml, ml_alt, plrt]; invalid rows (NaN, NaN, 1, NaN, NaN, 1)."""
    g_rot = np.asarray(g_rot, dtype=np.float32)
    rows = g_rot.shape[0]
    out = np.zeros((rows, 6), dtype=np.float64)
    init = init_reml if init_reml is not None else init_ml
    for r in range(rows):
        out[r] = (np.nan, np.nan, 1.0, np.nan, np.nan, 1.0)
        snp = g_rot[r].astype(np.float64)
        ssq = float(np.sum(snp * snp))
        if not math.isfinite(ssq) or ssq <= 1e-12:
            continue
        xr, _, _ = brent_minimize(lambda t: -reml_loglike(t, s, xcov, y, snp), low, high, tol, max_iter, init)
        beta, se, lbd = final_beta_se(xr, s, xcov, y, snp)
        if not (math.isfinite(beta) and math.isfinite(se) and se > 0.0):
            continue
        pw = min(max(2.0 * normal_sf(abs(beta / se)), MIN_POSITIVE), 1.0)
        xm, fm, _ = brent_minimize(lambda t: -ml_loglike(t, s, xcov, y, snp), low, high, tol, max_iter, xr)
        ml_alt = -fm
        if not math.isfinite(ml_alt):
            ml_alt = ml_loglike(xm, s, xcov, y, snp)
        stat = 2.0 * (ml_alt - nullml) if math.isfinite(ml_alt) else 0.0
        if not math.isfinite(stat) or stat < 0.0:
            stat = 0.0
        plrt = chi2_sf_df1(stat)
        out[r] = (beta, se, pw if math.isfinite(pw) else 1.0, lbd, ml_alt, plrt if math.isfinite(plrt) else 1.0)
    return out


def lmm2_null_ml(s, xcov, y, low, high, max_iter, tol, init=None):
    """Null ML of the LMM2 BED route (src/stats/lmm.rs:2902-2921) -> (log10 lambda, ml0)."""
    xm, fm, _ = brent_minimize(lambda t: -ml_loglike(t, s, xcov, y), low, high, tol, max_iter, init)
    ml0 = -fm
    if not math.isfinite(ml0):
        ml0 = ml_loglike(xm, s, xcov, y)
    return xm, ml0


def lmm_reml_chunk_from_snp(s, xcov, y_rot, low, high, snp_chunk, u_t, max_iter=50, tol=1e-2,
                            nullml=None):
    """`lmm_reml_chunk_from_snp_f32` (src/stats/lmm.rs:1479-1630): rotate then exact scan, no warm start."""
    g_rot = rotate_block_f32(snp_chunk, u_t)
    return lmm_scan_rotated_block(g_rot, np.asarray(s, dtype=np.float64), np.asarray(xcov, dtype=np.float64),
                                  np.asarray(y_rot, dtype=np.float64).ravel(), low, high, max_iter, tol, nullml)


# --------------------------------------------------------------------------------------------
# E  fixed-lambda scan (src/stats/fvlmm.rs:1484-1563, 1691-1805)
# --------------------------------------------------------------------------------------------

@dataclass
class FvCache:
    w: np.ndarray        # f32 (n)
    py: np.ndarray       # f32 (n)
    wx: np.ndarray       # f32 (n,p)
    a_chol: np.ndarray   # f64 (p,p) lower
    ypy: float
    log_det_v: float
    df: int


def fvlmm_prepare_cache(s, xcov, y, lbd) -> FvCache:
    s = np.asarray(s, dtype=np.float64).ravel()
    xcov = np.asarray(xcov, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).ravel()
    n, p = xcov.shape
    vv = s + lbd
    if not np.all(np.isfinite(vv) & (vv > 0.0)):
        raise RuntimeError("non-positive s[i]+lbd")
    w = (1.0 / vv).astype(np.float32)
    log_det_v = float(np.sum(np.log(vv)))
    w64 = w.astype(np.float64)
    a = (xcov * w64[:, None]).T @ xcov
    b = (xcov * w64[:, None]).T @ y
    ywy = float(np.sum(w64 * y * y))
    a = np.tril(a)
    a[np.diag_indices(p)] += 1e-6
    a = a + np.tril(a, -1).T
    if not cholesky_inplace(a):
        raise RuntimeError("X'WX not SPD")
    aib = cholesky_solve(a, b)
    ypy = max(ywy - float(np.dot(b, aib)), 0.0)
    wx = (w64[:, None] * xcov).astype(np.float32)
    py = (w64 * (y - xcov @ aib)).astype(np.float32)
    df = n - p - 1
    if df <= 0:
        raise RuntimeError("df <= 0")
    return FvCache(w, py, wx, a, ypy, log_det_v, df)


def fvlmm_assoc_rotated_block(g_rot, cache: FvCache, nullml=None):
    g_rot = np.asarray(g_rot, dtype=np.float32)
    rows, n = g_rot.shape
    p = cache.wx.shape[1]
    num = g_rot @ cache.py               # f32 GEMM
    cbuf = g_rot @ cache.wx              # f32 GEMM
    ncol = 4 if nullml is not None else 3
    out = np.zeros((rows, ncol), dtype=np.float64)
    w64 = cache.w.astype(np.float64)
    n_f = float(n)
    c_ml = n_f * (math.log(n_f) - 1.0 - math.log(2.0 * math.pi)) / 2.0
    for r in range(rows):
        g = g_rot[r].astype(np.float64)
        d = float(np.sum(w64 * g * g))
        c = cbuf[r].astype(np.float64)
        aic = cholesky_solve(cache.a_chol, c)
        schur = d - float(np.dot(c, aic))
        if schur <= 1e-12 or not math.isfinite(schur):
            out[r, :3] = np.nan
            if nullml is not None:
                out[r, 3] = 1.0
            continue
        nu = float(num[r])
        beta = nu / schur
        rwr = max(cache.ypy - (nu * nu) / schur, 0.0)
        sigma2 = rwr / float(cache.df)
        se = math.sqrt(sigma2 / schur)
        if math.isfinite(se) and se > 0.0 and math.isfinite(beta):
            pval = min(max(2.0 * normal_sf(abs(beta / se)), MIN_POSITIVE), 1.0)
        else:
            pval = 1.0
        out[r, 0], out[r, 1], out[r, 2] = beta, se, pval
        if nullml is not None:
            if rwr > 0.0 and math.isfinite(rwr):
                ml = c_ml - 0.5 * (n_f * math.log(rwr) + cache.log_det_v)
            else:
                ml = float("nan")
            stat = 2.0 * (ml - nullml) if math.isfinite(ml) else 0.0
            if not math.isfinite(stat) or stat < 0.0:
                stat = 0.0
            out[r, 3] = chi2_sf_df1(stat)
    return out


def fvlmm_assoc_chunk_from_snp(s, xcov, y_rot, log10_lbd, snp_chunk, u_t, nullml=None):
    """`fvlmm_assoc_chunk_from_snp_f32` (src/stats/fvlmm.rs:2114-2262)."""
    cache = fvlmm_prepare_cache(s, xcov, y_rot, 10.0 ** log10_lbd)
    return fvlmm_assoc_rotated_block(rotate_block_f32(snp_chunk, u_t), cache, nullml)


# --------------------------------------------------------------------------------------------
# G2  LMM -> LM fallback decision (src/stats/gwas_unified.rs:54-175)
# --------------------------------------------------------------------------------------------

def lmm_assoc_fixed_lambda_block(g_rot, s, xcov, y, log10_lbd, nullml=None):
    """`lmm_assoc_chunk_f32` (src/stats/lmm.rs:2010-2224): Wald statistics of rotated rows at ONE lambda.  W = 1 / (s + lambda)
    held in f32, c = X'Wg, d = g'Wg, e = g'Wy summed in f64, A = X'WX + 1e-6 I (Cholesky), Schur complement d - c'A^-1 c
    (<= 1e-12 -> (NaN, NaN, NaN), the plrt entry keeps the 0.0 of the allocation), beta = (e - c'A^-1 b) / schur,
    r'Wr = max(y'Wy - b'A^-1 b - num^2 / schur, 0), sigma2 = r'Wr / (n - p - 1).  -> (m, 3 or 4)."""
    s = np.asarray(s, dtype=np.float64)
    x = np.asarray(xcov, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    g_rot = np.asarray(g_rot, dtype=np.float32)
    n, p = x.shape
    lbd = 10.0 ** float(log10_lbd)
    vv = s + lbd
    w = (1.0 / vv).astype(np.float32).astype(np.float64)
    log_det_v = float(np.sum(np.log(vv)))
    nf = float(n)
    c_ml = nf * (math.log(nf) - 1.0 - math.log(2.0 * math.pi)) / 2.0
    a = (x * w[:, None]).T @ x + 1e-6 * np.eye(p)
    b = x.T @ (w * y)
    ywy = float(np.sum(w * y * y))
    la = a.copy()
    if not cholesky_inplace(la):
        raise RuntimeError("X'WX not SPD")
    a_inv_b = cholesky_solve(la, b)
    b_aib = float(b @ a_inv_b)
    df = n - p - 1
    m = g_rot.shape[0]
    out = np.zeros((m, 4 if nullml is not None else 3), dtype=np.float64)
    for j in range(m):
        g = g_rot[j].astype(np.float64)
        wg = w * g
        d = float(wg @ g)
        e = float(wg @ y)
        c = x.T @ wg
        schur = d - float(c @ cholesky_solve(la, c))
        if schur <= 1e-12 or not math.isfinite(schur):
            out[j, :3] = np.nan
            continue
        num = e - float(c @ a_inv_b)
        beta = num / schur
        rwr = max(ywy - (b_aib + num * num / schur), 0.0)
        se = math.sqrt(rwr / float(df) / schur)
        pv = 1.0
        if math.isfinite(se) and se > 0.0 and math.isfinite(beta):
            pv = min(max(2.0 * normal_sf(abs(beta / se)), 2.2250738585072014e-308), 1.0)
        out[j, :3] = (beta, se, pv)
        if nullml is not None:
            ml = c_ml - 0.5 * (nf * math.log(rwr) + log_det_v) if (rwr > 0.0 and math.isfinite(rwr)) else float("nan")
            stat = 2.0 * (ml - nullml) if math.isfinite(ml) else 0.0
            if not math.isfinite(stat) or stat < 0.0:
                stat = 0.0
            out[j, 3] = chi2_sf_df1(stat)
    return out


def lm_null_ml(y, xcov):
    y = np.asarray(y, dtype=np.float64).ravel()
    x = np.asarray(xcov, dtype=np.float64)
    n = y.shape[0]
    beta, *_ = np.linalg.lstsq(x, y, rcond=None)
    r = y - x @ beta
    rss = float(np.dot(r, r))
    if not (math.isfinite(rss) and rss > 0.0):
        return float("nan")
    n_f = float(n)
    return n_f * (math.log(n_f) - 1.0 - math.log(2.0 * math.pi)) / 2.0 - 0.5 * n_f * math.log(rss)


# --------------------------------------------------------------------------------------------
# A.9b  plain LM scan, the route taken after the fallback decision (src/stats/glm.rs:383-501, 3550-3860)
# --------------------------------------------------------------------------------------------

def lm_precompute_ixx_qr(x):
    """`_lm_precompute_ixx_qr` (python/janusx/pyBLUP/assoc.py:453-480): (X'X)^-1 through the reduced QR, the
    Hermitian pseudo-inverse when X is rank deficient."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    n, q = x.shape
    _q, r = np.linalg.qr(x, mode="reduced")
    diag = np.abs(np.diag(r))
    tol = EPS64 * float(max(n, q)) * (float(diag.max()) if diag.size else 0.0)
    if int(np.sum(diag > tol)) == q:
        rinv = np.linalg.inv(r)
        return np.ascontiguousarray(rinv @ rinv.T)
    return np.ascontiguousarray(np.linalg.pinv(x.T @ x, hermitian=True))


def betacf(a, b, x):
    """Continued fraction of the incomplete beta function (src/stats/glm.rs:383-433)."""
    eps, fpmin = 3.0e-14, 1.0e-300
    qab, qap, qam = a + b, a + 1.0, a - 1.0
    c = 1.0
    d = 1.0 - qab * x / qap
    if abs(d) < fpmin:
        d = fpmin
    d = 1.0 / d
    h = d
    for m in range(1, 201):
        fm = float(m)
        m2 = 2.0 * fm
        aa = fm * (b - fm) * x / ((qam + m2) * (a + m2))
        d = 1.0 + aa * d
        if abs(d) < fpmin:
            d = fpmin
        c = 1.0 + aa / c
        if abs(c) < fpmin:
            c = fpmin
        d = 1.0 / d
        h *= d * c
        aa = -(a + fm) * (qab + fm) * x / ((a + m2) * (qap + m2))
        d = 1.0 + aa * d
        if abs(d) < fpmin:
            d = fpmin
        c = 1.0 + aa / c
        if abs(c) < fpmin:
            c = fpmin
        d = 1.0 / d
        de = d * c
        h *= de
        if abs(de - 1.0) < eps:
            break
    return h


def betai(a, b, x):
    """Regularised incomplete beta function (src/stats/glm.rs:435-455)."""
    if not (0.0 <= x <= 1.0):
        return float("nan")
    if x == 0.0:
        return 0.0
    if x == 1.0:
        return 1.0
    ln_beta = math.lgamma(a) + math.lgamma(b) - math.lgamma(a + b)
    if x < (a + 1.0) / (a + b + 2.0):
        return math.exp(a * math.log(x) + b * math.log(1.0 - x) - ln_beta) / a * betacf(a, b, x)
    return 1.0 - math.exp(b * math.log(1.0 - x) + a * math.log(x) - ln_beta) / b * betacf(b, a, 1.0 - x)


def student_t_p_two_sided(t, df):
    """src/stats/glm.rs:458-481."""
    if df <= 0:
        return float("nan")
    if not math.isfinite(t):
        return float("nan") if math.isnan(t) else MIN_POSITIVE
    v = float(df)
    p = betai(v / 2.0, 0.5, v / (v + t * t))
    if not math.isfinite(p):
        p = 1.0
    return min(max(p, MIN_POSITIVE), 1.0)


def lm_chi2_sf_df1(stat):
    """The LM file's own chi-square tail (src/stats/glm.rs:483-492; NaN for an invalid statistic, unlike linalg.rs:7-17)."""
    if not math.isfinite(stat) or stat < 0.0:
        return float("nan")
    p = math.erfc(math.sqrt(0.5 * stat))
    return min(max(p, MIN_POSITIVE), 1.0) if math.isfinite(p) else 1.0


def lm_plrt_from_t2(t2, n_obs, df):
    """src/stats/glm.rs:495-501."""
    if df <= 0 or not math.isfinite(t2) or t2 < 0.0:
        return float("nan")
    return lm_chi2_sf_df1(float(n_obs) * math.log(1.0 + t2 / float(df)))


def lm_value_lut_f32(row_maf_f32, flip: bool):
    """Mean-imputed additive decode of the LM scan (src/math/bedmath.rs:984-989)."""
    mean_g = F32(min(max(F32(2.0) * F32(row_maf_f32), F32(0.0)), F32(2.0)))
    if flip:
        return np.array([2.0, mean_g, 1.0, 0.0], dtype=np.float32)
    return np.array([0.0, mean_g, 1.0, 2.0], dtype=np.float32)


def lm_block_assoc_packed(y, x, ixx, packed, n_samples, row_flip, row_maf, sample_idx=None):
    """`lm_block_assoc_packed` (src/stats/glm.rs:3550-3860).  x (n, q0) includes the intercept.  The products G X and
    G r_y take f32 operands like the reference's sgemm; they are summed in f64 here (the reference's f32 accumulation
    order is BLAS-dependent).  -> (m, 4) = beta, se, pwald, plrt."""
    y = np.asarray(y, dtype=np.float64).ravel()
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    ixx = np.asarray(ixx, dtype=np.float64)
    n, q0 = x.shape
    if n <= q0 + 1:
        raise RuntimeError(f"n too small: require n > q0+1, got n={n}, q0={q0}")
    df = n - q0 - 1
    c_xy = ixx @ (x.T @ y)
    ry = y - x @ c_xy
    yy_r = float(ry @ ry)
    ry32 = ry.astype(np.float32).astype(np.float64)
    x32 = x.astype(np.float32).astype(np.float64)
    codes = unpack_codes(packed, n_samples)
    if sample_idx is not None:
        codes = codes[:, np.asarray(sample_idx, dtype=np.int64)]
    m = codes.shape[0]
    out = np.full((m, 4), np.nan, dtype=np.float64)
    for j in range(m):
        g = lm_value_lut_f32(row_maf[j], bool(row_flip[j]))[codes[j]].astype(np.float64)
        u = g @ x32
        a = float(g @ ry32)
        d = float(g @ g)
        s = d - float(u @ (ixx @ u))
        if s < 1e-12 or not math.isfinite(s):
            continue
        b = a / s
        rss = max(yy_r - b * a, 0.0)
        ve = rss / float(df)
        if ve <= 0.0:
            out[j, 0] = b
            continue
        se = math.sqrt(ve / s)
        t = b / se
        out[j] = (b, se, student_t_p_two_sided(t, df), lm_plrt_from_t2(t * t, n, df))
    return out


def lm_block_assoc_dense(y, x, ixx, g):
    """`lm_block_assoc_f32` (src/stats/glm.rs:4313-4497): the formulas of `lm_block_assoc_packed` on decoded f32 rows g (m, n);
    u = G X and a = G r_y are f32 sgemm outputs (f32 operands, summed in f64 here, rounded to f32), d = sum g^2 in f64; a row
    needs a finite Schur complement > 1e-12, a finite positive variance (else only beta is kept) and a finite positive se."""
    y = np.asarray(y, dtype=np.float64).ravel()
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    ixx = np.asarray(ixx, dtype=np.float64)
    g = np.asarray(g, dtype=np.float32)
    n, q0 = x.shape
    df = n - q0 - 1
    ry = y - x @ (ixx @ (x.T @ y))
    yy_r = float(ry @ ry)
    ry32 = ry.astype(np.float32).astype(np.float64)
    x32 = x.astype(np.float32).astype(np.float64)
    out = np.full((g.shape[0], 4), np.nan, dtype=np.float64)
    for j in range(g.shape[0]):
        gj = g[j].astype(np.float64)
        u = (gj @ x32).astype(np.float32).astype(np.float64)
        a = float(np.float32(gj @ ry32))
        schur = float(gj @ gj) - float(u @ (ixx @ u))
        if not (math.isfinite(schur) and schur > 1e-12):
            continue
        b = a / schur
        ve = max(yy_r - b * a, 0.0) / float(df)
        if not (math.isfinite(ve) and ve > 0.0):
            out[j, 0] = b
            continue
        se = math.sqrt(ve / schur)
        if not (math.isfinite(b) and math.isfinite(se) and se > 0.0):
            continue
        t = b / se
        out[j] = (b, se, student_t_p_two_sided(t, df), lm_plrt_from_t2(t * t, n, df))
    return out


# --------------------------------------------------------------------------------------------
# A.10  TSV formatting (src/io/assoc2tsv.rs:45-57, 430-548; src/math/linalg.rs:327-340)
# --------------------------------------------------------------------------------------------

TSV_HEADER = "chrom\tpos\tsnp\tallele0\tallele1\taf\tmiss\tbeta\tse\tchisq\tpwald\n"


def rust_fmt_e4(v: float) -> str:
    """Rust `{:.4e}`: mantissa with 4 decimals, exponent without padding or plus sign."""
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    s = f"{v:.4e}"
    mant, ex = s.split("e")
    return f"{mant}e{int(ex)}"


def rust_fmt_f4(v: float) -> str:
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    return f"{v:.4f}"


def format_assoc_row(chrom, pos, snp, a0, a1, af, miss, beta, se, p) -> str:
    name = snp if (snp and snp != ".") else f"{chrom}_{pos}"
    if math.isfinite(beta) and math.isfinite(se) and se > 0.0:
        z = beta / se
        chisq = z * z
        pv = min(max(p, MIN_POSITIVE), 1.0) if math.isfinite(p) else 1.0
    else:
        chisq = float("nan")
        pv = 1.0
    return (f"{chrom}\t{pos}\t{name}\t{a0}\t{a1}\t{rust_fmt_f4(float(af))}\t{rust_fmt_f4(float(miss))}\t"
            f"{rust_fmt_f4(beta)}\t{rust_fmt_f4(se)}\t{rust_fmt_e4(chisq)}\t{rust_fmt_e4(pv)}\n")


# --------------------------------------------------------------------------------------------
# G1  GBLUP (src/stats/gblup.rs:1105-1240 `fit_gblup_reml_from_grm_row_major_f64`, :1258-1516 `gblup_reml_npy_grm`)
# --------------------------------------------------------------------------------------------

def gblup_reml_eval(s, x_rot, y_rot, n, log10_lbd, v_floor=1e-12):
    """One evaluation of the intercept-only spectral likelihood of the GBLUP fit (src/stats/gblup.rs:1140-1190; the
    Python twin is `BLUP._REML`, python/janusx/pyBLUP/mlm.py:1940-2050, which tests/golden/gen_fixtures.py runs to pin
    this function): v_i = max(s_i + lambda, v_floor).  -> (reml, ml, beta, q, 1 / v, r) or None when invalid."""
    lbd = 10.0 ** log10_lbd
    if not (math.isfinite(lbd) and lbd > 0.0):
        return None
    n_eff = float(n - 1)
    c_reml = n_eff * (math.log(n_eff) - 1.0 - math.log(2.0 * math.pi)) / 2.0
    c_ml = n * (math.log(n) - 1.0 - math.log(2.0 * math.pi)) / 2.0
    vi = np.maximum(s + lbd, v_floor)
    log_det_v = float(np.sum(np.log(vi)))
    inv = 1.0 / vi
    xtvx = float(np.sum(inv * x_rot * x_rot))
    xtvy = float(np.sum(inv * x_rot * y_rot))
    if not (math.isfinite(xtvx) and xtvx > v_floor):
        return None
    beta = xtvy / xtvx
    r = y_rot - x_rot * beta
    q = float(np.sum(inv * r * r))
    if not (math.isfinite(q) and q > v_floor):
        return None
    reml = c_reml - 0.5 * (n_eff * math.log(q) + log_det_v + math.log(xtvx))
    ml = c_ml - 0.5 * (n * math.log(q) + log_det_v)
    if not (math.isfinite(reml) and math.isfinite(ml)):
        return None
    return reml, ml, beta, q, inv, r


def gblup_fit(grm_f64, y, low=-6.0, high=6.0, tol=1e-4, max_iter=50):
    """Intercept-only REML on the spectral scale: eigh(K) (no ridge here; the caller adds g_eps), x~ = U'1,
    y~ = U'(y - mean), Brent over log10(lambda) with v_i = max(s_i + lambda, 1e-12), alpha = U (v^-1 r)."""
    k = np.asarray(grm_f64, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).ravel()
    n = y.shape[0]
    if n <= 1:
        raise RuntimeError("GBLUP REML requires at least 2 training samples.")
    y_mean = float(np.sum(y) / n)
    yc = y - y_mean
    s, u = eigh_sym(k)
    x_rot = u.sum(axis=0)
    y_rot = u.T @ yc
    n_eff = float(n - 1)

    def ev(x):
        return gblup_reml_eval(s, x_rot, y_rot, n, x)

    def cost(x):
        e = ev(x)
        return -e[0] if e is not None else 1e100

    xb, _, _ = brent_minimize(cost, low, high, tol, max_iter)
    e = ev(xb)
    if e is None:
        raise RuntimeError("GBLUP REML optimization failed to produce a valid optimum.")
    reml, ml, beta_rot, q, inv, r = e
    lbd = 10.0 ** xb
    alpha = u @ (inv * r)
    sg2 = q / max(n_eff, 1.0)
    se2 = lbd * sg2
    mean_s = float(np.sum(s) / n)
    var_g = sg2 * max(mean_s, 0.0)
    den = var_g + se2
    pve = var_g / den if (math.isfinite(den) and den > 0.0) else float("nan")
    return dict(alpha=alpha, beta0=y_mean + beta_rot, lbd=lbd, pve=pve, ml=ml, reml=reml, sigma_g2=sg2, sigma_e2=se2)


def gblup_reml_grm(k_full, train_idx, y_train, test_idx=None, g_eps=1e-8, low=-6.0, high=6.0, max_iter=50, tol=1e-4):
    """`gblup_reml_npy_grm` on an in-memory GRM: fit on K[train,train] + g_eps I, predictions K[*,train] alpha + beta0."""
    k_full = np.asarray(k_full)
    tr = np.asarray(train_idx, dtype=np.int64)
    te = np.zeros(0, dtype=np.int64) if test_idx is None else np.asarray(test_idx, dtype=np.int64)
    kt = k_full[np.ix_(tr, tr)].astype(np.float64)
    kt[np.diag_indices_from(kt)] += g_eps
    fit = gblup_fit(kt, y_train, low, high, tol, max_iter)
    pred_train = k_full[np.ix_(tr, tr)].astype(np.float64) @ fit["alpha"] + fit["beta0"]
    pred_test = k_full[np.ix_(te, tr)].astype(np.float64) @ fit["alpha"] + fit["beta0"]
    return pred_train, pred_test, fit


def grm_from_meta_additive(packed, n_samples, row_source_indices, row_flip, row_maf, sample_idx, block_rows=65536):
    """`build_grm_from_meta_stream`, Additive mode (src/stats/gblup.rs:406-652; `decode_meta_block_f32` :239-404; sample-subset
    rows through `decode_subset_row_from_full_scratch`, src/math/bedmath.rs:1359-1441, method 1) -- also what
    `grm_bed_f64_from_meta` (src/stats/grm.rs:3639-3753, method 1) returns.  -> (K f64 (n, n) scaled by 1 / sum(var) and
    symmetrised from the lower triangle, row_sum (m), sum(var))."""
    packed = np.asarray(packed, dtype=np.uint8)
    src = np.asarray(row_source_indices, dtype=np.int64)
    flip = np.asarray(row_flip, dtype=bool)
    maf = np.asarray(row_maf, dtype=np.float32)
    tr = np.asarray(sample_idx, dtype=np.int64)
    n_tr = tr.shape[0]
    m = src.shape[0]
    codes = unpack_codes(packed[src], n_samples)
    identity = n_tr == n_samples and np.array_equal(tr, np.arange(n_samples))
    z = np.empty((m, n_tr), dtype=np.float32)
    var = np.zeros(m, dtype=np.float64)
    row_sum = np.zeros(m, dtype=np.float64)
    for r in range(m):
        if identity:
            p = float(min(max(maf[r], F32(0.0)), F32(1.0)))
            mean_g = 2.0 * p
            var[r] = 2.0 * p * (1.0 - p)
            mg32 = F32(mean_g)
            lut = np.array([2.0, mg32, 1.0, 0.0] if flip[r] else [0.0, mg32, 1.0, 2.0], dtype=np.float32)
            z[r] = lut[codes[r]] - mg32
            row_sum[r] = mean_g * float(n_tr)
        else:
            dmg = F32(2.0) * min(max(maf[r], F32(0.0)), F32(1.0))
            lut = np.array([2.0, -9.0, 1.0, 0.0] if flip[r] else [0.0, -9.0, 1.0, 2.0], dtype=np.float32)
            gsub = lut[codes[r, tr]]
            gsub = np.where(gsub >= F32(0.0), gsub, dmg).astype(np.float32)
            z[r] = (gsub - dmg) * F32(1.0)
            pg = min(max(F32(0.5) * dmg, F32(0.0)), F32(1.0))
            var[r] = float(max(F32(2.0) * pg * (F32(1.0) - pg), F32(0.0)))
            row_sum[r] = float(dmg) * float(n_tr)
    acc = np.zeros((n_tr, n_tr), dtype=np.float64)
    for r0 in range(0, m, block_rows):
        blk = z[r0:r0 + block_rows]
        acc += (blk.T @ blk).astype(np.float64)          # f32 SYRK per block, f64 merge (grm.rs:1700-1772)
    var_sum = float(np.sum(var))
    k = acc * (1.0 / var_sum)
    k = np.tril(k) + np.tril(k, -1).T
    return k, row_sum, var_sum


def gblup_reml_packed_meta(packed, n_samples, row_source_indices, row_flip, row_maf, train_idx, y_train,
                           test_idx=None, train_pred_local=None, g_eps=1e-8, low=-6.0, high=6.0, max_iter=50,
                           tol=1e-4, block_rows=65536):
    """Metadata-streaming path of `gblup_reml_packed_bed` (src/stats/gblup.rs:1594-1958): GRM of the training samples
    from the 2-bit payload (`build_grm_from_meta_stream` :406-857 with `decode_meta_block_f32` :239-404, Additive mode;
    sample-subset rows through `decode_subset_row_from_full_scratch`, src/math/bedmath.rs:1359-1441, method 1), the
    spectral REML fit (:1756-1848), then marker effects instead of cross-GRM rows: m_alpha = M' alpha over the
    mean-imputed raw genotypes (`compute_malpha_from_meta_stream` :859-925, decode src/math/bedmath.rs:940-1010),
    effect_beta / effect_alpha0 (:1865-1882) and predictions alpha0 + M beta (`predict_from_effect_stream` :1037-1103).
    Returns (pred_train, pred_test, fit dict incl. effect_beta, effect_alpha0, var_sum)."""
    packed = np.asarray(packed, dtype=np.uint8)
    src = np.asarray(row_source_indices, dtype=np.int64)
    flip = np.asarray(row_flip, dtype=bool)
    maf = np.asarray(row_maf, dtype=np.float32)
    tr = np.asarray(train_idx, dtype=np.int64)
    te = np.zeros(0, dtype=np.int64) if test_idx is None else np.asarray(test_idx, dtype=np.int64)
    n_tr = tr.shape[0]
    m = src.shape[0]
    codes = unpack_codes(packed[src], n_samples)
    k, row_sum, var_sum = grm_from_meta_additive(packed, n_samples, src, flip, maf, tr, block_rows)
    k[np.diag_indices_from(k)] += g_eps
    fit = gblup_fit(k, y_train, low, high, tol, max_iter)
    alpha = fit["alpha"]
    # mean-imputed raw genotypes
    mg = np.clip(F32(2.0) * maf, F32(0.0), F32(2.0)).astype(np.float32)

    def raw(cols):
        out = np.empty((m, len(cols)), dtype=np.float32)
        for r in range(m):
            lut = np.array([2.0, mg[r], 1.0, 0.0] if flip[r] else [0.0, mg[r], 1.0, 2.0], dtype=np.float32)
            out[r] = lut[codes[r, cols]]
        return out

    m_alpha = raw(tr).astype(np.float64) @ alpha
    row_mean = row_sum / float(n_tr)
    alpha_sum = float(np.sum(alpha))
    mean_sq = float(np.sum(row_mean * row_mean))
    mean_malpha = float(np.sum(row_mean * m_alpha))
    inv_var = 1.0 / max(var_sum, 1e-12)
    effect_beta = (m_alpha - row_mean * alpha_sum) * inv_var
    effect_alpha0 = fit["beta0"] + (mean_sq * alpha_sum - mean_malpha) * inv_var
    pick = tr if train_pred_local is None else tr[np.asarray(train_pred_local, dtype=np.int64)]
    pred_train = effect_alpha0 + raw(pick).astype(np.float64).T @ effect_beta
    pred_test = effect_alpha0 + raw(te).astype(np.float64).T @ effect_beta if len(te) else np.zeros(0)
    fit = dict(fit, effect_beta=effect_beta, effect_alpha0=effect_alpha0, var_sum=var_sum, m_alpha=m_alpha)
    return pred_train, pred_test, fit


# --------------------------------------------------------------------------------------------
# Next row 8f-4: rrBLUP by preconditioned conjugate gradients over the 2-bit payload
# (src/stats/rrblup.rs:3494-4307 `rrblup_pcg_bed`, src/math/pcg.rs:870-949 `pcg_solve_into`)
# --------------------------------------------------------------------------------------------

def load_bed_2bit_packed_stats(packed, n_samples):
    """(missing_rate, maf, std_denom) f32 per SNP as `load_bed_2bit_packed` computes them
    (src/io/gfreader.rs:4460-4485) and the flip mask of `bed_packed_row_flip_mask` (src/stats/packed.rs:81-117)."""
    mi, he, ho = row_counts(np.asarray(packed, dtype=np.uint8), n_samples)
    m = mi.shape[0]
    miss = (mi.astype(np.float32) / F32(n_samples)).astype(np.float32)
    nm = n_samples - mi
    alt = he + 2 * ho
    maf = np.zeros(m, dtype=np.float32)
    std = np.zeros(m, dtype=np.float32)
    ok = nm > 0
    p = np.zeros(m, dtype=np.float32)
    p[ok] = alt[ok].astype(np.float32) / (F32(2.0) * nm[ok].astype(np.float32))
    maf[ok] = np.minimum(p[ok], F32(1.0) - p[ok])
    d = np.sqrt((F32(2.0) * p * (F32(1.0) - p)).astype(np.float32)).astype(np.float32)
    std[ok] = np.where(np.isfinite(d[ok]), d[ok], F32(0.0))
    flip = np.zeros(m, dtype=bool)
    flip[ok] = (alt[ok].astype(np.float64) / (2.0 * nm[ok].astype(np.float64))) > 0.5
    return miss, maf, std, flip


def rrblup_row_standardization(maf_keep, std_eps32):
    """`rrblup_parallel_row_standardization` (src/stats/rrblup.rs:512-566): mean = 2p, inv_sd = 1/sqrt(2p(1-p)) in f32,
    0 when the variance is <= std_eps; returns (row_mean, row_inv_sd, m_effective)."""
    p = np.clip(np.asarray(maf_keep, dtype=np.float32), F32(0.0), F32(0.5))
    mean = (F32(2.0) * p).astype(np.float32)
    var = np.maximum((F32(2.0) * p * (F32(1.0) - p)).astype(np.float32), F32(0.0))
    good = var > F32(std_eps32)
    inv = np.zeros_like(var)
    inv[good] = (F32(1.0) / np.sqrt(var[good])).astype(np.float32)
    return mean, inv, int(np.count_nonzero(good))


def rrblup_value_lut(row_mean, row_inv_sd, row_flip):
    """(m,4) f32 standardised design values by 2-bit code [00, 01, 10, 11]
    (`decode_standardized_packed_block_rows_f32_with_plan`, src/math/bedmath.rs:1199-1214; missing -> 0)."""
    mean = np.asarray(row_mean, dtype=np.float32)
    inv = np.asarray(row_inv_sd, dtype=np.float32)
    flip = np.asarray(row_flip, dtype=bool)
    g0 = np.where(flip, F32(2.0), F32(0.0)).astype(np.float32)
    g2 = np.where(flip, F32(0.0), F32(2.0)).astype(np.float32)
    lut = np.zeros((mean.shape[0], 4), dtype=np.float32)
    lut[:, 0] = (g0 - mean) * inv
    lut[:, 2] = (F32(1.0) - mean) * inv
    lut[:, 3] = (g2 - mean) * inv
    return lut


def pcg_solve_f32(b, apply_a, inv_diag, max_iter, tol, tiny=1e-20, dot=None):
    """`pcg_solve_into` for T = f32 without an initial guess (src/math/pcg.rs:870-949): vectors f32, dot products in
    f64 (`PcgScalar for f32`, :65-90), Jacobi preconditioner (:223-245). Returns (x, converged, iters, rel_res).
    `dot` (tests only): replaces the f64 dot product, e.g. by one that sums the ranks' shares of a marker-sharded solve."""
    b = np.asarray(b, dtype=np.float32)
    m = b.shape[0]
    x = np.zeros(m, dtype=np.float32)
    if m == 0 or max_iter == 0:
        return x, m == 0, 0, 0.0

    if dot is None:
        def dot(u, v):
            return float(np.dot(u.astype(np.float64), v.astype(np.float64)))

    bnorm = math.sqrt(dot(b, b))
    denom_b = max(bnorm, 1e-12)
    r = b.copy()
    z = (r * inv_diag).astype(np.float32)
    p = z.copy()
    rz_old = dot(r, z)
    rel_res = max(math.sqrt(dot(r, r)) / denom_b, 0.0)
    iters = 0
    tiny_use = max(tiny, 1e-30)
    tol_use = max(tol, 0.0)
    if math.isfinite(rel_res) and rel_res <= tol_use:
        return x, True, 0, rel_res
    converged = False
    for it in range(max_iter):
        ap = apply_a(p)
        denom = dot(p, ap)
        if not math.isfinite(denom) or denom <= tiny_use:
            break
        alpha = F32(rz_old / denom)
        x = (x + alpha * p).astype(np.float32)
        r = (r - alpha * ap).astype(np.float32)
        rel_res = max(math.sqrt(dot(r, r)) / denom_b, 0.0)
        iters = it + 1
        if math.isfinite(rel_res) and rel_res <= tol_use:
            converged = True
            break
        z = (r * inv_diag).astype(np.float32)
        rz_new = dot(r, z)
        if not math.isfinite(rz_new) or rz_new <= tiny_use:
            break
        beta = F32(rz_new / max(rz_old, tiny_use))
        p = (z + beta * p).astype(np.float32)
        rz_old = rz_new
    return x, converged, iters, rel_res


def rrblup_pcg_packed(packed, n_samples, maf, row_flip, train_idx, y_train, test_idx=None, train_pred_local=None,
                      site_keep=None, lambda_value=10000.0, tol=1e-4, max_iter=100, std_eps=1e-12,
                      compute_trainvar=False, row_mean=None, row_inv_sd=None):
    """`rrblup_pcg_bed` on a resident packed payload (src/stats/rrblup.rs:3519-4307): marker effects beta solve
    (Z_c Z_c' + lambda I) beta = Z y_c, Z (m, n_train) the standardised genotypes of the training samples, Z_c its
    row-centred form applied implicitly (`RrblupPcgOperator::apply` :865-929), right-hand side and Jacobi diagonal
    from one pre-pass (`rrblup_prepare_rhs_diag` :650-845, `row_major_block_prepare_rhs_diag_f32` :396-466).
    Returns the reference's tuple (pred_train (k,1), pred_test (t,1), pve_trainvar, converged, iters, rel_res,
    m_effective, pve_lambda_vc, k_trace_mean, beta f32 (m))."""
    packed = np.asarray(packed, dtype=np.uint8)
    m_total = packed.shape[0]
    maf_full = np.asarray(maf, dtype=np.float32)
    flip_full = np.asarray(row_flip, dtype=bool)
    rows = None
    maf_keep, flip_keep = maf_full, flip_full
    if site_keep is not None:
        keep_idx = np.nonzero(np.asarray(site_keep, dtype=bool))[0]
        if keep_idx.shape[0] == 0:
            raise RuntimeError("No SNPs remained after applying site_keep mask.")
        if not (keep_idx.shape[0] == m_total):
            rows = keep_idx
            maf_keep = np.clip(maf_full[keep_idx], F32(0.0), F32(0.5))
            flip_keep = flip_full[keep_idx]
    eff_m = maf_keep.shape[0]
    tr = np.asarray(train_idx, dtype=np.int64)
    te = np.zeros(0, dtype=np.int64) if test_idx is None else np.asarray(test_idx, dtype=np.int64)
    y = np.asarray(y_train, dtype=np.float64)
    n_train = tr.shape[0]
    lambda_use = F32(max(lambda_value, 1e-8))
    tol_use = max(tol, 1e-12)
    std_eps32 = F32(max(std_eps, 1e-12))
    if row_mean is not None and row_inv_sd is not None:
        rm = np.asarray(row_mean, dtype=np.float32)
        ri = np.asarray(row_inv_sd, dtype=np.float32)
        if rm.shape[0] != eff_m:
            rm, ri = rm[rows], ri[rows]
        m_effective = int(np.count_nonzero(np.isfinite(ri) & (ri > 0)))
    else:
        rm, ri, m_effective = rrblup_row_standardization(maf_keep, std_eps32)
    lut = rrblup_value_lut(rm, ri, flip_keep)
    codes = unpack_codes(packed if rows is None else packed[rows], n_samples)

    def decode(cols):
        return np.take_along_axis(lut, codes[:, cols].astype(np.int64), axis=1).astype(np.float32)

    z = decode(tr)                                     # (m, n_train) f32
    y_mean = float(np.sum(y)) / float(n_train)
    y_c = (y - y_mean).astype(np.float32)
    b = (z @ y_c).astype(np.float32)                   # row_major_block_mul_vec_f32 (f32 GEMV)
    z64 = z.astype(np.float64)
    s = z64.sum(axis=1)
    mean = s / float(n_train)
    ss = np.maximum((z64 * z64).sum(axis=1) - float(n_train) * mean * mean, 0.0)
    sum_ss_global = float(np.sum(ss))
    train_row_mean = mean.astype(np.float32)
    diag_inv = (F32(1.0) / np.maximum(ss.astype(np.float32) + lambda_use, F32(1e-12))).astype(np.float32)

    def apply_a(p):
        xp = (z.T @ p).astype(np.float32)
        ap = (z @ xp).astype(np.float32)
        mean_dot = F32(float(np.dot(train_row_mean.astype(np.float64), p.astype(np.float64))))
        ap = (ap - F32(n_train) * train_row_mean * mean_dot).astype(np.float32)
        return (ap + lambda_use * p).astype(np.float32)

    beta, converged, iters, rel_res = pcg_solve_f32(b, apply_a, diag_inv, max_iter, tol_use)
    acc = F32(0.0)
    for v in (train_row_mean * beta).astype(np.float32):   # `.sum::<f32>()`, sequential
        acc = F32(acc + v)
    alpha_use = F32(F32(y_mean) - acc)
    need_all = train_pred_local is None
    pred_train = np.zeros(0)
    pve_trainvar = float("nan")
    if compute_trainvar or need_all:
        full = ((z.T @ beta).astype(np.float32) + alpha_use).astype(np.float32).astype(np.float64)
        pred_train = full if need_all else full[np.asarray(train_pred_local, dtype=np.int64)]
        if compute_trainvar:
            resid = y - full
            vg = float(np.var(full, ddof=1)) if n_train > 1 else 0.0
            ve = float(np.var(resid, ddof=1)) if n_train > 1 else 0.0
            pve_trainvar = vg / (vg + ve) if (vg + ve) > 0 and math.isfinite(vg + ve) else float("nan")
    elif len(train_pred_local) > 0:
        cols = tr[np.asarray(train_pred_local, dtype=np.int64)]
        pred_train = ((decode(cols).T @ beta).astype(np.float32) + alpha_use).astype(np.float32).astype(np.float64)
    k_trace_mean = sum_ss_global / (float(m_effective) * float(n_train)) if (n_train > 0 and m_effective > 0) else float("nan")
    pve_lambda_vc = float("nan")
    if math.isfinite(k_trace_mean) and k_trace_mean > 0 and m_effective > 0:
        dv = k_trace_mean + float(lambda_use) / float(m_effective)
        if math.isfinite(dv) and dv > 0:
            pve_lambda_vc = k_trace_mean / dv
    pred_test = np.zeros(0)
    if te.shape[0] > 0:
        pred_test = (decode(te).T @ beta).astype(np.float32).astype(np.float64) + float(alpha_use)
    return (pred_train.reshape(-1, 1), pred_test.reshape(-1, 1), pve_trainvar, bool(converged), int(iters),
            float(rel_res), int(m_effective), pve_lambda_vc, k_trace_mean, beta)


# --------------------------------------------------------------------------------------------
# Row 8f-4, second half: Haseman-Elston variance components with stochastic traces over the same matrix-free GRM
# operator (src/stats/he.rs:1633-2070 `he_variance_components_with_source`, `he_pcg_bed` :2073-2636)
# --------------------------------------------------------------------------------------------

_M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    """src/stats/he.rs:899-906."""
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def he_rademacher_probe(seed: int, probe_idx: int, n: int) -> np.ndarray:
    """+-1 probe of index `probe_idx` (he.rs:1871-1881)."""
    state = splitmix64((seed ^ ((probe_idx * 0x517CC1B727220A95) & _M64)) & _M64)
    out = np.empty(n, dtype=np.float32)
    for i in range(n):
        state = splitmix64(state)
        out[i] = 1.0 if (state & 1) == 0 else -1.0
    return out


def he_row_standardization(packed_rows, n_samples, row_flip, row_maf, sample_idx, std_eps32, use_train_maf=True):
    """`build_row_standardization_stats_with_options` without the sample diagonal (he.rs:454-640): oriented minor
    frequency from the input (:367-384) or, with `use_train_maf`, from the training samples' calls (:506-534)."""
    flip = np.asarray(row_flip, dtype=bool)
    raw = np.asarray(row_maf, dtype=np.float32)
    if not np.all(np.isfinite(raw)):
        raise RuntimeError("row_maf contains non-finite values")
    af = np.clip(raw, F32(0.0), F32(1.0))
    p = np.where(af <= F32(0.5), af, np.where(flip, F32(1.0) - af, af)).astype(np.float32)
    p = np.clip(p, F32(0.0), F32(1.0))
    if use_train_maf:
        mi, he, ho = row_counts(packed_rows, n_samples, sample_idx)
        nm = len(sample_idx) - mi
        alt = he + 2 * ho
        dos = np.where(flip, 2 * nm - alt, alt)
        ok = nm > 0
        pt = np.zeros_like(p)
        pt[ok] = dos[ok].astype(np.float32) / (F32(2.0) * nm[ok].astype(np.float32))
        p = np.where(ok, pt, p).astype(np.float32)
    p = np.clip(p, F32(0.0), F32(1.0))
    mean = (F32(2.0) * p).astype(np.float32)
    var = np.maximum((F32(2.0) * p * (F32(1.0) - p)).astype(np.float32), F32(0.0))
    good = var > F32(std_eps32)
    inv = np.zeros_like(var)
    inv[good] = (F32(1.0) / np.sqrt(var[good])).astype(np.float32)
    return mean, inv, int(np.count_nonzero(good))


def he_solve_2x2(a00, a01, a11, b0, b1):
    """he.rs:874-897."""
    det = a00 * a11 - a01 * a01
    det_scale = max(abs(a00) + abs(a11) + 2.0 * abs(a01), 1.0)
    det_floor = det_scale * det_scale * EPS64
    if not math.isfinite(det) or abs(det) <= det_floor:
        raise RuntimeError(f"HE 2x2 solve is singular/ill-conditioned: det={det}, floor={det_floor}")
    return (b0 * a11 - b1 * a01) / det, (a00 * b1 - a01 * b0) / det


def he_project_nnls_2x2(a00, a01, a11, b0, b1, x0u, x1u):
    """he.rs:815-871: best feasible point among the interior solution, the two boundaries and the origin.
    Returns (x0, x1, projected, status) with status 0 interior, 1 sigma_g2 = 0, 2 sigma_e2 = 0, 3 origin."""
    def obj(x0, x1):
        r0 = a00 * x0 + a01 * x1 - b0
        r1 = a01 * x0 + a11 * x1 - b1
        return r0 * r0 + r1 * r1
    best = (0.0, 0.0, float("inf"), 3)
    cands = [(x0u, x1u, 0)]
    c1 = a01 * a01 + a11 * a11
    if math.isfinite(c1) and c1 > 0:
        cands.append((0.0, max((a01 * b0 + a11 * b1) / c1, 0.0), 1))
    c0 = a00 * a00 + a01 * a01
    if math.isfinite(c0) and c0 > 0:
        cands.append((max((a00 * b0 + a01 * b1) / c0, 0.0), 0.0, 2))
    cands.append((0.0, 0.0, 3))
    for x0, x1, st in cands:
        if not (math.isfinite(x0) and math.isfinite(x1)) or x0 < 0 or x1 < 0:
            continue
        o = obj(x0, x1)
        if math.isfinite(o) and o < best[2]:
            best = (x0, x1, o, st)
    if not math.isfinite(best[2]):
        return 0.0, 0.0, True, 3
    scale = max(abs(x0u), abs(x1u), 1.0)
    tolp = 1e-10 * scale
    projected = best[3] != 0 or abs(best[0] - x0u) > tolp or abs(best[1] - x1u) > tolp
    return best[0], best[1], projected, best[3]


def rrblup_exact_reml_cost_from_spectrum(lam, eigvals, y_proj, y_resid_ss, n_eff):
    """src/stats/rrblup.rs:1568-1611."""
    if not (math.isfinite(lam) and lam > 0.0):
        return math.inf
    r = len(eigvals)
    if r != len(y_proj) or n_eff == 0 or n_eff < r:
        return math.inf
    quad = log_det = y_proj_ss = 0.0
    for k in range(r):
        s, yk = float(eigvals[k]), float(y_proj[k])
        if not (math.isfinite(s) and s >= 0.0 and math.isfinite(yk)):
            return math.inf
        vk = s + lam
        if not (math.isfinite(vk) and vk > 0.0):
            return math.inf
        quad += (yk * yk) / vk
        log_det += math.log(vk)
        y_proj_ss += yk * yk
    null_df = max(n_eff - r, 0)
    null_ss = max(y_resid_ss - y_proj_ss, 0.0)
    if null_df > 0:
        quad += null_ss / lam
        log_det += float(null_df) * math.log(lam)
    if not (math.isfinite(quad) and quad > 0.0 and math.isfinite(log_det)):
        return math.inf
    return 0.5 * (float(n_eff) * math.log(quad) + log_det)


def rrblup_exact_snp_packed(packed, n_samples, train_idx, y_train, test_idx=None, train_pred_local=None, site_keep=None,
                            maf=None, row_flip=None, row_mean=None, row_inv_sd=None, log10_lambda_low=-6.0,
                            log10_lambda_high=6.0, reml_tol=1e-4, reml_max_iter=50, std_eps=1e-12):
    """Exact marker-space rrBLUP (src/stats/rrblup.rs:3179-3490; cache :1613-1899, fit :1951-2430): A* = Z Z' over the
    training samples (Z (m, n_train) standardised genotypes, f32 values accumulated in f64 by DSYRK) minus the rank-one
    centring term, eigendecomposition, REML over log10 lambda by Brent on the spectrum, beta = V diag(1 / (s + lambda)) V' z.
    Returns the reference's tuple (pred_train (k,1), pred_test (t,1), pve_trainvar, lambda, reml, (var_g, sigma_e2),
    m_effective, y_mean, beta f32 (m), row_mean f32, row_inv_sd f32, backend)."""
    packed = np.asarray(packed, dtype=np.uint8)
    m_total = packed.shape[0]
    maf_full = np.asarray(maf, dtype=np.float32)
    flip_full = np.asarray(row_flip, dtype=bool)
    rows = None
    maf_keep, flip_keep = maf_full, flip_full
    if site_keep is not None:
        keep_idx = np.nonzero(np.asarray(site_keep, dtype=bool))[0]
        if keep_idx.shape[0] == 0:
            raise RuntimeError("No SNPs remained after applying site_keep mask.")
        if keep_idx.shape[0] != m_total:
            rows = keep_idx
            maf_keep = np.clip(maf_full[keep_idx], F32(0.0), F32(0.5))
            flip_keep = flip_full[keep_idx]
    eff_m = maf_keep.shape[0]
    tr = np.asarray(train_idx, dtype=np.int64)
    te = np.zeros(0, dtype=np.int64) if test_idx is None else np.asarray(test_idx, dtype=np.int64)
    y = np.asarray(y_train, dtype=np.float64)
    n_train = tr.shape[0]
    if n_train <= 1:
        raise RuntimeError("rrblup_exact_snp_packed requires at least two training samples.")
    std_eps32 = F32(max(std_eps, 1e-12))
    if row_mean is not None and row_inv_sd is not None:
        rm = np.asarray(row_mean, dtype=np.float32)
        ri = np.asarray(row_inv_sd, dtype=np.float32)
        if rm.shape[0] != eff_m:
            rm, ri = rm[rows], ri[rows]
        m_effective = int(np.count_nonzero(np.isfinite(ri) & (ri > 0)))
    else:
        rm, ri, m_effective = rrblup_row_standardization(maf_keep, std_eps32)
    lut = rrblup_value_lut(rm, ri, flip_keep)
    codes = unpack_codes(packed if rows is None else packed[rows], n_samples)

    def decode(cols):
        return np.take_along_axis(lut, codes[:, cols].astype(np.int64), axis=1).astype(np.float32)

    z = decode(tr)                                      # (m, n_train) f32
    z64 = z.astype(np.float64)
    a_star = z64 @ z64.T                                # cblas_dsyrk over the sample blocks (:1757-1775)
    row_sum = z64.sum(axis=1)
    train_row_mean = (row_sum / float(n_train)).astype(np.float32)
    a_star -= np.outer(row_sum, row_sum) / float(n_train)          # symmetrize_upper_minus_rank1_in_place (:1549-1565)
    a_star = 0.5 * (a_star + a_star.T)
    evals_all, evecs = np.linalg.eigh(a_star)
    max_eval = max(float(evals_all[-1]), 0.0)
    tol = EPS64 * max(max_eval, 1.0) * float(max(eff_m, 1))
    n_eff = n_train - 1
    pos = np.nonzero(evals_all > tol)[0]
    if pos.size == 0:
        raise RuntimeError("rrblup_exact_snp_packed found no positive spectrum after centering.")
    keep_start = int(pos[0])
    if eff_m - keep_start > n_eff:
        keep_start = eff_m - n_eff
    eigvals = evals_all[keep_start:]
    v = evecs[:, keep_start:]                           # (m, rank)
    rank = eigvals.shape[0]
    y_mean = float(np.sum(y)) / float(n_train)
    y_c32 = (y - y_mean).astype(np.float32)
    y_center_ss = float(np.sum((y - y_mean) ** 2))
    zy = (z @ y_c32).astype(np.float32).astype(np.float64)          # f32 GEMV per sample block, summed in f64 (:2080-2110)
    coeff = v.T @ zy
    y_proj = coeff / np.sqrt(eigvals)
    low, high = min(log10_lambda_low, log10_lambda_high), max(log10_lambda_low, log10_lambda_high)
    best_log10, best_cost, _ = brent_minimize(
        lambda x: rrblup_exact_reml_cost_from_spectrum(10.0 ** x, eigvals, y_proj, y_center_ss, n_eff), low, high, reml_tol,
        reml_max_iter)
    lambda_opt = max(10.0 ** best_log10, 1e-12)
    denom = eigvals + lambda_opt
    quad = float(np.sum(y_proj * y_proj / denom))
    y_proj_ss = float(np.sum(y_proj * y_proj))
    g_center_ss = float(np.sum(eigvals * coeff * coeff / (denom * denom)))
    w = coeff / denom
    null_ss = max(y_center_ss - y_proj_ss, 0.0)
    if n_eff - rank > 0:
        quad += null_ss / lambda_opt
    sigma_beta2 = quad / float(n_eff)
    sigma_e2 = lambda_opt * sigma_beta2
    beta = (v @ w).astype(np.float32)
    alpha_use = y_mean - float(np.sum(train_row_mean.astype(np.float64) * beta.astype(np.float64)))
    var_g = g_center_ss / float(n_train - 1) if n_train > 1 else 0.0
    den = var_g + sigma_e2
    pve = var_g / den if (math.isfinite(den) and den > 0.0) else math.nan

    def predict(cols):
        zz = decode(cols)                               # pcg_x_mul_samples: f32 products, f32 output
        return (zz.T @ beta).astype(np.float32).astype(np.float64) + alpha_use

    if train_pred_local is not None:
        pick = np.asarray(train_pred_local, dtype=np.int64)
        pred_train = predict(tr[pick]) if pick.size else np.zeros(0)
    else:
        pred_train = predict(tr)
    pred_test = predict(te) if te.size else np.zeros(0)
    return (pred_train.reshape(-1, 1), pred_test.reshape(-1, 1), pve, lambda_opt, -best_cost, (var_g, sigma_e2),
            m_effective, y_mean, beta, rm, ri, "numpy")


def he_pcg_packed(packed, n_samples, maf, row_flip, train_idx, y_train, site_keep=None, trace_samples=32, tol=1e-6,
                  std_eps=1e-12, use_train_maf=True, exact_trace_debug=False, exact_trace_max_n=256, seed=20260512,
                  x_cov=None):
    """`he_pcg_bed`, resident packed form (he.rs:2101-2636 -> :1633-2070). K = Z'Z / m_effective on the training
    samples (Z standardised with the training-sample allele frequency, missing -> 0), P the projector off
    [1, x_cov]; y'PKPy, y'Py, tr(PKP) and tr((PKP)^2) by Hutchinson probes (or exactly); 2x2 HE normal equations with
    the non-negative projection. Returns the reference's 12-tuple (sigma_g2, sigma_e2, h2, converged, iters, rel_res,
    m_effective, tr_k2, y_ky, y_y, lambda, tr_k2_solve) plus (tr_k, tr_p, status) for tests."""
    packed = np.asarray(packed, dtype=np.uint8)
    m_total = packed.shape[0]
    maf = np.asarray(maf, dtype=np.float32)
    flip = np.asarray(row_flip, dtype=bool)
    rows = np.arange(m_total)
    if site_keep is not None:
        rows = np.nonzero(np.asarray(site_keep, dtype=bool))[0]
        if rows.shape[0] == 0:
            raise RuntimeError("No SNPs remained after applying site_keep mask.")
    eff_m = rows.shape[0]
    tr = np.asarray(train_idx, dtype=np.int64)
    y = np.asarray(y_train, dtype=np.float64)
    n = tr.shape[0]
    pk = packed[rows]
    mean, inv, m_eff = he_row_standardization(pk, n_samples, flip[rows], maf[rows], tr, F32(max(std_eps, 1e-12)),
                                              use_train_maf)
    if m_eff == 0:
        raise RuntimeError("No effective SNPs after std_eps filtering")
    lut = rrblup_value_lut(mean, inv, flip[rows])
    codes = unpack_codes(pk, n_samples)[:, tr].astype(np.int64)
    z = np.take_along_axis(lut, codes, axis=1).astype(np.float32)            # (m, n)
    m_scale = F32(m_eff)
    inv_m = F32(1.0) / max(m_scale, F32(1.0))
    # projector off [1, x_cov] (he.rs:206-354)
    if x_cov is not None:
        xc = np.asarray(x_cov, dtype=np.float64)
        if xc.shape[0] == n_samples and n != n_samples:
            xc = xc[tr]
        x = np.concatenate([np.ones((n, 1)), xc.reshape(n, -1)], axis=1)
    else:
        x = np.ones((n, 1))
    p = x.shape[1]
    if n <= p:
        raise RuntimeError(f"HE projection requires n > rank(X): n={n}, p(with intercept)={p}")
    xtx = x.T @ x

    def proj64(v):
        return v - x @ np.linalg.solve(xtx, x.T @ v)

    def proj32(v32):
        v = v32.astype(np.float64)
        return (v32 - (x @ np.linalg.solve(xtx, x.T @ v)).astype(np.float32)).astype(np.float32)

    def apply_k(v32):
        t = (z @ v32).astype(np.float32)
        return ((z.T @ t).astype(np.float32) * inv_m).astype(np.float32)

    y32 = proj64(y).astype(np.float32)
    k_y = apply_k(y32)
    y_ky = float(np.dot(y32.astype(np.float64), k_y.astype(np.float64)))
    y_y = float(np.dot(y32.astype(np.float64), y32.astype(np.float64)))
    exact = bool(exact_trace_debug) and n <= max(int(exact_trace_max_n), 1)
    tr_k_acc = tr_k2_acc = 0.0
    if exact:
        for i in range(n):
            e = np.zeros(n, dtype=np.float32)
            e[i] = 1.0
            v = proj32(apply_k(proj32(e)))
            tr_k_acc += float(v[i])
            tr_k2_acc += float(np.dot(v.astype(np.float64), v.astype(np.float64)))
        tr_k, tr_k2 = tr_k_acc, tr_k2_acc
    else:
        for t in range(trace_samples):
            zp = proj32(he_rademacher_probe(seed, t, n))
            v = proj32(apply_k(zp))
            tr_k_acc += float(np.dot(zp.astype(np.float64), v.astype(np.float64)))
            tr_k2_acc += float(np.dot(v.astype(np.float64), v.astype(np.float64)))
        tr_k, tr_k2 = tr_k_acc / trace_samples, tr_k2_acc / trace_samples
    if not (math.isfinite(tr_k) and tr_k > 0):
        raise RuntimeError(f"estimated Tr(PKP) is invalid: {tr_k}. Try increasing trace_samples.")
    if not (math.isfinite(tr_k2) and tr_k2 > 0):
        raise RuntimeError(f"estimated Tr((PKP)^2) is invalid: {tr_k2}. Try increasing trace_samples.")
    tr_p = max(float(n) - float(p), 1.0)
    tr_k2_solve = max(tr_k2, (tr_k * tr_k) / tr_p + tr_p * 1e-6)
    if tr_k2_solve > tr_k2 * 1.05:
        raise RuntimeError("Tr((PKP)^2) stochastic estimate violates PSD bound too much: "
                           f"raw={tr_k2}, adjusted={tr_k2_solve}. Increase trace_samples.")
    sg_u, se_u = he_solve_2x2(tr_k2_solve, tr_k, tr_p, y_ky, y_y)
    sg, se, _proj, status = he_project_nnls_2x2(tr_k2_solve, tr_k, tr_p, y_ky, y_y, sg_u, se_u)
    r0 = tr_k2_solve * sg + tr_k * se - y_ky
    r1 = tr_k * sg + tr_p * se - y_y
    rel_res = math.sqrt(r0 * r0 + r1 * r1) / max(math.sqrt(y_ky * y_ky + y_y * y_y), 1e-20)
    converged = math.isfinite(rel_res) and rel_res <= max(tol, 1e-12)
    den = sg + se
    h2 = sg / den if (math.isfinite(den) and den > 0) else float("nan")
    lam = se / sg if (math.isfinite(sg) and sg > 0) else float("inf")
    return (sg, se, h2, converged, 1, rel_res, min(m_eff, eff_m), tr_k2, y_ky, y_y, lam, tr_k2_solve, tr_k, tr_p, status)
