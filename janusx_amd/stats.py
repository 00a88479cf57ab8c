"""Per-SNP statistics, QC filters and design-value LUTs from integer genotype counts (host side, vectorised).

The counts come from the GPU (`jxg_row_counts_p32`); everything below is O(m) float logic that must agree
bit for bit with the reference so that the kept-SNP set, `af` and `miss` columns are identical.  numpy float32 /
float64 arithmetic is IEEE like Rust's, so the expressions are written with the reference's types and order:

* `gwas_scan_row_stats`   : src/stats/lmm.rs:1258-1320  (all compares in f32)
* `stream_grm_row_prepare`: src/stats/grm.rs:1465-1536  (all in f64, thresholds widened from f32)
* `grm_lut_from_maf`      : src/decode/decode.rs:813-839, 558-566 (packed GRM route)
* `scan_lut_from_counts`  : src/decode/decode.rs:163-189, 218-221 (scan design rows)
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def gwas_scan_row_stats(counts: np.ndarray, n: int, maf_thr: float, miss_thr: float, het_thr: float):
    """counts (m,3) int = (missing, het, hom_alt) over the n selected samples.
    -> keep (bool), af (f32 alt_freq), miss_rate (f32)."""
    counts = np.asarray(counts, dtype=np.int64)
    missing, het, hom = counts[:, 0], counts[:, 1], counts[:, 2]
    maf_thr, miss_thr, het_thr = F32(maf_thr), F32(miss_thr), F32(het_thr)
    nm = np.maximum(n - missing, 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        miss_rate = (missing.astype(np.float32) / F32(n)) if n > 0 else np.ones(len(missing), dtype=np.float32)
        keep = ~(miss_rate > miss_thr)
        zero_nm = nm == 0
        nmf = nm.astype(np.float32)
        if het_thr > F32(0.0):
            het_rate = het.astype(np.float32) / nmf
            keep &= ~((~zero_nm) & (het_rate > het_thr))
        alt_sum = (het + 2 * hom).astype(np.float32)
        alt_freq = alt_sum / (F32(2.0) * nmf)
        maf_v = np.minimum(alt_freq, F32(1.0) - alt_freq)
        keep &= ~((~zero_nm) & (maf_v < maf_thr))
    if maf_thr > F32(0.0):
        keep &= ~zero_nm
    af = np.where(zero_nm, F32(0.0), alt_freq).astype(np.float32)
    return keep, af, miss_rate.astype(np.float32)


def packed_prep_row_stats(counts: np.ndarray, n: int, maf_thr: float, miss_thr: float, het_thr: float):
    """QC of the packed workflow (`prepare_bed_2bit_packed`, src/io/gfreader.rs:5380-5420, 1911-1929, 6340-6400):
    f32 missing-rate and MAF compares, f64 het-rate compare (filter on only when het_thr > 0).
    -> keep, miss_rate f32, maf f32 (= alt allele frequency clamped to [0, 1]), std_denom f32 = sqrt(2p(1-p))."""
    counts = np.asarray(counts, dtype=np.int64)
    missing, het, hom = counts[:, 0], counts[:, 1], counts[:, 2]
    maf_thr, miss_thr, het_thr = F32(maf_thr), F32(miss_thr), F32(het_thr)
    nm = np.maximum(n - missing, 0)
    has = nm > 0
    alt_sum = het + 2 * hom
    with np.errstate(divide="ignore", invalid="ignore"):
        miss = ((n - nm).astype(np.float32) / F32(n)) if n > 0 else np.zeros(len(nm), dtype=np.float32)
        af = np.where(has, alt_sum.astype(np.float32) / (F32(2.0) * nm.astype(np.float32)), F32(0.0)).astype(np.float32)
        p = alt_sum.astype(np.float64) / (2.0 * nm.astype(np.float64))
        d = np.sqrt(2.0 * p * (1.0 - p)).astype(np.float32)
        std = np.where(has & np.isfinite(d), d, F32(0.0)).astype(np.float32)
        pass_maf = np.minimum(af, F32(1.0) - af) >= maf_thr
        keep = np.where(has, pass_maf, bool(maf_thr <= F32(0.0)))
        if het_thr > F32(0.0):
            keep &= ~(has & ((het.astype(np.float64) / nm.astype(np.float64)) > float(het_thr)))
        keep &= ~(miss > miss_thr)
    return keep.astype(bool), miss.astype(np.float32), np.clip(af, F32(0.0), F32(1.0)).astype(np.float32), std


def stream_grm_row_prepare(counts: np.ndarray, n_samples: int, method: int, maf_thr: float, miss_thr: float,
                           het_thr: float):
    """-> keep, mean_g (f32), std_scale (f32), flip (bool), var (f64). Thresholds are clamped like
    src/stats/grm.rs:4709-4711."""
    counts = np.asarray(counts, dtype=np.int64)
    missing, het, hom = counts[:, 0], counts[:, 1], counts[:, 2]
    maf_thr32 = F32(min(max(float(maf_thr), 0.0), 0.5))
    miss_thr32 = F32(min(max(float(miss_thr), 0.0), 1.0))
    het_thr32 = F32(min(max(float(het_thr), 0.0), 1.0))
    maf_thr64, miss_thr64, het_thr64 = float(maf_thr32), float(miss_thr32), float(het_thr32)
    eps64 = float(F32(1e-12))
    nm = n_samples - missing
    nmf = nm.astype(np.float64)
    m = len(missing)
    keep = np.ones(m, dtype=bool)
    with np.errstate(divide="ignore", invalid="ignore"):
        if het_thr32 > F32(0.0):
            keep &= ~((nm > 0) & ((het.astype(np.float64) / nmf) > het_thr64))
        missing_rate = 1.0 - (nmf / float(n_samples))
        keep &= ~(missing_rate > miss_thr64)
        zero_nm = nm == 0
        if maf_thr32 > F32(0.0):
            keep &= ~zero_nm
        alt_sum = (het + 2 * hom).astype(np.float64)
        alt_freq = alt_sum / (2.0 * nmf)
        flip = alt_freq > 0.5
        alt_sum = np.where(flip, 2.0 * nmf - alt_sum, alt_sum)
        alt_freq = np.where(flip, alt_sum / (2.0 * nmf), alt_freq)
        maf = np.minimum(alt_freq, 1.0 - alt_freq)
        keep &= ~((~zero_nm) & (maf < maf_thr64))
        mean_g = (alt_sum / nmf).astype(np.float32)
        var = np.maximum(2.0 * alt_freq * (1.0 - alt_freq), 0.0)
        if method == 2:
            scale = np.where(var > eps64, (1.0 / np.sqrt(var)), 0.0).astype(np.float32)
        else:
            scale = np.ones(m, dtype=np.float32)
    flip = np.where(zero_nm, False, flip)
    mean_g = np.where(zero_nm, F32(0.0), mean_g).astype(np.float32)
    scale = np.where(zero_nm, F32(0.0 if method == 2 else 1.0), scale).astype(np.float32)
    var = np.where(zero_nm, 0.0, var)
    return keep, mean_g, scale, flip, var


def grm_lut_from_mean_scale(mean_g, scale, flip):
    """(m,4) f32 LUT indexed by code [00, 01, 10, 11] for the stream-GRM route (decode.rs:446-461)."""
    mean_g = np.asarray(mean_g, dtype=np.float32)
    scale = np.asarray(scale, dtype=np.float32)
    flip = np.asarray(flip, dtype=bool)
    g0 = np.where(flip, F32(2.0), F32(0.0)).astype(np.float32)
    g2 = np.where(flip, F32(0.0), F32(2.0)).astype(np.float32)
    lut = np.zeros((len(mean_g), 4), dtype=np.float32)
    lut[:, 0] = (g0 - mean_g) * scale
    lut[:, 2] = (F32(1.0) - mean_g) * scale
    lut[:, 3] = (g2 - mean_g) * scale
    return lut


def grm_lut_from_maf(row_maf, row_flip, method: int):
    """Packed-GRM LUT from the passed maf (decode.rs:813-839 / bedmath.rs:1208-1224; eps = 1e-12 f32)."""
    p = np.clip(np.asarray(row_maf, dtype=np.float32), F32(0.0), F32(1.0))
    mean_g = (F32(2.0) * p).astype(np.float32)
    var = (F32(2.0) * p * (F32(1.0) - p)).astype(np.float32)
    if method == 2:
        with np.errstate(divide="ignore", invalid="ignore"):
            scale = np.where(var > F32(1e-12), F32(1.0) / np.sqrt(var), F32(0.0)).astype(np.float32)
    else:
        scale = np.ones_like(p)
    return grm_lut_from_mean_scale(mean_g, scale, row_flip)


GENETIC_MODELS = ("add", "dom", "rec", "het")


def genetic_model_code(model) -> int:
    """`PackedGeneticModel::parse` (src/decode/decode.rs:107-119): case-insensitive add / dom / rec / het -> 0 .. 3."""
    m = str(model).lower()
    if m not in GENETIC_MODELS:
        raise RuntimeError("model must be one of: add, dom, rec, het")
    return GENETIC_MODELS.index(m)


def _apply_genetic_model(code: int, g):
    """`PackedGeneticModel::apply` (decode.rs:132-160) on f32 table values (evaluated in f64 like the reference)."""
    g = np.asarray(g, dtype=np.float32)
    g64 = g.astype(np.float64)
    if code == 1:
        return (g64 > 0.0).astype(np.float32)
    if code == 2:
        return (np.abs(g64 - 2.0) < 1e-6).astype(np.float32)
    if code == 3:
        return (np.abs(g64 - 1.0) < 1e-6).astype(np.float32)
    return g


def scan_lut_from_counts(row_maf, row_flip, counts, n: int, model="add"):
    """(m,4) f32 scan design LUT: the genetic model applied to [0, mu, 1, 2] (or flipped; the imputed entry mu = 2 maf
    included, src/decode/decode.rs:163-178), minus the actual row mean (f64 sum / n -> f32; the sum of the f32 table
    values is exact in f64, so counts reproduce it)."""
    code = genetic_model_code(model)
    maf = np.asarray(row_maf, dtype=np.float32)
    flip = np.asarray(row_flip, dtype=bool)
    counts = np.asarray(counts, dtype=np.int64)
    mu = _apply_genetic_model(code, np.maximum(2.0 * maf.astype(np.float64), 0.0).astype(np.float32))
    v0 = _apply_genetic_model(code, np.where(flip, F32(2.0), F32(0.0)).astype(np.float32))
    v2 = _apply_genetic_model(code, np.full(len(maf), 1.0, dtype=np.float32))
    v3 = _apply_genetic_model(code, np.where(flip, F32(0.0), F32(2.0)).astype(np.float32))
    c00 = (n - counts[:, 0] - counts[:, 1] - counts[:, 2]).astype(np.float64)
    total = (c00 * v0.astype(np.float64) + counts[:, 0] * mu.astype(np.float64) + counts[:, 1] * v2.astype(np.float64) +
             counts[:, 2] * v3.astype(np.float64))
    mean = (total / float(n)).astype(np.float32)
    lut = np.empty((len(maf), 4), dtype=np.float32)
    lut[:, 0] = v0 - mean
    lut[:, 1] = mu - mean
    lut[:, 2] = v2 - mean
    lut[:, 3] = v3 - mean
    return lut


# ------------------------------------------------------------------------------------------------
# warm-start chains of the exact scan (src/stats/lmm.rs:134-161)
# ------------------------------------------------------------------------------------------------

def env_truthy(name: str) -> bool:
    """`env_truthy` (src/stats/common.rs:88-96): 1 / true / yes / y / on, case-insensitive, trimmed."""
    import os
    return os.environ.get(name, "").strip().lower() in ("1", "true", "yes", "y", "on")


def _halve(b0: int, b1: int, depth: int, out: list):
    """rayon's adaptive splitter on an indexed producer: a piece of `len` items is cut at len / 2 (left half first)."""
    if depth <= 0 or b1 - b0 < 2:
        out.append(b0)
        return
    mid = b0 + (b1 - b0) // 2
    _halve(b0, mid, depth - 1, out)
    _halve(mid, b1, depth - 1, out)


def warm_chain_offsets(block_starts, m: int, pieces: int = 1):
    """Chain offsets (int64, ascending, first 0, last m) of the reference's warm-start chains over m scanned rows.

    The reference carries the previous SNP's optimum as the next SNP's Brent start inside the per-worker state of
    `run_rotated_assoc_block_f32` (src/stats/reml.rs:69-105, `for_each_init`; src/stats/lmm.rs:134-161): the state lives for
    one BLOCK of rows handed to the association stage (`block_starts`: the first row of every block), and rayon creates one
    state per piece its splitter cuts the block into -- recursive halving at len / 2, 2 T pieces on T idle threads without
    work stealing (more with it, which is what makes the reference's own output depend on scheduling).  `pieces` (a power of
    two, default 1 = one chain per block: the semantics the code states) reproduces that halving deterministically."""
    pieces = int(pieces)
    if pieces < 1 or (pieces & (pieces - 1)):
        raise RuntimeError("warm_chain_pieces must be a power of two >= 1")
    depth = pieces.bit_length() - 1
    m = int(m)
    starts = np.unique(np.clip(np.asarray(block_starts, dtype=np.int64), 0, m))
    if len(starts) == 0 or starts[0] != 0:
        starts = np.concatenate([[0], starts])
    bounds = np.concatenate([starts, [m]]).astype(np.int64)
    if depth == 0:
        return np.unique(bounds) if m > 0 else np.array([0, 0], dtype=np.int64)
    out = []
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        if b1 > b0:
            _halve(int(b0), int(b1), depth, out)
    out.append(m)
    return np.asarray(out, dtype=np.int64)


def warm_chain_blocks_packed(m: int, rotate_block_rows: int, progress_every: int = 0):
    """First rows of the association blocks of `lmm_reml_assoc_packed_f32` (src/stats/lmm.rs:3213-3253): segments of
    `progress_every` rows (`rotate_block_rows` when 0) cut into blocks of `rotate_block_rows`."""
    block = max(1, int(rotate_block_rows))
    seg = block if int(progress_every) == 0 else max(1, int(progress_every))
    starts = []
    for s0 in range(0, int(m), seg):
        starts.extend(range(s0, min(s0 + seg, int(m)), block))
    return np.asarray(starts, dtype=np.int64)


def warm_chain_blocks_bed(kept_source_rows, n_units: int, rotate_block_rows: int):
    """First kept row of every association block of the BED route (`run_unified_bed_scan_to_tsv_common`,
    src/stats/lmm.rs:1121-1145): the scan walks the `n_units` scan units (all SNP rows of the file, or the prepared row list) in
    chunks of `rotate_block_rows`; a block is the rows of one chunk that pass the filters.  `kept_source_rows`: ascending unit
    index of every kept row."""
    chunk = max(1, min(int(rotate_block_rows), max(1, int(n_units))))
    kept = np.asarray(kept_source_rows, dtype=np.int64)
    edges = np.arange(0, max(int(n_units), 1), chunk, dtype=np.int64)
    return np.searchsorted(kept, edges, side="left").astype(np.int64)


def deal_whole_chains(chain_off, world: int):
    """Row ranges [cuts[r], cuts[r + 1]) of `world` ranks over chained rows: every cut is a chain boundary -- the one nearest to
    the even share -- so that no chain starts without its predecessor's state and the ranks' tables concatenate to the one-rank
    table.  -> int64 array of world + 1 non-decreasing cuts from 0 to the number of rows (a rank's range may be empty)."""
    co = np.asarray(chain_off, dtype=np.int64)
    m = int(co[-1])
    cuts = [int(co[np.argmin(np.abs(co - (m * r) // int(world)))]) for r in range(int(world) + 1)]
    cuts[0], cuts[-1] = 0, m
    return np.maximum.accumulate(np.asarray(cuts, dtype=np.int64))
