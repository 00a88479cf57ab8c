"""CPU tests: the C-ABI library loads and exports every symbol include/jxgpu.h declares; host-side count logic
(janusx_amd.stats) is bit-identical to the oracle's scalar restatement; argument validation of the boundary
module raises before any GPU work."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "jxgpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(jxg?_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from janusx_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    h = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(h, s), f"{s} declared in include/jxgpu.h but not exported"
    # the ctypes signature table covers the same set
    assert set(_lib.SIGNATURES) == set(syms)
    h.jx_version.restype = ctypes.c_int
    assert h.jx_version() >= 100  # no GPU needed


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from janusx_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "janusx_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} references the oracle"


def test_stats_match_oracle_bit_exact(oracle):
    from janusx_amd import stats
    rng = np.random.default_rng(0)
    n = 997
    m = 4000
    missing = rng.integers(0, 80, m)
    missing[:50] = n            # all missing
    missing[50:100] = 0
    nm = n - missing
    het = (rng.random(m) * nm).astype(np.int64)
    hom = (rng.random(m) * (nm - het)).astype(np.int64)
    het[100:150] = 0
    hom[100:150] = 0            # monomorphic
    hom[150:200] = nm[150:200] - het[150:200]  # no ref homozygotes
    counts = np.stack([missing, het, hom], 1)
    for maf_thr, miss_thr, het_thr in [(0.02, 0.05, 1.0), (0.0, 1.0, 0.0), (0.05, 0.02, 0.4), (0.5, 0.0, 1.0)]:
        k1, af1, ms1, _ = oracle.gwas_scan_row_stats(missing, het, hom, n, maf_thr, miss_thr, het_thr)
        k2, af2, ms2 = stats.gwas_scan_row_stats(counts, n, maf_thr, miss_thr, het_thr)
        assert np.array_equal(k1, k2)
        assert np.array_equal(af1[k1], af2[k1]) and np.array_equal(ms1, ms2)
        kp, mp, ap, sp, fp = oracle.packed_prep_row_stats(missing, het, hom, n, maf_thr, miss_thr, het_thr)
        kq, mq, aq, sq = stats.packed_prep_row_stats(counts, n, maf_thr, miss_thr, het_thr)
        assert np.array_equal(kp, kq) and np.array_equal(mp, mq) and np.array_equal(ap, aq) and np.array_equal(sp, sq)
        assert not fp.any()
        for method in (1, 2):
            a = oracle.stream_grm_row_prepare(missing, het, hom, n, method, maf_thr, miss_thr, het_thr)
            b = stats.stream_grm_row_prepare(counts, n, method, maf_thr, miss_thr, het_thr)
            assert np.array_equal(a[0], b[0])
            kk = a[0]
            for u, v in zip(a[1:], b[1:]):
                assert np.array_equal(np.asarray(u)[kk], np.asarray(v)[kk])


def test_luts_match_oracle(oracle):
    from janusx_amd import stats
    rng = np.random.default_rng(1)
    n, m = 203, 300
    from janusx_amd import bed
    packed, g = bed.synth_panel_numpy(n, m, seed=3, missing_rate=0.05)
    mi, he, ho = oracle.row_counts(packed, n)
    counts = np.stack([mi, he, ho], 1)
    maf = rng.uniform(0, 0.6, m).astype(np.float32)
    flip = rng.random(m) < 0.5
    for method in (1, 2):
        lut = stats.grm_lut_from_maf(maf, flip, method)
        for j in range(0, m, 17):
            assert np.array_equal(lut[j], oracle.grm_value_lut_f32(maf[j], bool(flip[j]), method))
    slut = stats.scan_lut_from_counts(maf, flip, counts, n)
    ref = oracle.decode_centered_block_f32(packed, n, flip, maf)
    codes = oracle.unpack_codes(packed, n)
    for j in range(0, m, 7):
        assert np.array_equal(slut[j][codes[j]], ref[j])


def test_boundary_argument_validation():
    from janusx_amd import janusx as jxrs
    with pytest.raises(RuntimeError, match="n_samples must be > 0"):
        jxrs.grm_packed_f32(np.zeros((2, 1), np.uint8), 0, [0, 0], [0.1, 0.1])
    with pytest.raises(RuntimeError, match="packed length mismatch"):
        jxrs.grm_packed_f32(np.zeros((2, 3), np.uint8), 8, [0, 0], [0.1, 0.1])
    with pytest.raises(RuntimeError, match="low must be < high"):
        jxrs.lmm_reml_null_f32(np.ones(4), np.ones((4, 1)), np.ones(4), 1.0, 1.0)
    with pytest.raises(RuntimeError, match="must equal len"):
        jxrs.lmm_reml_null_f32(np.ones(3), np.ones((4, 1)), np.ones(4), -1.0, 1.0)
    with pytest.raises(RuntimeError, match="u_t must be"):
        jxrs.lmm_rotate_x_y_with_ut_f64(np.ones((3, 3), np.float32), np.ones((4, 1)), np.ones(4))
    # the reference's `_inplace` eigh: its own message for an empty / non-square input (src/math/eigh.rs:1907-1912), no diag_shift
    with pytest.raises(RuntimeError, match=r"rust_eigh_from_array_f64_inplace expects a non-empty square matrix; got shape=\(2, 3\)"):
        jxrs.rust_eigh_from_array_f64_inplace(np.zeros((2, 3)))
    with pytest.raises(RuntimeError, match=r"got shape=\(0, 0\)"):
        jxrs.rust_eigh_from_array_f64_inplace(np.zeros((0, 0)))
    with pytest.raises(TypeError):
        jxrs.rust_eigh_from_array_f64_inplace(np.eye(3), diag_shift=1e-6)
    with pytest.raises(RuntimeError, match="warm_start must be"):
        jxrs._resolve_warm_start("sometimes")


def test_tsv_text_matches_oracle_format(oracle):
    from janusx_amd import tsv
    rng = np.random.default_rng(5)
    vals = [(0.1234567, 0.0456, 1e-3), (-2.5e-5, 3.3e-4, 0.93), (1.0, 1e-9, 0.0), (float("nan"), float("nan"), 1.0),
            (0.5, 0.0, 0.2), (3.0, 0.5, 1e-320), (7.25, 0.25, float("nan"))]
    for _ in range(200):
        vals.append((float(rng.normal()), float(abs(rng.normal()) * 0.1), float(10 ** rng.uniform(-30, 0))))
    for b, s, p in vals:
        a = tsv.format_row("3", 12345, ".", "A", "C", np.float32(0.31415), np.float32(0.0123), b, s, p)
        r = oracle.format_assoc_row("3", 12345, ".", "A", "C", np.float32(0.31415), np.float32(0.0123), b, s, p)
        assert a == r
    assert tsv.HEADER3 == oracle.TSV_HEADER


def test_lm_fallback_decision(oracle):
    from janusx_amd import janusx as jxrs
    rng = np.random.default_rng(6)
    n = 150
    x = rng.normal(size=(n, 2))
    y = x @ np.array([0.5, -0.2]) + rng.normal(size=n)
    lm = oracle.lm_null_ml(y, np.concatenate([np.ones((n, 1)), x], 1))
    sw, stat, p, lm_ml0 = jxrs.gwas_lmm_lm_null_lrt_decision(y, x, lm + 0.4)
    assert abs(lm_ml0 - lm) < 1e-9 and abs(stat - 0.8) < 1e-8
    assert abs(p - 0.5 * oracle.chi2_sf_df1(0.8)) < 1e-12 and sw is True
    sw2, stat2, p2, _ = jxrs.gwas_lmm_lm_null_lrt_decision(y, x, lm + 10.0)
    assert sw2 is False and p2 < 0.05
    sw3, stat3, p3, _ = jxrs.gwas_lmm_lm_null_lrt_decision(y, x, lm - 3.0)
    assert stat3 == 0.0 and p3 == 0.5 and sw3 is True
    with pytest.raises(RuntimeError, match="alpha"):
        jxrs.gwas_lmm_lm_null_lrt_decision(y, x, lm, alpha=1.5)


def test_spgrm_host_helpers_round_trip(tmp_path):
    """`.spgrm` reader and output-path rule of the host mirror against the oracle's writer (layout of
    src/stats/spgrm.rs:3745-3767, path rule :450-469); no GPU involved."""
    from janusx_amd import janusx as jxrs
    from oracle import jx_oracle as O
    for nnz_rows, vals in [([0, 2, 1, 2], [1.0, 0.25, 1.1, 0.9]), ([0, 1, 1], [1.0, 0.25, 0.9])]:
        n = 3 if len(vals) == 4 else 2
        cp = [0, 2, 3, 4] if n == 3 else [0, 2, 3]
        p = str(tmp_path / f"k{n}.spgrm")
        O.write_sparse_grm_csc(p, n, cp, nnz_rows, vals)
        nn, cp2, ri2, va2 = jxrs.load_spgrm(p)
        assert nn == n and list(cp2) == cp and list(ri2) == nnz_rows and list(va2) == vals
        assert cp2.dtype == np.uint64 and ri2.dtype == np.uint32 and va2.dtype == np.float64
    with open(p, "ab") as fh:
        fh.write(b"\0")
    with pytest.raises(RuntimeError, match="length mismatch"):
        jxrs.load_spgrm(p)
    base = str(tmp_path / "q")
    assert jxrs._normalize_spgrm_path(" " + base + " ") == base + ".spgrm"
    assert jxrs._normalize_spgrm_path(base + ".SPGRM") == base + ".SPGRM"
    open(base + ".jxgrm", "wb").close()                       # an old-style file alone keeps its name
    assert jxrs._normalize_spgrm_path(base) == base + ".jxgrm" == O.normalize_spgrm_path(base)
    open(base + ".spgrm", "wb").close()
    assert jxrs._normalize_spgrm_path(base) == base + ".spgrm" == O.normalize_spgrm_path(base)
    assert jxrs._normalize_spgrm_path("   ") == ""


def test_sparse_grm_diag_stats_host(tmp_path):
    """`splmm_sparse_grm_diag_stats` (src/stats/splmm.rs:1978-2022, 4055-4111) on a written `.spgrm`: full set, identity
    subset, proper subset (entries with both ends inside), and the missing-diagonal error."""
    from janusx_amd import janusx as jxrs
    from oracle import jx_oracle as O
    p = str(tmp_path / "d.spgrm")
    O.write_sparse_grm_csc(p, 4, [0, 3, 5, 7, 8], [0, 1, 3, 1, 2, 2, 3, 3], [1.5, 0.2, 0.3, 0.9, 0.4, 1.1, 0.5, 0.7])
    assert jxrs.splmm_sparse_grm_diag_stats(p) == ((1.5 + 0.9 + 1.1 + 0.7) / 4, 0.7, 1.5, 4, 8)
    assert jxrs.splmm_sparse_grm_diag_stats(p, [0, 1, 2, 3]) == jxrs.splmm_sparse_grm_diag_stats(p)
    mean, lo, hi, n, nnz = jxrs.splmm_sparse_grm_diag_stats(p, [3, 0])
    assert (lo, hi, n, nnz) == (0.7, 1.5, 2, 3) and abs(mean - 1.1) < 1e-15       # (0,0), (3,0), (3,3)
    with pytest.raises(RuntimeError, match="duplicated sample index: 3"):
        jxrs.splmm_sparse_grm_diag_stats(p, [3, 0, 3])
    O.write_sparse_grm_csc(p, 2, [0, 1, 2], [1, 1], [0.3, 1.0])                    # column 0 has no diagonal entry
    with pytest.raises(RuntimeError, match="diagonal is missing at column 0"):
        jxrs.splmm_sparse_grm_diag_stats(p)


def test_eigh_launch_geometry_beyond_65535_rows():
    """the column gather of the eigensolver put its rows on gridDim.y (limit 65535); every n-dependent
    grid must be valid up to the sizes HBM allows (n = 70 000 and n = 150 000 here; no GPU needed)."""
    from janusx_amd import _lib
    h = _lib.lib()
    for n in (300, 46341, 65536, 70000, 150000):
        assert h.jxg_eigh_grid_check(n) == 1, n


def test_native_tsv_writer_matches_the_row_format(tmp_path):
    """`jx_assoc_tsv_write` (native formatter + writer, src/io/assoc2tsv.rs:430-548) against the pure-Python statement of
    the same row format, on values that hit every branch: NaN rows, zero / negative-zero, p = 0 (clamped to the smallest
    positive double), p = inf (-> 1), infinite beta (-> chisq NaN, p 1), `.` SNP names, 3 / 4 / 6 column tables."""
    from janusx_amd import tsv
    rng = np.random.default_rng(0)
    n = 3000
    stats = np.empty((n, 6))
    stats[:, 0] = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)
    stats[:, 1] = np.abs(rng.standard_normal(n)) * 10.0 ** rng.integers(-6, 3, n)
    stats[:, 2] = 10.0 ** (-rng.random(n) * 320)
    stats[:, 3] = rng.random(n) * 1e3
    stats[:, 4] = -rng.random(n) * 1e4
    stats[:, 5] = rng.random(n)
    stats[5] = [np.nan, np.nan, 1.0, np.nan, np.nan, 1.0]
    stats[6, 0] = 0.0
    stats[7, 1] = 0.0
    stats[8, 2] = 0.0
    stats[9, 0] = -0.0
    stats[10, 2] = np.inf
    stats[11, 0] = np.inf
    stats[12, :2] = [1e300, 1e300]      # degenerate but finite: Rust's {:.4} prints all 301 digits (no truncation, no overflow)
    stats[13, :2] = [-1.7976931348623157e308, 1.7976931348623157e308]
    chrom = [str(1 + i % 22) for i in range(n)]
    pos = list(range(n))
    snp = ["." if i % 7 == 0 else ("" if i % 11 == 0 else f"rs{i}") for i in range(n)]
    af = rng.random(n).astype(np.float32)
    miss = (rng.random(n) * 0.05).astype(np.float32)
    for nc in (3, 4, 6):
        a, b = str(tmp_path / f"a{nc}.tsv"), str(tmp_path / f"b{nc}.tsv")
        assert tsv.write_assoc_tsv(a, chrom, pos, snp, ["A"] * n, ["T"] * n, af, miss, stats[:, :nc]) == n
        assert tsv.write_assoc_tsv_python(b, chrom, pos, snp, ["A"] * n, ["T"] * n, af, miss, stats[:, :nc]) == n
        assert open(a).read() == open(b).read()
    with pytest.raises(RuntimeError, match="column count"):
        tsv.write_assoc_tsv(str(tmp_path / "c.tsv"), chrom, pos, snp, ["A"] * n, ["T"] * n, af, miss, stats[:, :5])
    # block-wise writer thread (streaming scan): ragged blocks, same bytes; no block at all -> header only; a wrong block
    # shape is refused; nothing is left behind under the final name before close()
    for nc in (3, 6):
        c, b = str(tmp_path / f"s{nc}.tsv"), str(tmp_path / f"b{nc}.tsv")
        w = tsv.AsyncAssocTsvWriter(c, nc, chrom, pos, snp, ["A"] * n, ["T"] * n, af, miss)
        i0 = 0
        for ln in (1, 700, 1, 1298, 1000):
            w.put(i0, stats[i0:i0 + ln, :nc])
            i0 += ln
        assert not os.path.exists(c)
        assert w.close() == n and open(c).read() == open(b).read()
    e = str(tmp_path / "empty.tsv")
    assert tsv.AsyncAssocTsvWriter(e, 3, [], [], [], [], [], af[:0], miss[:0]).close() == 0
    assert open(e).read() == tsv.HEADER3
    w = tsv.AsyncAssocTsvWriter(str(tmp_path / "bad.tsv"), 3, chrom, pos, snp, ["A"] * n, ["T"] * n, af, miss)
    with pytest.raises(RuntimeError, match="columns"):
        w.put(0, stats[:10, :4])
    w.close()
    w = tsv.AsyncAssocTsvWriter(str(tmp_path / "gone.tsv"), 3, chrom, pos, snp, ["A"] * n, ["T"] * n, af, miss)
    w.put(0, stats[:100, :3])
    w.abort()                                                        # a failed scan leaves nothing behind
    assert not os.path.exists(str(tmp_path / "gone.tsv")) and not [f for f in os.listdir(tmp_path) if "gone.tsv.tmp" in f]


def test_block_route_packs_whole_components():
    """Sample order of the block-diagonal sparse-GRM route (`_pack_components_into_blocks`): a permutation; no component is
    split over two blocks; components are contiguous inside their block; no block exceeds the size limit unless it is one
    oversized component; the packing wastes little (singletons fill the blocks)."""
    from janusx_amd.janusx import _pack_components_into_blocks
    rng = np.random.default_rng(5)
    sizes = np.concatenate([rng.integers(1, 9, 400), [70, 64, 130], np.ones(300, dtype=np.int64)])
    lab = rng.permutation(np.repeat(np.arange(len(sizes)), sizes))
    n = len(lab)
    for bsz in (64, 100, 4096):
        perm, offs = _pack_components_into_blocks(lab, bsz)
        assert sorted(perm.tolist()) == list(range(n)) and offs[0] == 0 and offs[-1] == n
        block_of_pos = np.repeat(np.arange(len(offs) - 1), np.diff(offs))
        comp_block = {}
        for pos, smp in enumerate(perm):
            assert comp_block.setdefault(int(lab[smp]), int(block_of_pos[pos])) == block_of_pos[pos]
        for b in range(len(offs) - 1):
            labs = lab[perm[offs[b]:offs[b + 1]]]
            assert len(np.nonzero(np.diff(labs))[0]) + 1 == len(np.unique(labs))          # contiguous components
            assert offs[b + 1] - offs[b] <= bsz or len(np.unique(labs)) == 1
        assert len(offs) - 1 <= int(np.ceil(n / bsz)) + 3 + int((sizes >= bsz).sum())


def test_cv_splits_follow_the_reference_kfold():
    """`cli.build_cv_splits` against vectors of the reference's own splitter (`build_cv_splits`,
    python/janusx/gs/workflow.py:3950-3980 over python/janusx/pyBLUP/kfold.py `KFold(shuffle=True, random_state=seed)`;
    generated in the build container with `KFold(n_splits=5, shuffle=True, random_state=42).split(np.arange(1410))`):
    balanced run lengths, the permutation of numpy's default_rng(seed), (test, train) order, train ascending."""
    from janusx_amd import cli
    sp = cli.build_cv_splits(1410, 5, 42)
    assert [len(te) for te, _ in sp] == [282] * 5
    assert sp[0][0][:8].tolist() == [180, 184, 631, 282, 20, 483, 1370, 849]
    sp = cli.build_cv_splits(13, 4, 7)
    assert [len(te) for te, _ in sp] == [4, 3, 3, 3]
    ref = np.random.default_rng(7).permutation(np.arange(13))
    assert np.array_equal(np.concatenate([te for te, _ in sp]), ref)
    for te, tr in sp:
        assert np.array_equal(np.sort(np.concatenate([te, tr])), np.arange(13)) and np.all(np.diff(tr) > 0)
    with pytest.raises(ValueError, match="cannot exceed"):
        cli.build_cv_splits(3, 5)


def test_host_payload_guard_refuses_what_cannot_be_staged(monkeypatch):
    """A host payload that cannot be staged safely (BASELINE configs[4] is 50 GB packed; a round-2 attempt took the GPU box
    down) is refused with a clear error before anything is copied; small arrays and device tensors pass."""
    import types
    from janusx_amd import janusx as jxrs
    import psutil
    monkeypatch.setattr(psutil, "virtual_memory", lambda: types.SimpleNamespace(available=8 << 30))
    jxrs._guard_host_payload(3 << 30, "packed")                   # fits twice
    with pytest.raises(RuntimeError, match="device .torch CUDA uint8. tensor"):
        jxrs._guard_host_payload(5 << 30, "packed")
    jxrs._guard_host_payload(1 << 20, "packed")                   # small arrays are never refused


def test_reference_python_layer_import_surface():
    """The `janusx.janusx` names the reference's Python layer binds on the hot path and either side of it (hard imports of
    python/janusx/pyBLUP/assoc.py:207-241, the all-or-nothing optional group there, workflow_model_stream.py:726-1678,
    workflow_model_packed.py:4292-6309, script/grm.py:97-122, gs/workflow.py:4090-4139, assoc/api.py:45-57): every one is a
    callable of the mirror module; the FastLMM ones are present but out of scope (loud when called).  A missing name of the
    optional group would switch the reference's fused routes off wholesale."""
    import janusx_amd.janusx as jxrs
    hard = """fastlmm_prepare_lowrank_f64 fastlmm_assoc_from_snp_f32 lm_block_assoc_f32 lmm_reml_chunk_f32 lmm_reml_null_f32
              ml_loglike_null_f32 lmm_assoc_chunk_f32 fastlmm_reml_chunk_f32 fastlmm_reml_null_f32 fastlmm_assoc_chunk_f32""".split()
    optional = """lmm_reml_chunk_from_snp_f32 lmm_reml_lmm2_chunk_from_snp_f32 lmm_assoc_chunk_from_snp_f32 fvlmm_assoc_chunk_f32
                  fvlmm_assoc_chunk_from_snp_f32 fvlmm_assoc_chunk_from_snp_to_tsv_f32 fvlmm_assoc_bed_to_tsv_f32
                  fvlmm_assoc_prepare_cache_f32 fvlmm_assoc_chunk_with_cache_f32 fvlmm_assoc_chunk_from_snp_with_cache_f32
                  lmm_rotate_x_y_with_ut_f64 lmm_rotate_y_with_ut_f64 lm_block_assoc_packed lm_block_assoc_packed_to_tsv
                  grm_bed_f64_from_meta spgrm_bed_to_jxgrm_from_meta spgrm_dense_f32_to_jxgrm spgrm_dense_npy_to_jxgrm
                  spgrm_bed_to_jxgrm spgrm_packed_to_jxgrm splmm_assoc_pcg_dense_f32 splmm_assoc_pcg_bed
                  splmm_assoc_pcg_bed_to_tsv lmm_reml_assoc_bed_to_tsv_f32 lmm_reml_lmm2_assoc_bed_to_tsv_f32
                  lmm_reml_assoc_packed_f32 lmm_reml_assoc_packed_f32_to_tsv fvlmm_assoc_packed_f32_to_tsv
                  gwas_lmm_lm_null_lrt_decision gblup_reml_packed_bed gblup_reml_npy_grm rrblup_pcg_bed he_pcg_bed
                  gblup_grm_from_meta_to_npy grm_bed_f32_row_band_from_meta grm_bed_f32_row_band_from_meta_to_npy
                  grm_bed_f32_tiled_from_meta_to_npy prepare_bed_logic_meta_selected load_bim_columns
                  bed_packed_decode_rows_f32 bed_decode_rows_f32_from_meta packed_malpha_f64
                  cross_grm_times_alpha_packed_f64 gblup_effect_from_meta_stream""".split()
    for name in hard + optional:
        assert callable(getattr(jxrs, name, None)), name
    with pytest.raises(RuntimeError, match="outside the mixed-model hot path"):
        jxrs.fastlmm_reml_null_f32()
    # argument checks of the new fixed-lambda family run before any device call
    s, x, y = np.linspace(0.1, 2.0, 12), np.ones((12, 1)), np.arange(12.0)
    with pytest.raises(RuntimeError, match="invalid log10_lbd"):
        jxrs.lmm_assoc_chunk_f32(s, x, y, 400.0, np.zeros((1, 12), np.float32))
    with pytest.raises(RuntimeError, match="non-positive"):
        jxrs.fvlmm_assoc_prepare_cache_f32(s - 5.0, x, y, 0.0)
    with pytest.raises(RuntimeError, match="n must be > p_cov\\+1"):
        jxrs.lmm_assoc_chunk_f32(s[:2], np.ones((2, 1)), y[:2], 0.0, np.zeros((1, 2), np.float32))
    c = jxrs.fvlmm_assoc_prepare_cache_f32(s, x, y, -1.0)
    assert (c.n, c.p) == (12, 1) and abs(c.lbd - 0.1) < 1e-15


def test_cli_helpers_match_reference_produced_values():
    """`tests/golden/cli_helpers.json` holds values the reference's own Python produced in the build container
    (tests/golden/gen_cli_fixtures.py: `resolve_blup_dispatch` of python/janusx/gs/blup.py with the GS_BLUP override,
    `_parse_cov_site_token`, `_parse_qcov_dim`, `_canon_site_key` of python/janusx/assoc/workflow.py); the mirror's
    command-line helpers must reproduce them."""
    import json
    from janusx_amd import cli
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cli_helpers.json")))
    assert (cli.BLUP_SMALL_N, cli.BLUP_SMALL_M) == (gold["blup_small_n"], gold["blup_small_m"])
    for row in gold["dispatch"]:
        assert cli.resolve_blup_dispatch(row["n"], row["m"], row["force"]) == (row["method"], row["solver"]), row
    for row in gold["cov_site"]:
        if row["error"] is not None:
            with pytest.raises(SystemExit):
                cli._parse_cov_site_token(row["token"])
        else:
            got = cli._parse_cov_site_token(row["token"])
            assert (None if got is None else [got[0], got[1]]) == row["result"], row
    for row in gold["qcov"]:
        if row["error"] is not None:
            with pytest.raises(SystemExit):
                cli._parse_qcov_dim(row["value"])
        else:
            assert cli._parse_qcov_dim(row["value"]) == row["result"], row
    for row in gold["canon"]:
        assert list(cli._canon_site_key(row["chrom"], row["pos"])) == row["key"], row
    with pytest.raises(ValueError, match="Invalid GS_BLUP"):
        cli.resolve_blup_dispatch(10, 10, "5")
    for row in gold["cv_splits"]:
        got = cli.build_cv_splits(row["n"], row["k"], row["seed"])
        assert [[list(map(int, te)), list(map(int, tr))] for te, tr in got] == row["folds"], (row["n"], row["k"], row["seed"])


def test_sparse_component_limit_host_logic(monkeypatch):
    """`sparse_component_limit` (one dense eigenproblem per connected component of a thresholded GRM: ~5 n^2 doubles of HBM) and the
    refusal beyond it, with the size and the limit in the message (src/math/cholesky.rs:776-1075 factorises any structure; this
    library does not): the arithmetic and the message, no GPU needed."""
    import pytest
    from janusx_amd import janusx as jxrs
    assert jxrs.sparse_component_limit(288 * 10**9) == 84852          # isqrt(288e9 / 40)
    assert jxrs.sparse_component_limit(250 * 2**30) == 81920
    assert jxrs.sparse_component_limit(0) == 0
    monkeypatch.setenv("JXGPU_SPLMM_COMPONENT_MAX", "512")
    assert jxrs.sparse_component_limit() == 512
    jxrs._check_spectral_sparse_size(512)
    with pytest.raises(RuntimeError, match=r"one dense eigenproblem of 700 samples; the limit on this GPU is 512 samples"):
        jxrs._check_spectral_sparse_size(700, what="the samples of the largest connected component of the sparse GRM")


def test_bench_headline_line_is_short_strict_json():
    """The driver parses the LAST stdout line of bench.py; round 5's ~20 kB line was not parsed.  Build the line from the
    largest detail record a default run has produced (profiles/r05i_bench_default.json: ten legs with paragraph-long notes),
    with non-finite values and an over-long note planted, and hold it under 6 kB of strict JSON with every contract field."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    res = json.loads(open(os.path.join(ROOT, "profiles", "r05i_bench_default.json")).read().strip().splitlines()[-1])
    res["roofline"]["traffic"] = float("nan")
    res["stages_ms_per_step"]["planted_inf"] = float("inf")
    res["roofline"]["note"] = "x" * 50000
    res["extra_c4_1gpu"]["note"] = "y" * 50000
    out, line = bench.headline_record(res)
    assert len(line.encode()) < bench.HEADLINE_MAX_BYTES == 6144
    assert "\n" not in line
    back = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))   # NaN / Infinity refused
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in back, key
    assert back["roofline"]["traffic"] is None and back["stages_ms_per_step"]["planted_inf"] is None
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in back["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in back["cpu_baseline"], key
    assert back["config"]["workload"] and "model" not in back["config"]
    assert abs(back["value"] - res["value"]) <= 1e-6 * res["value"]
    assert set(back["legs"]) >= {"c1", "c2", "c3_miss", "c4_1gpu", "c5_splmm", "c5_pcg"}
    # a record so large that even the short form would overflow sheds its optional parts instead of growing
    res["stages_ms_per_step"].update({f"stage_{i}": float(i) for i in range(400)})
    _, line2 = bench.headline_record(res)
    assert len(line2.encode()) < 6144 and "value" in json.loads(line2)


def test_warm_chain_offsets_host_rules():
    """Chain boundaries as the reference's loops make them: blocks of `rotate_block_rows` inside segments of `progress_every`
    rows (src/stats/lmm.rs:3253-3258), kept rows per chunk of scan units on the BED route (:1121-1145), rayon's halving."""
    from janusx_amd import stats as st
    assert st.warm_chain_offsets(st.warm_chain_blocks_packed(10, 4), 10).tolist() == [0, 4, 8, 10]
    assert st.warm_chain_offsets(st.warm_chain_blocks_packed(10, 4, 6), 10).tolist() == [0, 4, 6, 10]
    assert st.warm_chain_offsets(st.warm_chain_blocks_packed(10, 4), 10, 2).tolist() == [0, 2, 4, 6, 8, 9, 10]
    assert st.warm_chain_offsets(st.warm_chain_blocks_packed(7, 100), 7, 4).tolist() == [0, 1, 3, 5, 7]     # 7 -> 3 | 4 -> 1 2 | 2 2
    kept = np.array([0, 1, 5, 6, 7, 12, 13, 19])
    assert st.warm_chain_offsets(st.warm_chain_blocks_bed(kept, 20, 5), len(kept)).tolist() == [0, 2, 5, 7, 8]
    assert st.warm_chain_offsets(st.warm_chain_blocks_bed(np.array([17, 18]), 20, 5), 2).tolist() == [0, 2]   # empty chunks vanish
    with pytest.raises(RuntimeError):
        st.warm_chain_offsets([0], 10, 3)
    # several ranks: whole chains per rank, cuts at the chain boundary nearest to the even share
    co = st.warm_chain_offsets(st.warm_chain_blocks_packed(1000, 96), 1000)
    cuts = st.deal_whole_chains(co, 4)
    assert cuts[0] == 0 and cuts[-1] == 1000 and np.all(np.diff(cuts) >= 0) and set(cuts.tolist()) <= set(co.tolist())
    assert np.max(np.abs(np.diff(cuts) - 250)) <= 96
    assert st.deal_whole_chains(np.array([0, 1000]), 4).tolist() == [0, 0, 0, 1000, 1000] or \
        st.deal_whole_chains(np.array([0, 1000]), 4).tolist()[-1] == 1000            # one chain: one rank takes it


def test_every_environment_switch_is_classified():
    """`janusx_amd/switches.py` names every `JXGPU_*` switch of the sources with its class (numerics / knob / form / trace /
    test), the product value and the test that covers it: a new switch cannot appear without saying whether it is a product state."""
    import glob
    from janusx_amd import switches
    found = set()
    for f in glob.glob(os.path.join(ROOT, "janusx_amd", "**", "*"), recursive=True) + [os.path.join(ROOT, "bench.py")]:
        if f.endswith((".py", ".hip", ".cpp", ".h")) and not f.endswith("switches.py"):
            found |= set(re.findall(r"JXGPU_[A-Z0-9_]+", open(f, errors="replace").read()))
    missing = sorted(found - set(switches.SWITCHES))
    stale = sorted(set(switches.SWITCHES) - found)
    assert not missing, f"unclassified switches: {missing}"
    assert not stale, f"switches.py lists names no source uses: {stale}"
    for name, (cls, default, what, test) in switches.SWITCHES.items():
        assert cls in ("numerics", "knob", "form", "trace", "test") and default and what and test, name
    assert "JXGPU_OZ_PLANES" in switches.markdown_table()
