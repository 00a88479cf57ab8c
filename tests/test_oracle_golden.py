"""CPU tests: the oracle against (a) the reference's own unit-test vectors, (b) values produced by the
reference's pure-numpy helpers (stored in tests/golden/panel_small.npz by gen_fixtures.py), (c) itself
(numpy restatement vs C restatement), and the committed fixture as a regression pin."""
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "panel_small.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_pack_codes_layout(oracle):
    # src/math/bedmath.rs:1528-1534: sample i at bits 2*(i&3) of byte i>>2
    codes = np.array([[0, 2, 3, 1, 0, 3, 2, 0]], dtype=np.uint8)
    p = oracle.pack_codes(codes)
    assert p.tolist() == [[0 | (2 << 2) | (3 << 4) | (1 << 6), 0 | (3 << 2) | (2 << 4) | (0 << 6)]]
    assert np.array_equal(oracle.unpack_codes(p, 8), codes)


def test_reference_decode_vectors(oracle):
    # near_full_additive_decode_matches_gather_decode (bedmath.rs:1537-1573): LUT [0, 2*maf, 1, 2]
    codes = np.array([0, 2, 3, 1, 0, 3, 2, 0], dtype=np.uint8)
    idx = [0, 1, 2, 3, 5, 6, 7]
    lut = oracle.scan_value_lut_f32(np.float32(0.25), False)
    assert lut.tolist() == [0.0, 0.5, 1.0, 2.0]
    assert lut[codes[idx]].tolist() == [0.0, 1.0, 2.0, 0.5, 2.0, 1.0, 0.0]
    # near_full_standardized (bedmath.rs:1576-1628): flipped LUT with mean .6 and inv_sd 1.25
    from janusx_amd import stats
    l2 = stats.grm_lut_from_mean_scale(np.float32([0.6]), np.float32([1.25]), [True])[0]
    exp = np.float32([(np.float32(2.0) - np.float32(0.6)) * np.float32(1.25), 0.0,
                      (np.float32(1.0) - np.float32(0.6)) * np.float32(1.25),
                      (np.float32(0.0) - np.float32(0.6)) * np.float32(1.25)])
    assert np.array_equal(l2, exp)
    # centered_subset_decode_uses_global_mean_for_method1 (bedmath.rs:1630-1660)
    codes = np.array([[0, 0, 0, 2, 2, 2, 2, 2]], dtype=np.uint8)
    packed = oracle.pack_codes(codes)
    z = oracle.decode_grm_block_f32(packed, 8, [False], np.float32([0.625]), [0, 1, 2], 1, 0, 1)
    assert np.allclose(z, [[-1.25, -1.25, -1.25]], atol=1e-6)
    v = oracle.grm_varsum(np.float32([0.625]), 1, False)
    assert abs(v - 2.0 * 0.625 * (1 - 0.625)) < 1e-9


def test_reference_scalar_vectors(oracle):
    # src/math/linalg.rs:374-380: chi2 inverse sf of 1.8885e-19 is ~81.8
    assert 0.5 < oracle.chi2_sf_df1(81.8) / 1.8885e-19 < 2.0
    assert oracle.chi2_sf_df1(float("nan")) == 1.0 and oracle.chi2_sf_df1(0.0) == 1.0
    assert abs(oracle.normal_sf(0.0) - 0.5) < 1e-16
    # src/math/eigh.rs:1982-1998
    s, u = oracle.eigh_sym(np.array([[2.0, 1.0], [1.0, 2.0]]))
    assert np.allclose(s, [1.0, 3.0], atol=1e-9) and np.allclose(u.T @ u, np.eye(2), atol=1e-9)


def test_brent_known_function(oracle):
    x, fx, ev = oracle.brent_minimize(lambda t: (t - 1.234) ** 2 + 3.0, -5.0, 5.0, 1e-8, 100)
    assert abs(x - 1.234) < 1e-6 and abs(fx - 3.0) < 1e-10
    # max_iter bound: evaluations = 1 + iterations
    _, _, ev = oracle.brent_minimize(lambda t: abs(t - 0.3), -5.0, 5.0, 1e-12, 7)
    assert ev == 8
    # init outside the bracket falls back to the midpoint
    x0, _, _ = oracle.brent_minimize(lambda t: (t - 1.0) ** 2, 0.0, 4.0, 1e-3, 0, init_x=9.0)
    assert x0 == 2.0


def test_reference_nullreml_values(oracle, oracle_c, gold):
    """REML formula pinned to values computed by the reference's own `LMM._NULLREML` (assoc.py:1917)."""
    for lam, val in zip(gold["ref_lams"], gold["ref_nullreml"]):
        for fn in (oracle.reml_loglike, oracle_c.reml_loglike):
            mine = fn(math.log10(lam), gold["S"], gold["Xcov"], gold["yrot"], None)
            assert abs(mine - val) < 1e-6 * max(1.0, abs(val))
    assert "LMM._NULLREML" in str(gold["reference_checked"])


def test_reference_python_values_round3(oracle, gold):
    """Values produced by the reference's OWN Python in the build container (tests/golden/gen_fixtures.py, stored in the
    fixture): the dense-Cholesky restricted likelihood `REML` (python/janusx/pyBLUP/blup.py:158-236), the spectral GBLUP
    likelihood `BLUP._REML` incl. its v_floor branch (pyBLUP/mlm.py:1940-2050), `_lm_plrt_from_beta_se` and
    `_lm_precompute_ixx_qr` (pyBLUP/assoc.py:74-100, 453-480).  The oracle must reproduce every one of them."""
    import math
    assert all(t in str(gold["reference_checked"]) for t in ("blup.REML", "mlm.BLUP._REML", "_lm_plrt_from_beta_se",
                                                             "_lm_precompute_ixx_qr"))
    for lam, val in zip(gold["ref_lams"], gold["ref_dense_reml"]):
        # blup.REML returns the objective to MINIMISE; the spectral form carries a 1e-6 ridge on X'V^-1X (reml.rs:311)
        mine = -oracle.reml_loglike(math.log10(lam), gold["S"], gold["Xcov"], gold["yrot"], None)
        assert abs(val - mine) < 1e-6 * max(1.0, abs(val)), (lam, val, mine)
    n = int(gold["n"])
    for lam, val in zip(gold["ref_lams"], gold["ref_gblup_reml"]):
        mine = oracle.gblup_reml_eval(gold["gblup_s"], gold["gblup_xrot"], gold["gblup_yrot"], n, math.log10(lam))[0]
        assert abs(val - mine) < 1e-10 * max(1.0, abs(val)), (lam, val, mine)
    mine = oracle.gblup_reml_eval(gold["gblup_s_floor"], gold["gblup_xrot"], gold["gblup_yrot"], n, math.log10(0.3))[0]
    assert abs(float(gold["ref_gblup_reml_floor"]) - mine) < 1e-10 * abs(mine)
    x = gold["x"]
    ixx = oracle.lm_precompute_ixx_qr(x)
    assert np.max(np.abs(ixx - gold["ref_lm_ixx"])) < 1e-14 * np.max(np.abs(ixx))
    ixd = oracle.lm_precompute_ixx_qr(gold["lm_x_deficient"])
    assert np.max(np.abs(ixd - gold["ref_lm_ixx_deficient"])) < 1e-12 * np.max(np.abs(ixd))
    pk, maf = gold["lm_pk"], gold["lm_maf"]
    out = oracle.lm_block_assoc_packed(gold["y"], x, ixx, pk, n, np.zeros(len(maf), bool), maf)
    ok = np.isfinite(gold["ref_lm_plrt"])
    assert np.array_equal(ok, np.isfinite(out[:, 3]))
    assert np.max(np.abs(out[ok, 3] - gold["ref_lm_plrt"][ok]) / gold["ref_lm_plrt"][ok]) < 1e-12


def test_python_vs_c_oracle(oracle, oracle_c, gold):
    s, x, y, grot = gold["S"], gold["Xcov"], gold["yrot"], gold["grot"]
    for t in (-2.0, 0.0, 1.5):
        snp = grot[7].astype(np.float64)
        assert abs(oracle.reml_loglike(t, s, x, y, snp) - oracle_c.reml_loglike(t, s, x, y, snp)) < 1e-9
        assert abs(oracle.ml_loglike(t, s, x, y, snp) - oracle_c.ml_loglike(t, s, x, y, snp)) < 1e-9
        b1 = oracle.final_beta_se(t, s, x, y, snp)
        b2 = oracle_c.final_beta_se(t, s, x, y, snp)
        assert np.allclose(b1, b2, rtol=1e-10)
    lo, hi = gold["bounds"]
    a = oracle.lmm_scan_rotated_block(grot[:40], s, x, y, lo, hi, 30, 1e-2)
    b = oracle_c.lmm_scan_rotated_block(grot[:40], s, x, y, lo, hi, 30, 1e-2)
    assert np.allclose(a, b, rtol=1e-7, equal_nan=True)
    n1 = oracle.lmm_reml_null(s, x, y, -5, 5, 50, 1e-3)
    n2 = oracle_c.lmm_reml_null(s, x, y, -5, 5, 50, 1e-3)
    assert np.allclose(n1, n2, rtol=1e-9)


def test_fixture_regression(oracle, oracle_c, gold):
    n = int(gold["n"])
    packed = gold["packed"]
    mi, he, ho = oracle.row_counts(packed, n)
    assert np.array_equal(np.stack([mi, he, ho], 1), gold["counts"])
    c2 = oracle_c.row_counts(packed, n)
    assert np.array_equal(np.stack(c2, 1), gold["counts"])
    keep, af, miss, flip = oracle.gwas_scan_row_stats(mi, he, ho, n, 0.02, 0.05, 1.0)
    assert np.array_equal(keep, gold["keep"]) and np.array_equal(af, gold["af"]) and np.array_equal(miss, gold["miss"])
    k1, eff, _ = oracle.grm_stream_bed(packed, n, 1, 0.02, 0.05, 0.0)
    assert eff == gold["eff_m"][0] and np.allclose(k1, gold["k_stream_m1"], rtol=0, atol=1e-6)
    s, u = oracle.gwas_eigh_from_grm(k1)
    assert np.allclose(s, gold["S"], rtol=0, atol=1e-9)
    nm = oracle.spectral_null_model(gold["y"], gold["x"], s, u)
    assert abs(nm.lbd_null - gold["lbd"]) < 1e-6 * gold["lbd"]
    assert abs(nm.ML0 - gold["ml0"]) < 1e-7 * abs(gold["ml0"]) and abs(nm.pve - gold["pve"]) < 1e-6
    lo, hi = gold["bounds"]
    out = oracle_c.lmm_scan_rotated_block(gold["grot"], gold["S"], gold["Xcov"], gold["yrot"], lo, hi, 30, 1e-2)
    assert np.allclose(out, gold["lmm"], rtol=1e-9, equal_nan=True)
    fv = oracle.fvlmm_assoc_rotated_block(gold["grot"], oracle.fvlmm_prepare_cache(gold["S"], gold["Xcov"],
                                                                                  gold["yrot"], float(gold["lbd"])))
    assert np.allclose(fv, gold["fvlmm"], rtol=1e-9, equal_nan=True)


def test_edge_rows(oracle, gold):
    """all-missing / monomorphic / all-het rows: filters and the (NaN, NaN, 1) convention."""
    n = int(gold["n"])
    keep = gold["keep"]
    assert not keep[0] and not keep[1] and not keep[2]      # all missing, monomorphic ref/alt fail maf
    assert keep[3]                                          # all het: maf 0.5 passes with het_thr 1.0
    assert not keep[4]                                      # 50 % missing fails geno 0.05
    gk = gold["gkeep"]
    assert gk[5] and gold["gflip"][5]                       # alt_freq > 0.5 flips in the stream GRM
    # a constant design row is centred to zero -> (NaN, NaN, 1.0) (src/stats/lmm.rs:74-81)
    z = np.zeros((1, n), dtype=np.float32)
    out = oracle.lmm_scan_rotated_block(z, gold["S"], gold["Xcov"], gold["yrot"], -5, 5, 30, 1e-2)
    assert math.isnan(out[0, 0]) and math.isnan(out[0, 1]) and out[0, 2] == 1.0


def test_tsv_format(oracle, gold):
    txt = str(gold["tsv"])
    lines = txt.splitlines()
    assert lines[0] == "chrom\tpos\tsnp\tallele0\tallele1\taf\tmiss\tbeta\tse\tchisq\tpwald"
    assert oracle.rust_fmt_e4(1.0) == "1.0000e0" and oracle.rust_fmt_e4(1.2345e-3) == "1.2345e-3"
    assert oracle.rust_fmt_e4(float("nan")) == "NaN" and oracle.rust_fmt_e4(float("inf")) == "inf"
    row = oracle.format_assoc_row("2", 77, ".", "A", "T", np.float32(0.25), np.float32(0.0), float("nan"), float("nan"), 1.0)
    assert row == "2\t77\t2_77\tA\tT\t0.2500\t0.0000\tNaN\tNaN\tNaN\t1.0000e0\n"
    assert len(lines) == 13 and all(len(l.split("\t")) == 11 for l in lines)


MOUSE = os.path.join(os.path.dirname(__file__), "golden", "mouse_hs1940.npz")


def test_mouse_effective_snp_count_matches_reference_readme(oracle, oracle_c):
    """The reference README's demo output reports `EffSNPs: 8960` for example/mouse_hs1940 at the default filters
    (maf 0.02, geno 0.05; README.md:115): a number produced by the reference itself, reproduced by the QC restatement
    (scan rule src/stats/lmm.rs:1258-1320 and stream-GRM rule src/stats/grm.rs:1465-1536)."""
    from janusx_amd import stats
    d = np.load(MOUSE)
    packed, n = d["packed"], len(d["ids"])
    assert n == 1940 and packed.shape[0] == 10300
    mi, he, ho = oracle_c.row_counts(packed, n)
    keep, af, miss, flip = oracle.gwas_scan_row_stats(mi, he, ho, n, 0.02, 0.05, 1.0)
    assert int(keep.sum()) == 8960
    gk = oracle.stream_grm_row_prepare(mi, he, ho, n, 1, 0.02, 0.05, 0.0)[0]
    assert int(gk.sum()) == 8960
    k2, af2, miss2 = stats.gwas_scan_row_stats(np.stack([mi, he, ho], 1), n, 0.02, 0.05, 1.0)
    assert np.array_equal(k2, keep) and np.array_equal(af2[keep], af[keep])
    assert int(np.isfinite(d["pheno"][:, 0]).sum()) == 1410  # README "Train size: 1410" (trait test0)


def test_rrblup_pcg_restatement_solves_the_ridge_system():
    """the reference ships no numeric test for `rrblup_pcg_bed` / `pcg_solve_into` (parity unpinned beyond
    this property): the restated f32 PCG must converge to the f64 direct solution of
    (Z_c Z_c' + lambda I) beta = Z y_c, and its predictions must be alpha + Z' beta."""
    from oracle import jx_oracle as O
    from janusx_amd import bed
    n, m = 200, 500
    packed, g = bed.synth_panel_numpy(n, m, seed=4, missing_rate=0.02)
    miss, maf, std, flip = O.load_bed_2bit_packed_stats(packed, n)
    rng = np.random.default_rng(0)
    tr = np.sort(rng.permutation(n)[:160])
    te = np.setdiff1d(np.arange(n), tr)
    y = rng.standard_normal(160) + (g[:20, tr].T @ rng.standard_normal(20)) * 0.3
    lam = 120.0
    out = O.rrblup_pcg_packed(packed, n, maf, flip, tr, y, te, lambda_value=lam, tol=1e-7, max_iter=500)
    assert out[3] and out[5] <= 1e-7
    rm, ri, me = O.rrblup_row_standardization(maf, np.float32(1e-12))
    assert me == out[6]
    lut = O.rrblup_value_lut(rm, ri, flip)
    codes = O.unpack_codes(packed, n).astype(np.int64)
    z = np.take_along_axis(lut, codes[:, tr], axis=1).astype(np.float64)
    zc = z - z.mean(1, keepdims=True)
    yc = y - y.mean()
    beta = np.linalg.solve(zc @ zc.T + lam * np.eye(m), z @ yc)
    assert np.max(np.abs(beta - out[9])) <= 2e-5 * np.max(np.abs(beta))
    alpha = y.mean() - float(np.sum(z.mean(1) * beta))
    zt = np.take_along_axis(lut, codes[:, te], axis=1).astype(np.float64)
    assert np.max(np.abs(out[1].ravel() - (zt.T @ beta + alpha))) <= 1e-4
    assert np.max(np.abs(out[0].ravel() - (z.T @ beta + alpha))) <= 1e-4


def test_rrblup_exact_restatement_is_the_ridge_solution_at_its_reml_optimum():
    """SURVEY 8f-4, exact marker-space route (src/stats/rrblup.rs:3179-3490): no numeric test in the reference (parity
    unpinned beyond these properties).  The restated route must (i) return the f64 direct solution of
    (Z_c Z_c' + lambda I) beta = Z y_c at ITS lambda, (ii) sit at a minimum of the restricted likelihood of the equivalent
    sample-space model y_c ~ N(0, sigma^2 (Z_c'Z_c / lambda + I)) evaluated densely, (iii) cap the rank at n_train - 1
    when there are more markers than samples."""
    from oracle import jx_oracle as O
    from janusx_amd import bed
    for n, m, ntr in ((160, 380, 120), (220, 70, 180)):
        packed, g = bed.synth_panel_numpy(n, m, seed=n, missing_rate=0.02)
        miss, maf, std, flip = O.load_bed_2bit_packed_stats(packed, n)
        rng = np.random.default_rng(1)
        tr = np.sort(rng.permutation(n)[:ntr])
        te = np.setdiff1d(np.arange(n), tr)
        y = rng.standard_normal(ntr) + (g[:20, tr].T @ rng.standard_normal(20)) * 0.4
        out = O.rrblup_exact_snp_packed(packed, n, tr, y, te, maf=maf, row_flip=flip, reml_tol=1e-8, reml_max_iter=200)
        lam = out[3]
        rm, ri, me = O.rrblup_row_standardization(maf, np.float32(1e-12))
        lut = O.rrblup_value_lut(rm, ri, flip)
        codes = O.unpack_codes(packed, n).astype(np.int64)
        z = np.take_along_axis(lut, codes[:, tr], axis=1).astype(np.float64)
        zc = z - z.mean(1, keepdims=True)
        yc = y - y.mean()
        beta = np.linalg.solve(zc @ zc.T + lam * np.eye(m), z @ yc)
        assert np.max(np.abs(beta - out[8])) <= 2e-6 * np.max(np.abs(beta))
        alpha = y.mean() - float(np.sum(z.mean(1) * beta))
        zt = np.take_along_axis(lut, codes[:, te], axis=1).astype(np.float64)
        assert np.max(np.abs(out[1].ravel() - (zt.T @ beta + alpha))) <= 1e-5
        # dense restricted likelihood (intercept projected out: n_train - 1 degrees of freedom) as a function of log10 lambda
        k = zc.T @ zc
        ev, u = np.linalg.eigh(k)            # the constant vector is a null direction of k and orthogonal to y_c
        ev = np.clip(ev, 0.0, None)
        yp = u.T @ yc

        def cost(l10):
            lamv = 10.0 ** l10
            v = ev / lamv + 1.0
            q = float(np.sum(yp * yp / v))
            return 0.5 * ((ntr - 1) * math.log(q) + float(np.sum(np.log(v))))
        l0 = math.log10(lam)
        assert cost(l0) <= cost(l0 - 0.05) + 1e-9
        assert l0 > 5.999 or cost(l0) <= cost(l0 + 0.05) + 1e-9        # an optimum on the upper bound is one-sided
        assert len(out[8]) == m and out[6] == me


def test_he_restatement_matches_dense_traces():
    """SURVEY 8f-4 (HE half): no numeric test in the reference (parity unpinned beyond this): with exact traces the
    restated estimator must reproduce tr(PKP), tr((PKP)^2), y'PKPy of the dense f64 standardised GRM, and its probes
    are the reference's splitmix64 stream (first values pinned from the published splitmix64 test vector)."""
    from oracle import jx_oracle as O
    from janusx_amd import bed
    assert O.splitmix64(0) == 0xE220A8397B1DCDAF and O.splitmix64(0xE220A8397B1DCDAF) != 0
    n, m = 150, 400
    packed, g = bed.synth_panel_numpy(n, m, seed=12, missing_rate=0.02)
    miss, maf, std, flip = O.load_bed_2bit_packed_stats(packed, n)
    rng = np.random.default_rng(3)
    tr = np.sort(rng.permutation(n)[:120])
    y = rng.standard_normal(120)
    out = O.he_pcg_packed(packed, n, maf, flip, tr, y, exact_trace_debug=True, exact_trace_max_n=256)
    mean, inv, me = O.he_row_standardization(packed, n, flip, maf, tr, np.float32(1e-12), True)
    lut = O.rrblup_value_lut(mean, inv, flip)
    codes = O.unpack_codes(packed, n)[:, tr].astype(np.int64)
    z = np.take_along_axis(lut, codes, axis=1).astype(np.float64)
    k = z.T @ z / float(me)
    pm = np.eye(120) - np.ones((120, 120)) / 120.0
    pkp = pm @ k @ pm
    assert abs(out[12] - np.trace(pkp)) <= 2e-5 * np.trace(pkp)
    assert abs(out[7] - np.sum(pkp * pkp)) <= 5e-5 * np.sum(pkp * pkp)
    yp = pm @ y
    assert abs(out[8] - yp @ k @ yp) <= 2e-5 * abs(yp @ k @ yp)
    assert abs(out[9] - yp @ yp) <= 1e-6 * (yp @ yp)
    sg, se = np.linalg.solve(np.array([[np.sum(pkp * pkp), np.trace(pkp)], [np.trace(pkp), 119.0]]),
                             np.array([yp @ k @ yp, yp @ yp]))
    if sg >= 0 and se >= 0:
        assert abs(out[0] - sg) <= 1e-3 * (abs(sg) + abs(se)) and abs(out[1] - se) <= 1e-3 * (abs(sg) + abs(se))


def test_sparse_grm_restatement_keep_rule_layout_and_dense_equivalence(tmp_path):
    """Sparse GRM (next row 8f-3).  Pins: the keep rule against the reference's unit tests
    (src/stats/spgrm.rs:6618-6630), the `.spgrm` layout against its writer tests (:6125-6183: header words, total
    length, 4 zero bytes after an odd number of row indices), and the CSC content against the dense GRM restatement
    (itself pinned above): every stored entry equals the dense f64 K, every dropped one fails the rule.
    The reference's toy test :6067-6122 is not usable as a vector: its hand-written expectation centres by the data
    mean and keeps a negative entry at cut-off 0.2, which contradicts `spgrm_keep_value` and the tests at :6618."""
    from oracle import jx_oracle as O
    assert O.spgrm_keep_value(-1.5, -0.1, False) and O.spgrm_keep_value(0.0, -0.1, False)
    assert O.spgrm_keep_value(2.0, -0.1, False)
    assert not O.spgrm_keep_value(-1e-6, 0.0, False) and not O.spgrm_keep_value(0.0, 0.0, False)
    assert O.spgrm_keep_value(1e-6, 0.0, False)
    assert O.spgrm_keep_value(-0.3, 0.2, True) and not O.spgrm_keep_value(-0.3, 0.2, False)
    # writer: n = 3, nnz = 4 (no padding) and n = 2, nnz = 3 (4 bytes of padding)
    p = str(tmp_path / "a.spgrm")
    O.write_sparse_grm_csc(p, 3, [0, 2, 3, 4], [0, 2, 1, 2], [1.0, 0.25, 1.1, 0.9])
    raw = open(p, "rb").read()
    assert len(raw) == 16 + 4 * 8 + 4 * 4 + 0 + 4 * 8
    assert int.from_bytes(raw[0:8], "little") == 3 and int.from_bytes(raw[8:16], "little") == 4
    O.write_sparse_grm_csc(p, 2, [0, 2, 3], [0, 1, 1], [1.0, 0.25, 0.9])
    raw = open(p, "rb").read()
    row_end = 16 + 3 * 8 + 3 * 4
    assert row_end % 8 == 4 and raw[row_end:row_end + 4] == b"\0" * 4 and len(raw) == row_end + 4 + 3 * 8
    n, cp, ri, va = O.read_sparse_grm_csc(p)
    assert n == 2 and list(cp) == [0, 2, 3] and list(ri) == [0, 1, 1] and list(va) == [1.0, 0.25, 0.9]
    assert O.normalize_spgrm_path(" x ") == "x.spgrm" and O.normalize_spgrm_path("y.SPGRM") == "y.SPGRM"
    assert O.normalize_spgrm_path("z.jxgrm") == "z.jxgrm" and O.normalize_spgrm_path("  ") == ""
    # dense equivalence
    rng = np.random.default_rng(3)
    nn, m = 37, 260
    g = rng.integers(0, 3, (m, nn)).astype(float)
    g[:, 5] = g[:, 4]
    g[rng.random((m, nn)) < 0.02] = np.nan
    packed = O.pack_codes(O.genotypes_to_codes(g))
    maf = (np.nanmean(g, axis=1) / 2.0).astype(np.float32)
    flip = np.zeros(m, dtype=bool)
    for method, thr, abs_thr in [(1, 0.05, False), (2, 0.03, True), (1, -0.5, False)]:
        cp, ri, va = O.sparse_grm_csc_from_packed(packed, nn, flip, maf, None, method, thr, abs_thr)
        k64, _ = O.grm_packed(packed, nn, flip, maf, None, method, out_dtype=np.float64)
        seen = np.zeros((nn, nn), dtype=bool)
        for c in range(nn):
            rows = ri[int(cp[c]):int(cp[c + 1])]
            assert rows[0] == c and np.all(np.diff(rows.astype(np.int64)) > 0)
            assert np.array_equal(va[int(cp[c]):int(cp[c + 1])], k64[rows, c])
            seen[rows, c] = True
        for c in range(nn):
            for r in range(c + 1, nn):
                assert seen[r, c] == O.spgrm_keep_value(float(k64[r, c]), thr, abs_thr)
    with pytest.raises(RuntimeError, match="threshold must be finite"):
        O.sparse_grm_csc_from_packed(packed, nn, flip, maf, None, 1, float("inf"), False)


def test_sparse_reml_restatement_matches_reference_vectors():
    """Sparse REML null model (src/stats/spreml.rs).  Pins from the reference's own tests: the fixed-lambda profile
    objective on an indefinite K against explicit 2x2 algebra (:1209-1262), the fastGWA fixed-Vp objective (:1264-1329,
    recomputed here with numpy's dense inverse), the feasibility bisection (:1197-1207), the subset reordering
    (src/math/cholesky.rs:1656-1669) and the smoke properties of the grid search (:1165-1195)."""
    from oracle import jx_oracle as O
    k = O.sparse_grm_dense_subset(2, [0, 2, 3], [0, 1, 1], [1.0, 2.0, 1.0])
    assert np.array_equal(k, [[1.0, 2.0], [2.0, 1.0]])
    lam = 1.5
    ev = O.spreml_evaluate(k, np.ones((2, 1)), np.array([0.5, -1.25]), math.log10(lam))
    vi = np.array([[2.5, -2.0], [-2.0, 2.5]]) / 2.25               # (K + 1.5 I)^-1, det = 2.25
    y, x = np.array([0.5, -1.25]), np.ones(2)
    xvx, xvy = x @ vi @ x, x @ vi @ y
    ypy = y @ vi @ y - xvy ** 2 / xvx
    reml = (0.0 - 1.0 - math.log(2 * math.pi)) * 0.5 - 0.5 * (math.log(ypy) + math.log(2.25) + math.log(xvx))
    ml = 2.0 * (math.log(2.0) - 1.0 - math.log(2 * math.pi)) * 0.5 - 0.5 * (2.0 * math.log(ypy) + math.log(2.25))
    assert abs(ev["lam"] - lam) < 1e-12 and abs(ev["sigma_g2"] - ypy) < 1e-12
    assert abs(ev["sigma_e2"] - lam * ypy) < 1e-12 and abs(ev["reml"] - reml) < 1e-12 and abs(ev["ml"] - ml) < 1e-12
    with pytest.raises(RuntimeError):                              # K + 0.5 I is indefinite
        O.spreml_evaluate(k, np.ones((2, 1)), y, math.log10(0.5))
    # fastGWA fixed-Vp objective
    k3 = O.sparse_grm_dense_subset(3, [0, 2, 4, 5], [0, 1, 1, 2, 2], [1.0, 0.2, 1.0, 0.1, 1.0])
    y3, lam, vp = np.array([0.75, -0.10, -0.65]), 1.25, 1.2
    ev = O.spreml_evaluate(k3, np.ones((3, 1)), y3, math.log10(lam), vp_fixed=vp)
    sg2 = vp / (1.0 + lam)
    v = k3 * sg2 + np.eye(3) * (lam * sg2)
    vinv = np.linalg.inv(v)
    x3 = np.ones(3)
    ypy_v = y3 @ vinv @ y3 - (x3 @ vinv @ y3) ** 2 / (x3 @ vinv @ x3)
    want = -0.5 * (np.linalg.slogdet(v)[1] + math.log(x3 @ vinv @ x3) + ypy_v)
    assert abs(ev["sigma_g2"] - sg2) < 1e-12 and abs(ev["sigma_e2"] - lam * sg2) < 1e-12
    assert abs(ev["reml"] - want) < 1e-12 and math.isnan(ev["ml"])
    assert abs(O.refine_monotone_valid_lower_bound(lambda t: t >= -0.137, -0.625, 0.0, 1e-6, 64) + 0.137) <= 1e-4
    sub = O.sparse_grm_dense_subset(3, [0, 3, 5, 6], [0, 1, 2, 1, 2, 2], [1.0, 0.2, 0.3, 1.0, 0.4, 1.0], [2, 0, 1])
    assert np.allclose(sub[np.tril_indices(3)], [1.0, 0.3, 1.0, 0.4, 0.2, 1.0], atol=1e-12)   # rows of the lower triangle
    with pytest.raises(RuntimeError, match="duplicated sample index: 1"):
        O.sparse_grm_dense_subset(3, [0, 3, 5, 6], [0, 1, 2, 1, 2, 2], [1.0] * 6, [1, 1])
    res = O.spreml_sparse_reml_brent(3, [0, 2, 4, 5], [0, 1, 1, 2, 2], [1.0, 0.2, 1.0, 0.1, 1.0], [1.0, 0.5, -0.3],
                                     low=-3.0, high=2.0, grid_size=9, grid_only=True)
    assert math.isfinite(res[0]) and res[0] > 0.0 and math.isfinite(res[1]) and res[1] > 0.0 and len(res[6]) == 9
    # Brent never ends below the best grid point
    full = O.spreml_sparse_reml_brent(3, [0, 2, 4, 5], [0, 1, 1, 2, 2], [1.0, 0.2, 1.0, 0.1, 1.0], [1.0, 0.5, -0.3],
                                      low=-3.0, high=2.0, grid_size=9)
    assert full[4] >= max(full[7]) - 1e-12


def test_lm_restatement_is_ordinary_least_squares_with_a_t_test():
    """Plain LM scan (src/stats/glm.rs:3550-3860), the route after the LMM -> LM fallback: the reference ships no numeric
    test for it (parity unpinned beyond this property): per SNP the restatement must reproduce the f64 least-squares fit
    of y on [X, g] (g mean-imputed), its standard error, the two-sided Student-t p (scipy) and the likelihood-ratio p; the
    incomplete beta function of glm.rs:383-455 must agree with scipy's."""
    from scipy import special, stats as sst
    from oracle import jx_oracle as O
    from janusx_amd import bed
    for a, b, xx in ((0.5, 0.5, 0.3), (50.0, 0.5, 0.97), (2500.0, 0.5, 0.999), (3.0, 7.0, 0.1), (120.5, 0.5, 0.5)):
        assert abs(O.betai(a, b, xx) - special.betainc(a, b, xx)) <= 1e-10 * max(special.betainc(a, b, xx), 1e-300)
    assert O.student_t_p_two_sided(float("inf"), 10) == O.MIN_POSITIVE and math.isnan(O.student_t_p_two_sided(1.0, 0))
    n, m = 403, 60
    packed, g = bed.synth_panel_numpy(n, m, seed=8, missing_rate=0.03)
    rng = np.random.default_rng(2)
    x = np.concatenate([np.ones((n, 1)), rng.standard_normal((n, 2))], axis=1)
    y = x @ np.array([1.0, 0.4, -0.2]) + 0.5 * np.where(g[3] < 0, 0, g[3]) + rng.standard_normal(n)
    mi, he, ho = O.row_counts(packed, n)
    _keep, maf, _miss, flip = O.gwas_scan_row_stats(mi, he, ho, n, 0.0, 1.0, 1.0)
    flip = rng.random(m) < 0.3
    packed[7] = 0                                          # a monomorphic row: s = 0 -> NaN
    out = O.lm_block_assoc_packed(y, x, O.lm_precompute_ixx_qr(x), packed, n, flip, maf, None)
    assert np.isnan(out[7]).all()
    codes = O.unpack_codes(packed, n)
    df = n - 4
    for j in range(m):
        if j == 7:
            continue
        gj = O.lm_value_lut_f32(maf[j], bool(flip[j]))[codes[j]].astype(np.float64)
        d = np.concatenate([x, gj[:, None]], axis=1)
        coef, *_ = np.linalg.lstsq(d, y, rcond=None)
        r = y - d @ coef
        se = math.sqrt(float(r @ r) / df * np.linalg.inv(d.T @ d)[3, 3])
        t = out[j, 0] / out[j, 1]
        assert abs(out[j, 0] - coef[3]) <= 2e-6 * abs(coef[3]) + 2e-6 * se      # X, r_y rounded through f32
        assert abs(out[j, 1] - se) <= 2e-6 * se
        p = 2.0 * sst.t.sf(abs(t), df)
        assert abs(out[j, 2] - p) <= 1e-9 * p
        plrt = sst.chi2.sf(n * math.log1p(t * t / df), 1)
        assert abs(out[j, 3] - plrt) <= 1e-9 * plrt
    assert out[3, 2] < 1e-6                                                     # the causal SNP
    sub = np.sort(rng.permutation(n)[:250])
    o2 = O.lm_block_assoc_packed(y[sub], x[sub], O.lm_precompute_ixx_qr(x[sub]), packed, n, flip, maf, sub)
    gj = O.lm_value_lut_f32(maf[3], bool(flip[3]))[codes[3, sub]].astype(np.float64)
    coef, *_ = np.linalg.lstsq(np.concatenate([x[sub], gj[:, None]], axis=1), y[sub], rcond=None)
    assert abs(o2[3, 0] - coef[3]) <= 2e-6 * abs(coef[3])


def test_splmm_approx_route_restatement(oracle):
    """The `-splmm` approximate route (src/stats/splmm_approx.rs:612-795, src/stats/splmm.rs:2935-3316): (i) the ChaCha core
    behind the seeded marker choice against RFC 7539 section 2.3.2 (20 rounds; `StdRng` runs the same core with 12 -- the
    seeding and range-sampling conventions of rand 0.9.2 are restated from memory of the published source and stay unpinned,
    as the oracle's header says); (ii) `choose_rhat_rows`: sorted, unique, inside the range, at most 2 count, everything
    when count >= m, a function of the seed; (iii) with K = I the route must collapse to ordinary least squares: gamma =
    1 / ((1 + lambda) sigma2), beta = the OLS coefficient of the SNP given X, se^2 = RSS0 / (df g'M g)."""
    import math
    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    blk = oracle._chacha_block(key, 1 | (0x09000000 << 32), 20, (0x4a000000, 0))
    assert blk[:4] == [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3] and blk[15] == 0x4e3c50a2
    rr = oracle.choose_rhat_rows(100000, 30, 20260527)
    assert np.all(np.diff(rr) > 0) and rr.min() >= 0 and rr.max() < 100000 and 30 <= len(rr) <= 60
    assert np.array_equal(rr, oracle.choose_rhat_rows(100000, 30, 20260527))
    assert not np.array_equal(rr, oracle.choose_rhat_rows(100000, 30, 20260528))
    assert np.array_equal(oracle.choose_rhat_rows(20, 30, 1), np.arange(20))
    big = oracle.StdRngU32(7)
    assert all(0 <= big.random_range(5_000_000_000) < 5_000_000_000 for _ in range(50))
    rng = np.random.default_rng(3)
    n, m = 120, 40
    g = rng.integers(0, 3, size=(m, n)).astype(np.int8)
    packed = oracle.pack_codes(oracle.genotypes_to_codes(g))
    maf = (g.sum(1) / (2.0 * n)).astype(np.float32)
    flip = np.zeros(m, dtype=bool)
    x = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, 1))], axis=1)
    y = rng.normal(size=n) + 0.4 * g[3]
    lam = 0.7
    gamma, out, used, rows = oracle.splmm_approx_assoc(np.eye(n), lam, x, y, packed, n, maf, flip, rhat_markers=30)
    q, _ = np.linalg.qr(x)
    yr = y - q @ (q.T @ y)
    rss, df = float(yr @ yr), n - 2
    sigma2 = rss / (df * (1.0 + lam))
    assert abs(gamma - 1.0 / ((1.0 + lam) * sigma2)) < 1e-12 * gamma and used == len(rows)
    for j in range(m):
        gj = g[j].astype(np.float64)
        gr = gj - q @ (q.T @ gj)
        sms = float(gr @ gr)
        beta, se = float(gr @ yr) / sms, math.sqrt(rss / (df * sms))
        assert abs(out[j, 0] - beta) < 2e-6 * max(abs(beta), se) and abs(out[j, 1] - se) < 1e-6 * se       # f32 dots
        assert abs(out[j, 2] - oracle.chi2_sf_df1((beta / se) ** 2)) < 1e-4 * out[j, 2] + 1e-300


REFMODEL = os.path.join(os.path.dirname(__file__), "golden", "reference_model.npz")


def test_reference_model_layer_pins_the_oracle_composition(oracle):
    """Values the REFERENCE'S OWN model layer produced in the build container (tests/golden/gen_reference_model_fixtures.py:
    python/janusx/pyBLUP/assoc.py `LMM.__init__` / `_initialize_from_spectral` :1702-1876, `LMM.gwas` :1962, `LMM2.gwas`,
    `FastLMM.gwas`, `FvLMM.gwas / gwas_rotated` :2072-2180 and `janusx/assoc/api.py::ASSOC` run over a stub native module
    that records its calls).  The oracle's own composition of the path -- the 1e-6 ridge before the eigendecomposition, the
    design [1, X], `spectral_null_model` (lambda_0, ML0, LL0, sigma_g2, sigma_e2, trace mean, the diagonal-scaled PVE, the scan
    bounds log10 lambda_0 +- 2 with the (-5, 5) fallback outside 0.05 <= PVE <= 0.95), the exact scan's bounds / 30 iterations /
    tolerance 1e-2 without a null ML, the fixed-lambda scans at log10 lambda_0, FastLMM's switch -- must reproduce them."""
    r = np.load(REFMODEL)
    n = int(r["n"])
    y, xe, k = r["y"], r["x_extra"], r["k"]
    # the layer's eigendecomposition input: f64 copy of K with 1e-6 on the diagonal (assoc.py:1623-1628)
    a = k.astype(np.float64)
    a.flat[:: n + 1] += 1e-6
    assert np.array_equal(a, r["eigh_input"])
    s, u = oracle.gwas_eigh_from_grm(k, 1e-6)
    assert np.max(np.abs(s - r["eigh_w"])) < 1e-12 * max(1.0, float(np.max(np.abs(s))))
    x = np.concatenate([np.ones((n, 1)), xe], axis=1)
    assert np.array_equal(x, r["rot_x_in"])
    assert r["null_args"].tolist() == [-5.0, 5.0, 50.0, 1e-3]
    for tag, yy in (("lmm", y), ("noise", r["y_noise"]), ("gen", r["y_gen"])):
        nm = oracle.spectral_null_model(yy, x, r["eigh_w"], r["eigh_v"])
        for mine, key in ((nm.lbd_null, "lbd_null"), (nm.ML0, "ML0"), (nm.LL0, "LL0"), (nm.sigma_g2, "sigma_g2"),
                          (nm.sigma_e2, "sigma_e2"), (nm.pve, "pve"), (nm.trace_mean, "trace_mean")):
            ref = float(r[f"{tag}_{key}"])
            assert abs(mine - ref) <= 1e-12 * max(1.0, abs(ref)), (tag, key, mine, ref)
        assert np.allclose(np.array(nm.bounds), r[f"{tag}_bounds"], rtol=0, atol=1e-12), (tag, nm.bounds)
        assert np.array_equal(nm.Dh, r[f"{tag}_Dh"]) and np.array_equal(nm.Xcov, r[f"{tag}_Xcov"])
        assert np.array_equal(nm.y, r[f"{tag}_yrot"]) and not bool(r[f"{tag}_lowrank"]) and int(r[f"{tag}_rank"]) == n
    assert r["noise_bounds"].tolist() == [-5.0, 5.0] and float(r["noise_pve"]) < 0.05
    assert r["gen_bounds"].tolist() == [-5.0, 5.0] and float(r["gen_pve"]) > 0.95
    nm = oracle.spectral_null_model(y, x, r["eigh_w"], r["eigh_v"])
    # LMM.gwas: what the layer hands the exact scan
    lo, hi, it, tol, rot_rows = r["lmm_gwas_args"]
    assert (lo, hi) == nm.bounds and it == 30 and tol == 1e-2 and rot_rows == r["snp"].shape[0] and bool(r["lmm_gwas_nullml_is_none"])
    t = oracle.lmm_reml_chunk_from_snp(nm.S, nm.Xcov, nm.y, lo, hi, r["snp"], nm.Dh, 30, 1e-2)
    assert np.array_equal(t, r["lmm_gwas"], equal_nan=True)
    # the in-memory API ends in the same native calls with the same arguments
    assert np.array_equal(r["api_lmm_table"], r["lmm_gwas"], equal_nan=True)
    assert np.array_equal(r["api_fvlmm_table"], r["fvlmm_gwas"], equal_nan=True)
    assert str(r["api_lmm_native_calls"]).split(";")[-1] == "lmm_reml_chunk_from_snp_f32"
    assert str(r["api_fvlmm_native_calls"]).split(";")[-2:] == ["fvlmm_assoc_prepare_cache_f32", "fvlmm_assoc_chunk_from_snp_with_cache_f32"]
    # FvLMM / FastLMM: log10 lambda_0, one cache per trait, raw and rotated entry points agree
    assert abs(float(r["fvlmm_log10_lbd"]) - math.log10(nm.lbd_null)) < 1e-15
    fv = oracle.fvlmm_assoc_chunk_from_snp(nm.S, nm.Xcov, nm.y, math.log10(nm.lbd_null), r["snp"], nm.Dh)
    assert np.array_equal(fv, r["fvlmm_gwas"], equal_nan=True)
    assert np.array_equal(oracle.rotate_block_f32(r["snp"], nm.Dh), r["grot"])
    assert np.array_equal(r["fvlmm_gwas_rotated"], r["fvlmm_gwas"], equal_nan=True)
    assert str(r["fastlmm_route"]) == "lmm_assoc_chunk_from_snp_f32"            # 0.05 <= PVE <= 0.95: the fixed-lambda kernel
    fl = oracle.lmm_assoc_fixed_lambda_block(r["grot"], nm.S, nm.Xcov, nm.y, math.log10(nm.lbd_null))
    assert np.array_equal(fl, r["fastlmm_gwas"], equal_nan=True)
    # LMM2: the layer's null ML optimum (scipy bounded search over the native ML likelihood inside the scan bounds)
    lo2, hi2, it2, tol2, nullml = r["lmm2_gwas_args"]
    assert (lo2, hi2, it2, tol2) == (lo, hi, 30.0, 1e-2) and nullml == float(r["lmm2_ml0_exact"])
    ev = r["lmm2_ml_evals"]
    assert np.all((ev >= lo) & (ev <= hi))
    best = max(oracle.ml_loglike(float(t_), nm.S, nm.Xcov, nm.y, None) for t_ in ev)
    assert abs(best - nullml) < 1e-12 * abs(nullml)
    assert abs(10.0 ** float(ev[-1]) - float(r["lmm2_lbd_null_ml"])) < 1e-12
    l2 = oracle.lmm2_scan_rotated_block(r["grot"], nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2, nullml)
    assert np.array_equal(l2, r["lmm2_gwas"], equal_nan=True)
    # the oracle's own LMM2 null ML (Brent, the BED route src/stats/lmm.rs:2902-2921: a different optimiser, tolerance 1e-2 in
    # log10 lambda) stops at the same boundary optimum within its tolerance
    xm, ml0 = oracle.lmm2_null_ml(nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2)
    assert abs(xm - math.log10(float(r["lmm2_lbd_null_ml"]))) < 5e-2 and nullml - 0.1 < ml0 <= nullml + 1e-9


def test_stdrng_published_vectors(oracle):
    """`StdRng` (rand 0.9: ChaCha12, `splmm_choose_rhat_rows` draws from it; src/stats/splmm.rs:1493-1507) against PUBLISHED
    known answers: the all-zero key / nonce keystream blocks of ChaCha8 / 12 / 20 (Strombergson's ChaCha test vectors, TC1), rand's
    own value-stability test of `StdRng` (rngs/std.rs `test_stdrng_construction`: seed bytes 1, 23, 200+256, 210+30*256 ->
    next_u64 = 10719222850664546238, then an StdRng filled from that one -> 14064965282130556830: ChaCha12, word order, u64
    assembly, `from_rng` filling the seed from consecutive words) and rand_chacha's `test_chacha_construction` (ChaCha20, seed words
    0, 0, 1, 0, 2, 0, 3, 0 -> next_u32 = 137206642).  What stays restated without a vector: the PCG32 expansion of `seed_from_u64` and
    the Canon range sampler of `random_range` (oracle header; README)."""
    import struct
    def hexblock(rounds):
        return b"".join(struct.pack("<I", w) for w in oracle._chacha_block([0] * 8, 0, rounds)).hex()
    assert hexblock(8).startswith("3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e")
    assert hexblock(12).startswith("9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f")
    assert hexblock(20).startswith("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7")

    def rng_from_words(words):
        g = oracle.StdRngU32.__new__(oracle.StdRngU32)
        g.key, g.counter, g.buf = list(words), 0, []
        return g
    seed = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)
    g0 = rng_from_words(struct.unpack("<8I", seed))
    assert g0.next_u64() == 10719222850664546238
    g1 = rng_from_words([g0.next_u32() for _ in range(8)])
    assert g1.next_u64() == 14064965282130556830
    assert oracle._chacha_block([0, 0, 1, 0, 2, 0, 3, 0], 0, 20)[0] == 137206642


def test_genetic_model_tables(oracle):
    """`PackedGeneticModel` (src/decode/decode.rs:100-178): the model is applied to the decode table [0 | 2, 2 maf, 1, 2 | 0] -- the
    imputed entry included -- before the row is centred; parse is case-insensitive and refuses anything else.  The host-side
    table of the product (stats.scan_lut_from_counts: model, then the row mean from the genotype counts) must reproduce the
    oracle's decoded rows bit for bit."""
    from janusx_amd import stats
    f = np.float32
    assert oracle.scan_value_lut_f32(f(0.25), False, "dom").tolist() == [0.0, 1.0, 1.0, 1.0]      # imputed 0.5 > 0 -> 1
    assert oracle.scan_value_lut_f32(f(0.25), True, "DOM").tolist() == [1.0, 1.0, 1.0, 0.0]
    assert oracle.scan_value_lut_f32(f(0.25), False, "rec").tolist() == [0.0, 0.0, 0.0, 1.0]
    assert oracle.scan_value_lut_f32(f(1.0), False, "rec").tolist() == [0.0, 1.0, 0.0, 1.0]       # imputed 2 maf = 2 counts as hom-alt
    assert oracle.scan_value_lut_f32(f(0.5), False, "het").tolist() == [0.0, 1.0, 1.0, 0.0]       # imputed 2 maf = 1 counts as het
    assert oracle.scan_value_lut_f32(f(0.3), True, "Het").tolist() == [0.0, 0.0, 1.0, 0.0]
    with pytest.raises(RuntimeError, match="model must be one of: add, dom, rec, het"):
        oracle.scan_value_lut_f32(f(0.3), False, "overdominant")
    with pytest.raises(RuntimeError, match="model must be one of: add, dom, rec, het"):
        stats.genetic_model_code("x")
    rng = np.random.default_rng(11)
    n, m = 97, 60
    g = rng.integers(0, 3, size=(m, n)).astype(np.int8)
    g[rng.random((m, n)) < 0.05] = -9
    g[5, :] = np.where(g[5] == 2, 1, g[5])                      # no hom-alt call: `rec` makes the row constant
    from janusx_amd import bed
    packed = bed.pack_dosage(g)
    mi, he, ho = oracle.row_counts(packed, n)
    counts = np.stack([mi, he, ho], 1)
    maf = ((he + 2 * ho) / np.maximum(2 * (n - mi), 1)).astype(np.float32)
    maf[7] = 0.5                                                 # imputed entry = 1: a het under `het`
    flip = rng.random(m) < 0.3
    for gm in ("add", "dom", "rec", "het"):
        lut = stats.scan_lut_from_counts(maf, flip, counts, n, model=gm)
        dec = oracle.decode_centered_block_f32(packed, n, flip, maf, model=gm)
        codes = oracle.unpack_codes(packed, n)
        assert np.array_equal(np.take_along_axis(lut, codes.astype(np.int64), axis=1), dec), gm
    assert np.all(oracle.decode_centered_block_f32(packed, n, flip, maf, rows=np.array([5]), model="rec") == 0.0) or flip[5]
