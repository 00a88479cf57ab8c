"""GPU parity tests added in round 6: the image scope of the PCG routes with host payloads, the eigensolver at the digit-plane
count the pipeline really runs, a true end-to-end leg above the two-stage threshold, the reference's warm-start chain.
Same norms and tolerances as tests/test_gpu_parity.py (north star: beta / SE / Wald p within 1e-5, SNP set bit-exact)."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from janusx_amd import bed  # noqa: E402


def _panel_stats(oracle, packed, n):
    _miss, maf, _std, flip = oracle.load_bed_2bit_packed_stats(packed, n)
    return maf, flip


def test_pcg_image_scope_two_host_payloads_do_not_share_images(oracle):
    """the scope's cache was keyed on the address of the call-owned upload buffer, which the next call's upload
    usually gets back -- a second HOST payload of the same shape inside one scope then solved on the first payload's images.
    Two different host payloads (same shape, same training samples) inside one scope must each give their own solution; a
    DEVICE payload inside the scope still reuses its images (same result as outside the scope)."""
    import torch
    from janusx_amd import janusx as jxrs
    n, m = 384, 1500
    pa, ga = bed.synth_panel_numpy(n, m, seed=41, missing_rate=0.01)
    pb, gb = bed.synth_panel_numpy(n, m, seed=43, missing_rate=0.01)
    assert pa.shape == pb.shape and not np.array_equal(pa, pb)
    rng = np.random.default_rng(3)
    tr = np.sort(rng.permutation(n)[:300]).astype(np.int64)
    te = np.setdiff1d(np.arange(n), tr).astype(np.int64)
    y = rng.standard_normal(len(tr))
    lam = float(m)

    def solve(pk):
        maf, flip = _panel_stats(oracle, pk if isinstance(pk, np.ndarray) else pk.cpu().numpy(), n)
        return jxrs.rrblup_pcg_bed("", tr, y, te, lambda_value=lam, tol=1e-8, max_iter=300, packed=pk, packed_n_samples=n,
                                   maf=maf, row_flip=flip)

    ref_a, ref_b = solve(pa), solve(pb)
    assert not np.allclose(ref_a[9], ref_b[9])
    with jxrs.pcg_image_scope():
        in_a = solve(pa)
        in_b = solve(pb)          # same shape, same rows, same training samples, most likely the same upload address
        in_a2 = solve(pa)
    for got, ref in ((in_a, ref_a), (in_b, ref_b), (in_a2, ref_a)):
        assert got[4] == ref[4] and np.array_equal(got[9], ref[9]) and np.array_equal(got[1], ref[1])
    da = torch.from_numpy(pa).cuda()
    with jxrs.pcg_image_scope():
        d1 = solve(da)
        maf, flip = _panel_stats(oracle, pa, n)
        he = jxrs.he_pcg_bed("", tr, y, packed=da, packed_n_samples=n, maf=maf, row_flip=flip)    # reuses the images
        d2 = solve(da)
    assert np.array_equal(d1[9], ref_a[9]) and np.array_equal(d2[9], ref_a[9]) and len(he) == 12


def _parity():
    """Norm helpers of the main parity file (`_assoc_err`, `TOL`): one definition of the bars."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import test_gpu_parity as P
    return P


@pytest.mark.parametrize("missing", [0.002, 0.01])
def test_end_to_end_two_stage(oracle, oracle_c, missing):
    """Both missing-call rates put every SNP on the dense missing-call form of the GRM (csrc/k_grm.hip: the missing call's count in
    int8 digits).  With TWO digits (rounds 4 - 5) this leg measured beta off by 1.7e-5 at 0.2 % and 1.3e-4 at 1 % missing calls --
    an entry error of 1.3e-6 / 3.2e-6 in K, inside the GRM bar, amplified by the eigendecomposition (scripts/diag_e2e_two_stage.py;
    the full-size legs hand the GPU's own spectrum to the oracle and cannot see it) -- hence the third digit.
    A TRUE end-to-end leg above the two-stage threshold: n = 5000, m = 20 000 through `pipeline.run_gwas` (GRM on the int8
    pipes, the own two-stage eigensolver with Q1 and the divide-and-conquer merges on 5 digit planes -- sliced products engage
    from n = 3000 -- f32 U^T, null, exact-scan / fixed-lambda scan) against an oracle that builds its OWN GRM (f32 SYRK + f64
    merge), its OWN dsyevd, null fit, f32 rotation and scan (as `test_pipeline_end_to_end` does at n = 400, below every sliced
    product).  Reference contract: python/janusx/assoc/workflow.py:5639-5641 (ridge 1e-6, f64 eigh), src/stats/reml.rs:109-198
    (U^T kept f32).  Bars: the north star's 1e-5 on beta / SE / p, kept set and af / miss bit-exact, lambda 1e-5."""
    import torch
    from janusx_amd import pipeline
    P = _parity()
    n, m = 5000, 20000
    packed, g = bed.synth_panel_numpy(n, m, seed=61, missing_rate=missing)
    y = bed.synth_phenotype(g, n_causal=40, pve=0.5, seed=61)
    del g
    mi, he, ho = oracle.row_counts(packed, n)
    k_ref, eff, _ = oracle.grm_stream_bed(packed, n, 1, 0.02, 0.05, 0.0)
    s, u = oracle.gwas_eigh_from_grm(k_ref)
    nm = oracle.spectral_null_model(y, np.ones((n, 1)), s, u)
    keep, maf, miss, flip = oracle.gwas_scan_row_stats(mi, he, ho, n, 0.02, 0.05, 1.0)
    rows = np.nonzero(keep)[0]
    pick = np.arange(len(rows))            # every kept SNP (the C oracle scans 20 000 rows at n = 5000 in seconds)
    gd = oracle.decode_centered_block_f32(packed, n, flip, maf, rows=rows[pick])
    grot = oracle.rotate_block_f32(gd, nm.Dh)
    dev_payload = torch.from_numpy(packed).cuda()
    for mode in ("lmm", "fvlmm"):
        res = pipeline.run_gwas(dev_payload, n, y, mode=mode)
        assert pipeline.LAST_EIGH["planes"] == 5
        assert res.grm_eff_m == eff
        assert np.array_equal(keep, res.keep)
        assert np.array_equal(res.af, maf[rows]) and np.array_equal(res.miss, miss[rows])
        assert abs(res.null.lbd - nm.lbd_null) < 1e-5 * nm.lbd_null, (res.null.lbd, nm.lbd_null)
        assert abs(res.null.ml0 - nm.ML0) < 1e-6 * abs(nm.ML0)
        assert abs(res.null.pve - nm.pve) < 1e-5
        if mode == "lmm":
            ref = oracle_c.lmm_scan_rotated_block(grot, nm.S, nm.Xcov, nm.y, nm.bounds[0], nm.bounds[1], 30, 1e-2,
                                                  threads=os.cpu_count() or 1)
        else:
            ref = oracle.fvlmm_assoc_rotated_block(grot, oracle.fvlmm_prepare_cache(nm.S, nm.Xcov, nm.y, nm.lbd_null))
        # Wald p: relative error / max(1, z^2) within 1e-5 (`pe`: a relative error eps on beta / SE IS z^2 eps on the normal tail)
        # and RAW relative error within 1e-5 wherever the tail does not amplify it (z^2 <= 10); the strongest SNP of this panel
        # has z^2 in the hundreds, so the fixed raw bound of the other legs (1.2e-4, set at z^2 ~ 170) does not apply here
        be, se, pe = P._assoc_err(res.stats[pick], ref, tag=mode, raw_p_bound=None)
        assert max(be, se, pe) < P.TOL, (mode, be, se, pe)
        ok = ~np.isnan(ref[:, 0])
        z2 = (ref[ok, 0] / ref[ok, 1]) ** 2
        praw = np.abs(res.stats[pick][ok, 2] - ref[ok, 2]) / ref[ok, 2]
        assert float(praw[z2 <= 10.0].max()) < P.TOL, (mode, float(praw[z2 <= 10.0].max()))
        assert float((praw / np.maximum(1.0, z2)).max()) < P.TOL
        print(f"end to end n={n} m={m} {mode}: beta {be:.2e} se {se:.2e} p(norm) {pe:.2e} raw p max {praw.max():.2e} at z2 "
              f"{z2[np.argmax(praw)]:.0f}, raw p (z2<=10) {praw[z2 <= 10.0].max():.2e}")


# ---- the reference's default exact scan: warm-start chains (src/stats/lmm.rs:134-161, 2627, 3244-3245) ---------------------

@pytest.fixture(scope="module")
def chain_case(oracle):
    """n = 600 samples, m = 3000 SNPs with 1 % missing calls, intercept + one covariate; spectral inputs from the oracle."""
    n, m = 600, 3000
    packed, g = bed.synth_panel_numpy(n, m, seed=71, missing_rate=0.01)
    y = bed.synth_phenotype(g, n_causal=25, pve=0.5, seed=71)
    k, _eff, _keep = oracle.grm_stream_bed(packed, n, 1, 0.02, 0.05, 0.0)
    s, u = oracle.gwas_eigh_from_grm(k)
    x = np.concatenate([np.ones((n, 1)), np.random.default_rng(8).normal(size=(n, 1))], axis=1)
    nm = oracle.spectral_null_model(y, x, s, u)
    mi, he, ho = oracle.row_counts(packed, n)
    keep, maf, miss, flip = oracle.gwas_scan_row_stats(mi, he, ho, n, 0.02, 0.05, 1.0)
    kept = np.nonzero(keep)[0]
    grot = oracle.rotate_block_f32(oracle.decode_centered_block_f32(packed, n, flip, maf, rows=kept), nm.Dh)
    return dict(n=n, m=m, packed=packed, nm=nm, keep=keep, kept=kept, maf=maf, miss=miss, flip=flip, grot=grot, y=y, x=x)


@pytest.mark.parametrize("bounds", ["workflow", "wide"])
def test_warm_start_chain_packed_entry_point(oracle, oracle_c, chain_case, bounds, monkeypatch):
    """`lmm_reml_assoc_packed_f32` with the reference's default warm start (`true, true`, src/stats/lmm.rs:3244-3245) against the
    oracle's sequential chains: with and without `init_log10_lbd`, `rotate_block_rows` 256 and 100, `progress_every` cutting the
    blocks, rayon's halving (`warm_chain_pieces`), the plrt column; bounds = the workflow's [log10 lambda0 -/+ 2] (series form:
    per-SNP series + one Brent launch over all chains) and the API default [-5, 5] (no series form: chains walked block by block
    with carried states).  beta / SE / p within 1e-5, NaN pattern identical; a chain is also DIFFERENT from the no-warm-start scan
    (Brent stops within tol = 1e-2 of the optimum from wherever it starts) -- that the chain is really taken is part of the
    assertion.  JX_LMM_UNIFIED_NO_WARM_START and warm_start='none' both give the no-chain scan."""
    from janusx_amd import janusx as jxrs
    from janusx_amd import stats as st
    P = _parity()
    monkeypatch.delenv("JX_LMM_UNIFIED_NO_WARM_START", raising=False)
    c = chain_case
    n, nm, kept = c["n"], c["nm"], c["kept"]
    pk = np.ascontiguousarray(c["packed"][kept])
    maf_k, flip_k = c["maf"][kept], c["flip"][kept]
    lo, hi = nm.bounds if bounds == "workflow" else (-5.0, 5.0)
    init = math.log10(nm.lbd_null)
    m = len(kept)
    plain = oracle_c.lmm_scan_rotated_block(c["grot"], nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2)
    for kw, blocks, pieces in ((dict(), st.warm_chain_blocks_packed(m, 256), 1),
                               (dict(init_log10_lbd=init, rotate_block_rows=100), st.warm_chain_blocks_packed(m, 100), 1),
                               (dict(init_log10_lbd=init, rotate_block_rows=256, progress_every=600),
                                st.warm_chain_blocks_packed(m, 256, 600), 1),
                               (dict(rotate_block_rows=512, warm_chain_pieces=4), st.warm_chain_blocks_packed(m, 512), 4)):
        co = st.warm_chain_offsets(blocks, m, pieces)
        ref = oracle_c.lmm_scan_rotated_chains(c["grot"], nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2, co, init=kw.get("init_log10_lbd"))
        got = jxrs.lmm_reml_assoc_packed_f32(pk, n, flip_k, maf_k, nm.S, nm.Xcov, nm.y, nm.Dh, low=lo, high=hi, max_iter=30,
                                             tol=1e-2, **kw)
        be, se, pe = P._assoc_err(got, ref, tag=f"{bounds}:{sorted(kw)}")
        assert max(be, se, pe) < P.TOL, (bounds, kw, be, se, pe)
        ok = ~np.isnan(ref[:, 0])
        d_chain = np.abs(got[ok, 0] - ref[ok, 0]) / np.maximum(np.abs(ref[ok, 0]), ref[ok, 1])
        d_plain = np.abs(got[ok, 0] - plain[ok, 0]) / np.maximum(np.abs(plain[ok, 0]), plain[ok, 1])
        assert d_plain.max() > 10 * max(d_chain.max(), 1e-7), "the chain scan is indistinguishable from the no-warm-start scan"
    # plrt column along the chains
    co = st.warm_chain_offsets(st.warm_chain_blocks_packed(m, 256), m)
    ref4 = oracle_c.lmm_scan_rotated_chains(c["grot"], nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2, co, nullml=nm.ML0)
    got4 = jxrs.lmm_reml_assoc_packed_f32(pk, n, flip_k, maf_k, nm.S, nm.Xcov, nm.y, nm.Dh, low=lo, high=hi, max_iter=30, tol=1e-2,
                                          nullml=nm.ML0)
    ok = ~np.isnan(ref4[:, 0])
    assert got4.shape[1] == 4 and np.max(np.abs(got4[ok, 3] - ref4[ok, 3]) / np.maximum(ref4[ok, 3], 1e-300)) < 1e-4
    # the two ways of switching the chain off
    off = jxrs.lmm_reml_assoc_packed_f32(pk, n, flip_k, maf_k, nm.S, nm.Xcov, nm.y, nm.Dh, low=lo, high=hi, max_iter=30, tol=1e-2,
                                         warm_start="none")
    be, se, pe = P._assoc_err(off, plain, tag=f"{bounds}:none")
    assert max(be, se, pe) < P.TOL
    monkeypatch.setenv("JX_LMM_UNIFIED_NO_WARM_START", "yes")
    off2 = jxrs.lmm_reml_assoc_packed_f32(pk, n, flip_k, maf_k, nm.S, nm.Xcov, nm.y, nm.Dh, low=lo, high=hi, max_iter=30, tol=1e-2)
    assert np.array_equal(off, off2, equal_nan=True)


def test_warm_start_chain_device_pipeline_matches_trajectories(oracle, oracle_c, chain_case):
    """`pipeline.scan_rows(chain_off=...)` (the route `jx gwas -lmm` takes): the number of Brent evaluations of every SNP equals
    the oracle's along the same chains (the trajectory, not only the result, is the reference's), results within 1e-5; the series
    form (one Brent launch behind the last block) and the block-by-block form with carried states (wide bounds; blocks of 700 rows
    so that chains are cut by block boundaries) both."""
    import torch
    from janusx_amd import pipeline
    from janusx_amd import stats as st
    P = _parity()
    c = chain_case
    n, nm, kept = c["n"], c["nm"], c["kept"]
    dev = torch.device("cuda", 0)
    panel = pipeline.Panel(torch.from_numpy(c["packed"]).to(dev), n)
    s_t = torch.from_numpy(nm.S).to(dev)
    ut64 = torch.from_numpy(nm.Dh.astype(np.float64)).to(dev)
    # the oracle's model in the device container (same S, Dh; X~ / y~ recomputed on the device from the same Dh)
    model = pipeline.SpectralModel(s_t, ut64, c["x"], c["y"])
    counts = panel.counts()
    lut = st.scan_lut_from_counts(c["maf"][kept], np.zeros(len(kept), bool), counts[kept], n)
    sh, xh, yh = model.S.cpu().numpy(), model.xcov.cpu().numpy(), model.y.cpu().numpy()
    co = st.warm_chain_offsets(st.warm_chain_blocks_bed(kept, panel.m, 500), len(kept))
    assert len(co) > 4
    for lo, hi, br in ((model.null.bounds[0], model.null.bounds[1], 8192), (-5.0, 5.0, 700)):
        init = min(max(math.log10(model.null.lbd), lo), hi)
        ref, ev_ref = oracle_c.lmm_scan_rotated_chains(c["grot"], sh, xh, yh, lo, hi, 30, 1e-2, co, init=init, return_evals=True)
        out, ev = pipeline.scan_rows(panel, model, kept, lut, "lmm", low=lo, high=hi, max_iter=30, tol=1e-2, init_log10_lbd=init,
                                     block_rows=br, return_evals=True, chain_off=co)
        be, se, pe = P._assoc_err(out.cpu().numpy(), ref, tag=f"br{br}")
        assert max(be, se, pe) < P.TOL, (lo, hi, be, se, pe)
        same = float(np.mean(ev.cpu().numpy() == ev_ref))
        assert same > 0.995, f"only {same:.4f} of the SNPs took the oracle's number of Brent evaluations"


def test_warm_start_chain_bed_route_tsv_text(oracle, oracle_c, chain_case, tmp_path, monkeypatch):
    """`lmm_reml_assoc_bed_to_tsv_f32` as the reference's workflow calls it (init_log10_lbd = log10 lambda0, rotate_block_rows = the
    chunk size; chain on unless JX_LMM_UNIFIED_NO_WARM_START, src/stats/lmm.rs:2627): the TSV text equals the oracle's rendering of
    the oracle's chain scan (kept rows of every chunk of 400 SNP rows of the file = one chain) up to last-digit flips of the
    4-digit columns; with the variable set the file equals the no-chain scan's."""
    from janusx_amd import janusx as jxrs
    from janusx_amd import stats as st
    monkeypatch.delenv("JX_LMM_UNIFIED_NO_WARM_START", raising=False)
    c = chain_case
    n, m, nm, kept = c["n"], c["m"], c["nm"], c["kept"]
    prefix = str(tmp_path / "panel")
    ids = [f"s{i}" for i in range(n)]
    bim = bed.Bim([str(1 + j % 3) for j in range(m)], [f"rs{j}" for j in range(m)], [10 + j for j in range(m)], ["A"] * m, ["C"] * m)
    bed.write_bed(prefix, c["packed"], ids, bim)
    lo, hi = nm.bounds
    init = math.log10(nm.lbd_null)
    out = str(tmp_path / "chain.tsv")
    rows = jxrs.lmm_reml_assoc_bed_to_tsv_f32(prefix, out, nm.S, nm.Xcov, nm.y, nm.Dh, 0.02, 0.05, 1.0, low=lo, high=hi,
                                              max_iter=30, tol=1e-2, init_log10_lbd=init, rotate_block_rows=400)
    assert rows == len(kept)
    co = st.warm_chain_offsets(st.warm_chain_blocks_bed(kept, m, 400), len(kept))
    assert len(co) == (m + 399) // 400 + 1
    ref = oracle_c.lmm_scan_rotated_chains(c["grot"], nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2, co, init=init)
    plain = oracle_c.lmm_scan_rotated_block(c["grot"], nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2)
    lines = open(out).read().splitlines()[1:]
    got = np.array([[float(f) for f in (ln.split("\t")[7], ln.split("\t")[8])] for ln in lines])
    assert np.max(np.abs(got[:, 0] - ref[:, 0])) < 1.01e-4 and np.max(np.abs(got[:, 1] - ref[:, 1])) < 1.01e-4
    # the text differs from the no-chain table's in printed digits on some SNPs -- the chain is what the reference prints
    n_chain = int(np.sum(np.abs(got[:, 0] - np.round(ref[:, 0], 4)) < 5e-5))
    n_plain = int(np.sum(np.abs(got[:, 0] - np.round(plain[:, 0], 4)) < 5e-5))
    assert n_chain >= 0.995 * len(kept) and n_plain < n_chain, (n_chain, n_plain, len(kept))   # a few last-digit flips at .xxxx5
    monkeypatch.setenv("JX_LMM_UNIFIED_NO_WARM_START", "1")
    out2 = str(tmp_path / "plain.tsv")
    jxrs.lmm_reml_assoc_bed_to_tsv_f32(prefix, out2, nm.S, nm.Xcov, nm.y, nm.Dh, 0.02, 0.05, 1.0, low=lo, high=hi, max_iter=30,
                                       tol=1e-2, init_log10_lbd=init, rotate_block_rows=400)
    got2 = np.array([[float(f) for f in (ln.split("\t")[7], ln.split("\t")[8])] for ln in open(out2).read().splitlines()[1:]])
    seeded = oracle_c.lmm_scan_rotated_block(c["grot"], nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2, warm=1, init=init)
    assert np.max(np.abs(got2[:, 0] - seeded[:, 0])) < 1.01e-4


def test_warm_start_chain_at_c3_sample_size(oracle, oracle_c):
    """The chain scan at the sample size of BASELINE configs[2] (n = 20 000; 3000 SNPs): two-stage eigensolver at 5 planes,
    int8 rotation, series form, chains of 512 rows -- against the oracle's chains on the device's own spectral inputs (as the
    full-size legs do), beta / SE / p within 1e-5, identical NaN pattern, and through `pipeline.run_gwas(warm_chain=...)`."""
    import torch
    import bench
    from janusx_amd import pipeline
    from janusx_amd import stats as st
    P = _parity()
    n, m = 20000, 3000
    dev = torch.device("cuda", 0)
    packed, dos = bench.synth_panel_gpu(n, m, 20260611, dev, missing_rate=0.0)
    y = bench.make_phenotype(dos, n, 20260611, dev)
    k, _eff, panel = pipeline.build_grm(packed, n, 1, 0.02, 0.05)
    s, ut64 = pipeline.eigh_from_grm(k, 1e-6, f32_consumer=True)
    model = pipeline.SpectralModel(s, ut64, np.ones((n, 1)), y)
    del ut64, k
    counts = panel.counts()
    keep, af, miss = st.gwas_scan_row_stats(counts, n, 0.02, 0.05, 1.0)
    rows = np.nonzero(keep)[0]
    lut = st.scan_lut_from_counts(af[rows], np.zeros(len(rows), bool), counts[rows], n)
    co = st.warm_chain_offsets(st.warm_chain_blocks_bed(rows, m, 512), len(rows))
    lo, hi = model.null.bounds
    init = min(max(math.log10(model.null.lbd), lo), hi)
    out = pipeline.scan_rows(panel, model, rows, lut, "lmm", max_iter=30, tol=1e-2, init_log10_lbd=init, chain_off=co).cpu().numpy()
    gd = oracle.decode_centered_block_f32(packed.cpu().numpy(), n, np.zeros(m, bool), af, rows=rows)
    grot = oracle.rotate_block_f32(gd, model.ut.cpu().numpy())
    ref = oracle_c.lmm_scan_rotated_chains(grot, model.S.cpu().numpy(), model.xcov.cpu().numpy(), model.y.cpu().numpy(), lo, hi, 30,
                                           1e-2, co, init=init)
    be, se, pe = P._assoc_err(out, ref)
    assert max(be, se, pe) < P.TOL, (be, se, pe)
    res = pipeline.run_gwas(packed, n, y, mode="lmm", warm_chain=(512, 1))
    assert np.array_equal(res.keep, keep)
    be, se, pe = P._assoc_err(res.stats, ref, tag="run_gwas")
    assert max(be, se, pe) < P.TOL, (be, se, pe)


def test_eigh_inplace_mirror_reads_its_argument_only():
    """`rust_eigh_from_array_f64_inplace` (src/math/eigh.rs:1883-1962): the reference reads its argument read-only (:1914) and
    returns new arrays -- the caller's matrix is untouched, a Fortran-ordered input is accepted, the result is the plain entry
    point's."""
    from janusx_amd import janusx as jxrs
    rng = np.random.default_rng(12)
    b = rng.standard_normal((300, 300))
    a = np.asfortranarray(b @ b.T / 300.0)
    keep = a.copy()
    w, v, *_rest = jxrs.rust_eigh_from_array_f64_inplace(a)
    assert np.array_equal(a, keep) and a.flags["F_CONTIGUOUS"]
    w0, v0, *_ = jxrs.rust_eigh_from_array_f64(np.ascontiguousarray(keep))
    # two decompositions of the same matrix: the reduction's atomics reorder sums (eigenvalues to 1e-13, vectors up to sign)
    assert np.abs(w - w0).max() < 1e-12 and np.abs(np.abs(v.T @ v0) - np.eye(300)).max() < 1e-8
    assert np.abs(w - np.linalg.eigvalsh(keep)).max() < 1e-12 * max(1.0, float(np.abs(w).max()))
    assert jxrs.rust_eigh_from_array_f64_inplace(keep, jobz="N")[1] is None


@pytest.mark.parametrize("gm", ["dom", "rec", "HET"])
def test_splmm_approx_route_genetic_models(oracle, tmp_path, gm):
    """`model=` of the SparseLMM entry points (`PackedGeneticModel`, parsed case-insensitively): the approx (GRAMMAR-gamma) route
    decodes the sampled markers and the scanned rows with the model applied to [0 | 2, max(2 maf, 0), 1, 2 | 0] including the
    imputed entry, not centred (`decode_packed_row_model_into_f64`, src/decode/decode.rs:305-364; src/stats/splmm.rs:3211-3262,
    1514-1560) -- against the oracle's restatement of that branch: gamma 1e-6, beta / SE / p 1e-5 on every row whose residual sum
    of squares is not rounding noise.  The exact route refuses a non-additive model with the reference's message (:2662-2664)."""
    from janusx_amd import janusx as jxrs
    P = _parity()
    n, m = 320, 700
    packed, g = P._related_panel(n, m, 37, 0.02)
    prefix = str(tmp_path / "p")
    bim = bed.Bim(["1"] * m, [f"rs{j}" for j in range(m)], list(range(1, m + 1)), ["A"] * m, ["G"] * m)
    bed.write_bed(prefix, packed, [f"s{i}" for i in range(n)], bim)
    rng = np.random.default_rng(12)
    mi, he, ho = oracle.row_counts(packed, n, None)
    keep, miss, af, _, _ = oracle.packed_prep_row_stats(mi, he, ho, n, 0.05, 0.05, 0.0)
    path, _, _ = jxrs.spgrm_packed_to_jxgrm(np.ascontiguousarray(packed[keep]), n, np.zeros(int(keep.sum()), bool), af[keep],
                                            str(tmp_path / "k"), None, 1, 0.05)
    nn, cp, ri, va = oracle.read_sparse_grm_csc(path)
    kd = oracle.sparse_grm_dense_subset(nn, cp, ri, va, None)
    gv = np.where(g < 0, 0, g).astype(np.float64)
    y = gv[40] * 0.5 + gv[300:330].T @ rng.normal(0, 0.2, 30) + rng.normal(0, 1.0, n)
    xc = rng.normal(size=(n, 1))
    xd = oracle.spreml_design_matrix(xc, n)
    rows = np.nonzero(keep)[0][::2].astype(np.int64)
    maf_r, miss_r = af[rows].astype(np.float32), miss[rows].astype(np.float32)
    flip_r = np.zeros(len(rows), dtype=bool)
    flip_r[::5] = True
    lam = 1.3
    got = jxrs.splmm_assoc_pcg_bed(prefix, y, lam, x_cov=xc, maf=maf_r, row_flip=flip_r, row_missing=miss_r, row_indices=rows,
                                   sparse_jxgrm_path=path, rhat_markers=40, scan_mode="approx", model=gm)
    gamma, ref, used, _rr = oracle.splmm_approx_assoc(kd, lam, xd, y, packed[rows], n, maf_r, flip_r, rhat_markers=40, model=gm)
    add = jxrs.splmm_assoc_pcg_bed(prefix, y, lam, x_cov=xc, maf=maf_r, row_flip=flip_r, row_missing=miss_r, row_indices=rows,
                                   sparse_jxgrm_path=path, rhat_markers=40, scan_mode="approx")
    assert abs(got[0] - gamma) < 1e-6 * gamma and got[8] == used, (got[0], gamma)
    out = got[9]
    # rows the model leaves (numerically) constant given X: the reference divides rounding noise there, the kernel reports an invalid row
    codes = oracle.unpack_codes(np.ascontiguousarray(packed[rows]), n)
    solid = np.zeros(len(rows), bool)
    for k in range(len(rows)):
        gk = oracle.splmm_additive_row_f64(codes[k], maf_r[k], bool(flip_r[k]), gm)
        r = gk - xd @ np.linalg.lstsq(xd, gk, rcond=None)[0]
        solid[k] = float(r @ r) > 1e-6 * max(float(gk @ gk), 1.0)
    assert solid.sum() > 0.8 * len(rows)
    ok = solid & ~np.isnan(ref[:, 0])
    assert not np.isnan(out[ok, 0]).any()
    be, se, pe = P._assoc_err(out[ok], ref[ok], tag=gm)
    assert max(be, se, pe) < P.TOL, (gm, be, se, pe)
    assert np.nanmax(np.abs(out[ok, 0] - add[9][ok, 0])) > 1e-3          # the model is really applied
    with pytest.raises(RuntimeError, match="SparseLMM exact denominator mode requires additive model"):
        jxrs.splmm_assoc_pcg_bed(prefix, y, lam, x_cov=xc, maf=maf_r, row_flip=flip_r, row_missing=miss_r, row_indices=rows,
                                 sparse_jxgrm_path=path, scan_mode="exact", model=gm)
    with pytest.raises(RuntimeError, match="model must be one of: add, dom, rec, het"):
        jxrs.splmm_assoc_pcg_bed(prefix, y, lam, maf=maf_r, row_flip=flip_r, row_indices=rows, sparse_jxgrm_path=path,
                                 scan_mode="approx", model="overdominant")


def test_warm_start_chain_two_ranks_share_one_gpu(tmp_path, monkeypatch):
    """`jx gwas -lmm` with the reference's default warm-start chain on two ranks (one device, gloo): `run_trait` deals WHOLE
    chains over the ranks (a chain never starts without its predecessor's state), so with the same kinship file the table is the
    one-rank table bit for bit -- with chunks of 700 rows, and with each chunk cut into 4 pieces."""
    import socket
    import subprocess
    import sys
    monkeypatch.delenv("JX_LMM_UNIFIED_NO_WARM_START", raising=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, m = 600, 3000
    packed, g = bed.synth_panel_numpy(n, m, seed=33, missing_rate=0.003)
    y = bed.synth_phenotype(g, n_causal=20, pve=0.5, seed=33)
    prefix = str(tmp_path / "p")
    ids = [f"s{i}" for i in range(n)]
    bim = bed.Bim(["1"] * m, [f"rs{j}" for j in range(m)], list(range(1, m + 1)), ["A"] * m, ["G"] * m)
    bed.write_bed(prefix, packed, ids, bim)
    with open(prefix + ".pheno", "w") as fh:
        fh.write("id\tt1\n")
        for i in range(n):
            fh.write(f"{ids[i]}\t{float(y[i])!r}\n")
    env = dict(os.environ, JXGPU_DIST_BACKEND="gloo", PYTHONPATH=root)
    env.pop("JX_LMM_UNIFIED_NO_WARM_START", None)

    def run(out, ranks, extra, grm):
        base = [sys.executable]
        if ranks > 1:
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            base += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                     "--master-port", str(port)]
        sub = ["gwas", "-bfile", prefix, "-p", prefix + ".pheno", "-lmm", "-force-model", "-k", grm, "-o", out] + extra
        r = subprocess.run(base + ["-m", "janusx_amd"] + sub, env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (ranks, r.stdout[-1500:], r.stderr[-3000:])

    r0 = subprocess.run([sys.executable, "-m", "janusx_amd", "grm", "-bfile", prefix, "-o", str(tmp_path / "k")], env=env, cwd=root,
                        capture_output=True, text=True, timeout=600)
    assert r0.returncode == 0, r0.stderr[-2000:]
    grm = str(tmp_path / "k") + ".cGRM.npy"
    for tag, extra in (("c700", ["-chunksize", "700"]), ("c700p4", ["-chunksize", "700", "-warm-chain-pieces", "4"])):
        one, two = str(tmp_path / f"one_{tag}"), str(tmp_path / f"two_{tag}")
        run(one, 1, extra, grm)
        run(two, 2, extra, grm)
        a, b = open(one + ".t1.lmm.tsv").read(), open(two + ".t1.lmm.tsv").read()
        assert a == b and a.count("\n") > 2500, tag
    none = str(tmp_path / "none")
    run(none, 1, ["-warm-start", "none"], grm)
    assert open(none + ".t1.lmm.tsv").read() != open(str(tmp_path / "one_c700") + ".t1.lmm.tsv").read()      # the chain is really taken


@pytest.mark.parametrize("q", [4, 6])
def test_warm_start_chain_with_covariates_block_form(oracle, oracle_c, q):
    """The chain scan with q covariates beside the intercept (dim = q + 2 = 6 / 8: the BLOCK form of the evaluation, whose
    objective is interpolated per SNP from one node evaluation per lane -- `blk_objective`, csrc/k_scan_fast.hip) at n = 3000,
    chains of 128 rows seeded with log10 lambda0: beta / SE / p within 1e-5 of the oracle's sequential chains and the same
    number of Brent evaluations on (nearly) every SNP; the per-SNP (no-chain) scan of the same rows as well."""
    import torch
    from janusx_amd import pipeline, stats
    P = _parity()
    n, m = 3000, 600
    packed, g = bed.synth_panel_numpy(n, m, seed=80 + q, missing_rate=0.005, family=True)
    y = bed.synth_phenotype(g, n_causal=15, pve=0.5, seed=80 + q)
    rng = np.random.default_rng(q)
    cov = rng.normal(size=(n, q))
    cov[:, 0] += 0.5 * y
    x = np.concatenate([np.ones((n, 1)), cov], axis=1)
    k, _eff, _ = oracle.grm_stream_bed(packed, n, 1, 0.02, 0.05, 0.0)
    s, u = oracle.gwas_eigh_from_grm(k)
    nm = oracle.spectral_null_model(y, x, s, u)
    mi, he, ho = oracle.row_counts(packed, n)
    keep, maf, miss, flip = oracle.gwas_scan_row_stats(mi, he, ho, n, 0.02, 0.05, 1.0)
    kept = np.nonzero(keep)[0]
    grot = oracle.rotate_block_f32(oracle.decode_centered_block_f32(packed, n, flip, maf, rows=kept), nm.Dh)
    lo, hi = nm.bounds
    init = min(max(math.log10(nm.lbd_null), lo), hi)
    dev = torch.device("cuda", 0)
    panel = pipeline.Panel(torch.from_numpy(packed).to(dev), n)
    model = pipeline.SpectralModel(torch.from_numpy(nm.S).to(dev), torch.from_numpy(nm.Dh.astype(np.float64)).to(dev), x, y)
    lut = stats.scan_lut_from_counts(maf[kept], np.zeros(len(kept), bool), panel.counts()[kept], n)
    sh, xh, yh = model.S.cpu().numpy(), model.xcov.cpu().numpy(), model.y.cpu().numpy()
    co = stats.warm_chain_offsets(stats.warm_chain_blocks_bed(kept, panel.m, 128), len(kept))
    ref, ev_ref = oracle_c.lmm_scan_rotated_chains(grot, sh, xh, yh, lo, hi, 30, 1e-2, co, init=init, return_evals=True)
    out, ev = pipeline.scan_rows(panel, model, kept, lut, "lmm", low=lo, high=hi, max_iter=30, tol=1e-2, init_log10_lbd=init,
                                 return_evals=True, chain_off=co)
    be, se, pe = P._assoc_err(out.cpu().numpy(), ref, tag=f"chain-q{q}")
    assert max(be, se, pe) < P.TOL, (q, be, se, pe)
    assert float(np.mean(ev.cpu().numpy() == ev_ref)) > 0.99
    ref0, ev0 = oracle_c.lmm_scan_rotated_block(grot, sh, xh, yh, lo, hi, 30, 1e-2, return_evals=True)
    out0, e0 = pipeline.scan_rows(panel, model, kept, lut, "lmm", low=lo, high=hi, max_iter=30, tol=1e-2, return_evals=True)
    be, se, pe = P._assoc_err(out0.cpu().numpy(), ref0, tag=f"plain-q{q}")
    assert max(be, se, pe) < P.TOL, (q, be, se, pe)
    assert float(np.mean(e0.cpu().numpy() == ev0)) > 0.99


def test_warm_start_chain_kernels_on_rotated_rows_with_invalid_rows(oracle_c, chain_case):
    """The chain kernels called directly on rotated rows (`jxg_lmm_scan_chain`: tables built inside, series + interpolant form;
    `jxg_lmm_scan_exact_chain`: the reference-formulation kernel): rows with zero variance inside and at the head of a chain are
    (NaN, NaN, 1) and leave the chain's state untouched (the reference returns before it stores the optimum,
    src/stats/lmm.rs:128-132), empty chains, one-row chains, a chain that continues from a carried state, and the carried states
    the kernel hands back = the optimum of each chain's last valid row."""
    import torch
    from janusx_amd._lib import check, lib
    P = _parity()
    c = chain_case
    nm, n = c["nm"], c["n"]
    g = np.ascontiguousarray(c["grot"][:900]).copy()
    g[0] = 0.0            # head of the first chain
    g[257] = 0.0          # inside a chain
    g[300:303] = 0.0      # a run of invalid rows
    co = np.array([0, 200, 200, 201, 520, 900], dtype=np.int64)       # an empty chain and a one-row chain
    lo, hi = nm.bounds
    init = math.log10(nm.lbd_null)
    ref, ev_ref = oracle_c.lmm_scan_rotated_chains(g, nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2, co, init=init, return_evals=True)
    assert np.isnan(ref[0, 0]) and np.isnan(ref[257, 0]) and ref[301, 2] == 1.0
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)      # noqa: E731
    d_g, d_s, d_x, d_y = t(g, torch.float32), t(nm.S), t(nm.Xcov), t(nm.y)
    d_co = torch.from_numpy(co.astype(np.int32)).to(dev)
    p = nm.Xcov.shape[1]
    for fn in ("jxg_lmm_scan_chain", "jxg_lmm_scan_exact_chain"):
        carry = torch.full((len(co) - 1,), init, dtype=torch.float64, device=dev)
        out = torch.zeros((900, 3), dtype=torch.float64, device=dev)
        ev = torch.zeros(900, dtype=torch.int32, device=dev)
        check(getattr(lib(), fn)(d_g.data_ptr(), 900, n, d_s.data_ptr(), d_x.data_ptr(), d_y.data_ptr(), p, lo, hi, 1e-2, 30,
                                 d_co.data_ptr(), len(co) - 1, carry.data_ptr(), 0, 0.0, out.data_ptr(), ev.data_ptr(), st))
        be, se, pe = P._assoc_err(out.cpu().numpy(), ref, tag=fn)
        # identical rotated input: the reference-formulation kernel differs by summation order only; the tabulated / interpolated
        # form evaluates the objective to ~1e-13, which moves a Brent optimum by ~1e-7 and beta with it
        assert max(be, se, pe) < (1e-7 if "exact" in fn else 2e-6), (fn, be, se, pe)
        assert float(np.mean(ev.cpu().numpy() == ev_ref)) >= (1.0 if "exact" in fn else 0.99), fn
        # split in two calls at row 400 (inside the chain [201, 520)): the second call continues from the carried state
        carry2 = torch.full((len(co) - 1,), init, dtype=torch.float64, device=dev)
        out2 = torch.zeros((900, 3), dtype=torch.float64, device=dev)
        a_off = torch.tensor([0, 200, 200, 201, 400], dtype=torch.int32, device=dev)
        b_off = torch.tensor([0, 120, 500], dtype=torch.int32, device=dev)
        check(getattr(lib(), fn)(d_g.data_ptr(), 400, n, d_s.data_ptr(), d_x.data_ptr(), d_y.data_ptr(), p, lo, hi, 1e-2, 30,
                                 a_off.data_ptr(), 4, carry2.data_ptr(), 0, 0.0, out2.data_ptr(), None, st))
        check(getattr(lib(), fn)(d_g[400:].data_ptr(), 500, n, d_s.data_ptr(), d_x.data_ptr(), d_y.data_ptr(), p, lo, hi, 1e-2, 30,
                                 b_off.data_ptr(), 2, carry2[3:].data_ptr(), 0, 0.0, out2[400:].data_ptr(), None, st))
        assert np.array_equal(out2.cpu().numpy(), out.cpu().numpy(), equal_nan=True), fn
        assert torch.equal(carry2, carry)
        if fn == "jxg_lmm_scan_chain":
            # the chain scan in three launches (interpolants of all rows / Brent along the chains / final evaluations of all
            # rows) against the one-kernel form, and with a row forced onto direct evaluations (its chain then takes the
            # one-kernel form, the others the split form): the same bits, evaluation counts and carried states
            for env in ({"JXGPU_SCAN_CHAIN_SPLIT": "0"}, {"JXGPU_SCAN_CHAIN_FORCE_DIRECT": "350"},
                        {"JXGPU_SCAN_CHAIN_FORCE_DIRECT": "5"}):
                old_env = {k: os.environ.get(k) for k in env}
                os.environ.update(env)
                try:
                    carry3 = torch.full((len(co) - 1,), init, dtype=torch.float64, device=dev)
                    out3 = torch.zeros((900, 3), dtype=torch.float64, device=dev)
                    ev3 = torch.zeros(900, dtype=torch.int32, device=dev)
                    check(lib().jxg_lmm_scan_chain(d_g.data_ptr(), 900, n, d_s.data_ptr(), d_x.data_ptr(), d_y.data_ptr(), p, lo, hi,
                                                   1e-2, 30, d_co.data_ptr(), len(co) - 1, carry3.data_ptr(), 0, 0.0,
                                                   out3.data_ptr(), ev3.data_ptr(), st))
                finally:
                    for k, v in old_env.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                assert np.array_equal(out3.cpu().numpy(), out.cpu().numpy(), equal_nan=True), env
                assert torch.equal(ev3, ev) and torch.equal(carry3, carry), env


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,missing", [(4200, 3000, 0.0), (6000, 2500, 0.004)])
def test_int8_rotation_forms_give_the_same_bits(n, m, missing):
    """The two forms of the int8 rotation (k_rotate_i8.hip: operands staged through registers / copied by the LDS DMA with the
    payload decoded in registers) compute the same exact i32 plane sums and share the epilogue: the rotated block is the same
    bits, rows with a missing-call term included.  The switch is read once per process, hence two child processes."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for form in ("0", "1"):
        env = dict(os.environ, JXGPU_ROT_I8_DMA=form)
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "ab_rotate_i8.py"), str(n), str(m), str(missing)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = r.stdout.strip().splitlines()[-1]
        digests[form] = re.search(r"digest=([0-9a-f]+)", line).group(1)
        assert float(re.search(r"\|out\|max=([0-9.]+)", line).group(1)) > 0.1
        if missing > 0:
            assert int(re.search(r"with_missing_term=(\d+)", line).group(1)) > 0
    assert digests["0"] == digests["1"], digests


@pytest.mark.gpu
def test_pcg_operator_forms_agree():
    """The two halves of the matrix-free PCG / HE operator (`jxg_packed_tdot_f32`: Z u over the SNP-major image, `jxg_packed_dot_t32`:
    Z'p over the sample-major image) in their two forms -- int8 MFMA on the bit planes against a four-digit image of the vector
    (k_pcg_i8.hip, the default) and the f32 bit-plane tables (JXGPU_PCG_I8=0) -- against a dense f64 decode: ragged sizes, 3 % missing
    calls, arbitrary per-SNP tables, weights over four decades, and a row list.  The int8 form is exact in its sums (what is left is
    the f32 rounding of the vector / of the per-SNP weights the reference's f32 operator carries as well)."""
    import torch
    import bench
    from janusx_amd import pipeline
    from janusx_amd._lib import check, lib
    n, m = 5003, 20011
    dev = torch.device("cuda", 0)
    packed, _ = bench.synth_panel_gpu(n, m, 7, dev, missing_rate=0.03)
    p = pipeline.Panel(packed, n, None)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    lut = torch.randn((m, 4), device=dev, dtype=torch.float32, generator=g)
    beta = torch.randn(m, device=dev, dtype=torch.float64, generator=g) * torch.exp(2 * torch.randn(m, device=dev, dtype=torch.float64, generator=g))
    alpha = torch.randn(n, device=dev, dtype=torch.float64, generator=g)
    pk = packed.to(torch.int64)
    codes = torch.stack([(pk >> (2 * k)) & 3 for k in range(4)], dim=2).reshape(m, -1)[:, :n]
    z = torch.gather(lut.to(torch.float64), 1, codes)
    ref_t = z @ alpha.to(torch.float32).to(torch.float64)
    ref_d = torch.gather((lut * beta.to(torch.float32)[:, None]).to(torch.float64), 1, codes).sum(0)
    rows = torch.arange(m - 1, -1, -3, device=dev, dtype=torch.int32)[:6000].contiguous()         # a descending row list
    ref_tr = ref_t[rows.long()]
    t32 = torch.empty(int(lib().jxg_t32_bytes(n, m)), dtype=torch.uint8, device=dev)
    work = torch.empty(16 * m + 16, dtype=torch.uint8, device=dev)
    check(lib().jxg_p32_transpose(p.p32.data_ptr(), p.m, n, None, m, t32.data_ptr(), st))
    errs = {}
    old = os.environ.get("JXGPU_PCG_I8")
    try:
        for form in ("1", "0"):
            os.environ["JXGPU_PCG_I8"] = form
            om = torch.full((m,), float("nan"), device=dev, dtype=torch.float64)
            on = torch.full((n,), float("nan"), device=dev, dtype=torch.float64)
            orr = torch.full((len(rows),), float("nan"), device=dev, dtype=torch.float64)
            check(lib().jxg_packed_tdot_f32(p.p32.data_ptr(), p.m, n, None, m, lut.data_ptr(), alpha.data_ptr(), om.data_ptr(), st))
            check(lib().jxg_packed_dot_t32(t32.data_ptr(), n, m, lut.data_ptr(), beta.data_ptr(), work.data_ptr(), on.data_ptr(), st))
            check(lib().jxg_packed_tdot_f32(p.p32.data_ptr(), p.m, n, rows.data_ptr(), len(rows), lut[rows.long()].contiguous().data_ptr(),
                                            alpha.data_ptr(), orr.data_ptr(), st))
            torch.cuda.synchronize()
            errs[form] = (float((ref_t - om).abs().max() / ref_t.abs().max()), float((ref_d - on).abs().max() / ref_d.abs().max()),
                          float((ref_tr - orr).abs().max() / ref_tr.abs().max()))
    finally:
        if old is None:
            os.environ.pop("JXGPU_PCG_I8", None)
        else:
            os.environ["JXGPU_PCG_I8"] = old
    assert errs["1"][0] < 5e-9 and errs["1"][2] < 5e-9 and errs["1"][1] < 2e-7, errs
    assert max(errs["0"]) < 2e-6, errs


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["sorted80", "shuffled_windows", "sparse10", "duplicates"])
def test_repack_of_a_sample_subset_window_form(case):
    """`jxg_repack_p32` with a sample list: the window form (one descriptor per output dword position, three dword loads of the row
    and 16 shifts of a 64-bit window; k_pack.hip) writes the same image as the per-code gather (JXGPU_REPACK_WINDOW=0) and as a numpy
    decode -- rows that are not 4-byte aligned (n_src = 5003), a row list, a ragged last tile, subsets too sparse for a window (those
    dwords keep the gather), samples out of order and repeated."""
    import torch
    from janusx_amd._lib import check, lib
    rng = np.random.default_rng(11)
    n_src, m = 5003, 700
    bps = (n_src + 3) // 4
    packed = rng.integers(0, 256, size=(m, bps), dtype=np.uint8)
    if case == "sorted80":
        idx = np.sort(rng.choice(n_src, 4001, replace=False))
    elif case == "shuffled_windows":
        idx = np.sort(rng.choice(n_src, 4000, replace=False)).reshape(-1, 8)
        idx = np.stack([rng.permutation(r) for r in idx]).ravel()
    elif case == "sparse10":
        idx = np.sort(rng.choice(n_src, 517, replace=False))
    else:
        idx = np.sort(rng.integers(0, n_src, size=3000))
    idx = idx.astype(np.int32)
    rows = rng.permutation(m)[:650].astype(np.int64)
    n_sel = len(idx)
    nt = (n_sel + 127) // 128
    codes = (packed[rows][:, idx >> 2] >> (2 * (idx & 3))[None, :]) & 3                    # (650, n_sel)
    full = np.ones((len(rows), nt * 128), dtype=np.uint8)                                   # padding: code 01
    full[:, :n_sel] = codes
    ref = np.zeros((nt, len(rows), 32), dtype=np.uint8)
    for k in range(4):
        ref |= (full[:, k::4].reshape(len(rows), nt, 32).transpose(1, 0, 2) << (2 * k)).astype(np.uint8)
    dev = torch.device("cuda", 0)
    d_p, d_i, d_r = (torch.from_numpy(a).to(dev) for a in (packed, idx, rows))
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    old = os.environ.get("JXGPU_REPACK_WINDOW")
    try:
        for form in ("1", "0"):
            os.environ["JXGPU_REPACK_WINDOW"] = form
            out = torch.zeros((nt, len(rows), 32), dtype=torch.uint8, device=dev)
            check(lib().jxg_repack_p32(d_p.data_ptr(), bps, n_src, m, d_i.data_ptr(), n_sel, d_r.data_ptr(), len(rows), out.data_ptr(), st))
            torch.cuda.synchronize()
            outs[form] = out.cpu().numpy()
    finally:
        if old is None:
            os.environ.pop("JXGPU_REPACK_WINDOW", None)
        else:
            os.environ["JXGPU_REPACK_WINDOW"] = old
    assert np.array_equal(outs["0"], ref)
    assert np.array_equal(outs["1"], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("m", [4000, 7013])
def test_rank_2k_update_forms_give_the_same_bits(m):
    """The trailing update of the band reduction (`jxg_dsyr2k_lower_nt_f64`: lower tiles of C -= A B', K = 128) as one stream of K
    steps over a workgroup's tiles (dsyr2k_pipe_kernel: operands one step ahead across tile boundaries, the C values of a tile
    requested a strip per step, tiles dealt by XCD) against one tile per workgroup (JXGPU_SYR2K_PIPE=0): the same products in the same
    order -- the lower triangle is the same bits, the strict upper triangle is not touched, and both agree with an f64 reference."""
    import torch
    from janusx_amd._lib import check, lib
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev)
    g.manual_seed(m)
    a = torch.randn((128, m), device=dev, dtype=torch.float64, generator=g)          # column-major (m, 128)
    b = torch.randn((128, m), device=dev, dtype=torch.float64, generator=g)
    c0 = torch.randn((m, m), device=dev, dtype=torch.float64, generator=g)
    outs = {}
    old = os.environ.get("JXGPU_SYR2K_PIPE")
    try:
        for form in ("1", "0"):
            os.environ["JXGPU_SYR2K_PIPE"] = form
            c = c0.clone()
            check(lib().jxg_dsyr2k_lower_nt_f64(m, 128, -1.0, a.data_ptr(), m, b.data_ptr(), m, 1.0, c.data_ptr(), m, st))
            torch.cuda.synchronize()
            outs[form] = c
    finally:
        if old is None:
            os.environ.pop("JXGPU_SYR2K_PIPE", None)
        else:
            os.environ["JXGPU_SYR2K_PIPE"] = old
    assert torch.equal(outs["1"], outs["0"])
    # memory is column-major: the tensor's [j, i] is C[i][j]; lower triangle of C = upper triangle of the tensor
    ref = c0 - b.T @ a                                                                # C^T = C0^T - B A'
    up = torch.triu(torch.ones((m, m), dtype=torch.bool, device=dev))
    assert float((outs["1"] - ref)[up].abs().max()) < 1e-11
    assert torch.equal(outs["1"][~up], c0[~up])


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,missing", [(2700, 3000, 0.0), (4200, 2500, 0.002)])
def test_grm_count_gram_on_the_fp4_pipes(n, m, missing):
    """JXGPU_GRM_FP4=1: the count Gram of SNPs without missing calls as v_mfma_scale_f32_32x32x64_f8f6f4 products of an e2m1 nibble
    image of the counts (k_grm_fp4.hip; f32 sums of small integers, exact below 2^24) against the int8 kernel (exact i32 sums): the
    accumulators agree to the rounding of the affine terms both add in f64 (their own sums are order-dependent f64 atomics), ragged
    sample counts, a few SNPs with missing calls beside (those take their own path in both)."""
    import torch
    import bench
    from janusx_amd import pipeline, stats as st
    dev = torch.device("cuda", 0)
    packed, _ = bench.synth_panel_gpu(n, m, 5, dev, missing_rate=missing)
    p = pipeline.Panel(packed, n, None)
    keep, mean_g, scale, flip, var = st.stream_grm_row_prepare(p.counts(), n, 1, 0.02, 0.05, 0.0)
    rows = np.nonzero(keep)[0]
    lut = st.grm_lut_from_mean_scale(mean_g[rows], scale[rows], flip[rows])
    outs = {}
    old = {k: os.environ.get(k) for k in ("JXGPU_GRM_FP4", "JXGPU_GRM_I8_TILE")}
    try:
        os.environ["JXGPU_GRM_I8_TILE"] = "256"                  # the 256-tile form at a test size
        for form in ("0", "1"):
            os.environ["JXGPU_GRM_FP4"] = form
            acc = torch.zeros((p.npad, p.npad), dtype=torch.float64, device=dev)
            pipeline.grm_accumulate(p, rows, lut, acc=acc)
            torch.cuda.synchronize()
            outs[form] = torch.tril(acc[:n, :n]).cpu().numpy()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    scale_k = float(np.abs(outs["0"]).max())
    assert scale_k > 1.0
    assert float(np.abs(outs["1"] - outs["0"]).max()) <= 1e-12 * scale_k
