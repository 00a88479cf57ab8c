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


def test_end_to_end_two_stage(oracle, oracle_c):
    """A TRUE end-to-end leg above the two-stage threshold: n = 5000, m = 20 000 through `pipeline.run_gwas` (GRM on the int8
    pipes, the own two-stage eigensolver with Q1 and the divide-and-conquer merges on 5 digit planes -- sliced products engage
    from n = 3000 -- f32 U^T, null, exact-scan / fixed-lambda scan) against an oracle that builds its OWN GRM (f32 SYRK + f64
    merge), its OWN dsyevd, null fit, f32 rotation and scan (as `test_pipeline_end_to_end` does at n = 400, below every sliced
    product).  Reference contract: python/janusx/assoc/workflow.py:5639-5641 (ridge 1e-6, f64 eigh), src/stats/reml.rs:109-198
    (U^T kept f32).  Bars: the north star's 1e-5 on beta / SE / p, kept set and af / miss bit-exact, lambda 1e-5."""
    import torch
    from janusx_amd import pipeline
    P = _parity()
    n, m = 5000, 20000
    packed, g = bed.synth_panel_numpy(n, m, seed=61, missing_rate=0.002)
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
        be, se, pe = P._assoc_err(res.stats[pick], ref, tag=mode)
        assert max(be, se, pe) < P.TOL, (mode, be, se, pe)
