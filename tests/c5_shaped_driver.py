"""BASELINE configs[4]-shaped `-splmm` run with the panel GENERATED ON THE DEVICE (run as its own process by
tests/test_gpu_parity.py::test_c5_full_size_splmm_device_panel, so that the peak host RSS it reports is this run's alone).

n samples in sibships of four, m SNPs; sparse GRM through the row-panel builder (`jxg_grm_accumulate_rows`, the accumulator of
n = 200 000 does not fit HBM as a square), block-diagonal spectral route, SparseLMM exact scan.  No (m x n) or packed
(m x n / 4) array ever exists on the host.  Checks a 150-SNP sample against the oracle's restatement of
`exact_scan_blocks_core` (src/stats/splmm.rs:2567-2880) with a sparse factorisation of K + lambda I read back from the
`.spgrm` file, and the sparse REML optimum against the oracle's evaluation at that lambda.  Prints one JSON line.

    python tests/c5_shaped_driver.py N M [SAMPLE]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _rss_gib():
    """(resident now, peak so far) in GiB, from VmRSS / VmHWM of this process image (ru_maxrss would also carry the peak of
    the parent that forked us: the pytest process with its full-size host arrays)."""
    now = peak = 0.0
    with open("/proc/self/status") as f:
        for ln in f:
            if ln.startswith("VmRSS:"):
                now = int(ln.split()[1]) / 2**20
            elif ln.startswith("VmHWM:"):
                peak = int(ln.split()[1]) / 2**20
    return round(now, 3), round(peak, 3)


def main():
    import torch
    import bench
    from janusx_amd import janusx as jxrs
    from janusx_amd import stats as st
    from janusx_amd._lib import lib
    from oracle import jx_oracle as O
    import scipy.sparse as sp
    n, m = int(sys.argv[1]), int(sys.argv[2])
    n_pick = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    dev = torch.device("cuda", 0)
    # baseline of the process before any panel exists: interpreter + torch + the HIP runtime with its code objects loaded
    lib()
    (torch.ones(8, device=dev) @ torch.ones(8, device=dev)).item()
    torch.cuda.synchronize()
    rss = {"init": _rss_gib()}
    t0 = time.perf_counter()
    packed_t, dos = bench.family_panel_gpu(n, m, 4, 11, dev)
    y = bench.make_phenotype(dos, n, 7, dev)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    rss["generated"] = _rss_gib()
    counts = jxrs.bed_row_counts(packed_t, n)
    rss["counted"] = _rss_gib()
    keep, _miss, maf, _std = st.packed_prep_row_stats(counts, n, 0.02, 0.05, 0.0)
    pk = packed_t if bool(keep.all()) else packed_t[torch.from_numpy(np.nonzero(keep)[0]).to(dev)]
    del packed_t
    maf_k = maf[keep]
    flip = np.zeros(len(maf_k), dtype=bool)
    res = {"n": n, "m": m, "m_kept": int(len(maf_k)), "gen_s": t_gen}
    td = tempfile.mkdtemp()
    import atexit
    import shutil
    atexit.register(shutil.rmtree, td, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    path, nn, nnz = jxrs.spgrm_packed_to_jxgrm(pk, n, flip, maf_k, os.path.join(td, "k"), None, 1, 0.05)
    path_keep = path
    t1 = time.perf_counter()
    rss["sparse_grm"] = _rss_gib()
    out, l10, null = jxrs.splmm_exact_scan_from_jxgrm(path, y, pk, n, maf_k, flip)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rss["scanned"] = _rss_gib()
    res.update(spgrm_s=t1 - t0, scan_s=t2 - t1, nnz=int(nnz), log10_lambda=float(l10),
               snps_per_s=len(maf_k) / (t2 - t0), route="block" if jxrs._sparse_block_route(n) else "dense")
    # ---- checker: the oracle on the sparse GRM file (a sparse LU of K + lambda I is the reference's sparse LLT restated)
    _n, cp, ri, va = O.read_sparse_grm_csc(path)
    low = sp.csc_matrix((va, ri.astype(np.int64), cp.astype(np.int64)), shape=(n, n))
    ksym = (low + sp.tril(low, -1).T).tocsc()
    # structure: families of four -> every sample related to at most three others above the cut-off
    per_col = np.diff(ksym.indptr)
    res["max_relatives"] = int(per_col.max()) - 1
    lam = 10.0 ** l10
    x = np.ones((n, 1))
    pick = np.sort(np.random.default_rng(0).choice(len(maf_k), n_pick, replace=False))
    pk_small = pk[torch.from_numpy(pick).to(dev)].cpu().numpy()
    ref = O.splmm_exact_scan(ksym, lam, x, y, pk_small, n, maf_k[pick], flip[pick])
    got = out[pick]
    ok = ~np.isnan(ref[:, 0])
    res["nan_pattern_equal"] = bool(np.array_equal(np.isnan(got[:, 0]), ~ok))
    res["se_err"] = float(np.max(np.abs(got[ok, 1] - ref[ok, 1]) / ref[ok, 1]))
    res["beta_err"] = float(np.max(np.abs(got[ok, 0] - ref[ok, 0]) / np.maximum(np.abs(ref[ok, 0]), ref[ok, 1])))
    z2 = (ref[ok, 0] / ref[ok, 1]) ** 2
    res["p_err"] = float(max(np.max(np.abs(got[ok, 2] - ref[ok, 2]) / ref[ok, 2] / np.maximum(1.0, z2)),
                             np.max(np.abs(np.log10(got[ok, 2]) - np.log10(ref[ok, 2])) /
                                    np.maximum(1.0, -np.log10(ref[ok, 2])))))
    # sparse REML at the optimum the GPU path found: the oracle's evaluation (dense form per the reference's own test
    # spreml.rs:1209-1329, here through the sparse factor) must give the same likelihood
    ev = O.spreml_evaluate(ksym, x, y, l10)
    # null 10-tuple: (lambda, sigma_g2, sigma_e2, ml, reml, log10_lambda, ...)
    res["reml_err"] = float(abs(ev["reml"] - null[4]) / max(1.0, abs(ev["reml"])))
    res["ml_err"] = float(abs(ev["ml"] - null[3]) / max(1.0, abs(ev["ml"])))
    res["all_rows_finite_p"] = bool(np.all((out[:, 2] > 0) & (out[:, 2] <= 1)))
    # ---- the reference's DEFAULT `-splmm` route on the same panel: fastGWA fixed-Vp null + residualised GRAMMAR-gamma scan
    # (`splmm_assoc_pcg_bed`, scan_mode "approx"; 300 sampled markers here instead of the workflow's 1000 to bound the host
    # memory of the checker, which decodes them in f64)
    jxrs.spectral_cache_clear()          # the approx leg pays for its own block eigendecompositions (the null fit builds them,
    torch.cuda.synchronize()             # the scan call reuses them: one-entry cache of the spectral form)
    t3 = time.perf_counter()
    yc = y - y.mean()
    vp = float(yc @ yc) / float(n - 1)
    nullf = jxrs.spreml_sparse_fastgwa_fixed_vp_brent_from_jxgrm(path_keep, yc, vp, low=-5.0, high=5.0, grid_size=17, tol=1e-3,
                                                                 max_iter=20)
    t4 = time.perf_counter()
    lam_a = float(nullf[0])
    n_rhat = 300
    ga = jxrs.splmm_assoc_pcg_bed("device", y, lam_a, packed=pk, packed_n_samples=n, maf=maf_k, row_flip=flip,
                                  sparse_jxgrm_path=path_keep, rhat_markers=n_rhat, scan_mode="approx")
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    out_a = ga[9]
    res.update(approx_null_s=t4 - t3, approx_scan_s=t5 - t4, approx_lambda=lam_a, approx_gamma=float(ga[0]),
               approx_markers_used=int(ga[8]), approx_snps_per_s=len(maf_k) / ((t1 - t0) + (t5 - t3)))
    rr = O.choose_rhat_rows(len(maf_k), n_rhat, 20260527)
    fac, _yr, a_vec, sigma2 = O.splmm_approx_null(ksym, x, y, lam_a)
    codes = O.unpack_codes(pk[torch.from_numpy(rr).to(dev)].cpu().numpy(), n)
    markers = np.stack([O.splmm_additive_row_f64(codes[i], maf_k[r], bool(flip[r])) for i, r in enumerate(rr)])
    del codes
    gamma_ref, used_ref = O.splmm_estimate_gamma(fac, x, markers, a_vec, n_rhat, 1.0 / sigma2)
    del markers
    a_resid = O.splmm_residualize(x, O.splmm_xtx_chol(x), a_vec)
    # the scan's dots: f32 operands; the GPU accumulates them in f64 and rounds once, a BLAS sgemm accumulates in f32 in its own
    # order (at n = 200 000 that alone is worth 1e-5 of a standard error: recorded as approx_sgemm_order_effect)
    ref_a = O.splmm_grammar_scan(pk_small, n, maf_k[pick], flip[pick], x, a_resid, gamma_ref, exact_dots=True)
    ref_f32 = O.splmm_grammar_scan(pk_small, n, maf_k[pick], flip[pick], x, a_resid, gamma_ref)
    okf = ~np.isnan(ref_a[:, 0]) & ~np.isnan(ref_f32[:, 0])
    res["approx_sgemm_order_effect"] = float(np.max(np.abs(ref_f32[okf, 0] - ref_a[okf, 0]) / np.maximum(np.abs(ref_a[okf, 0]), ref_a[okf, 1])))
    got_a = out_a[pick]
    oka = ~np.isnan(ref_a[:, 0])
    res["approx_gamma_err"] = float(abs(ga[0] - gamma_ref) / gamma_ref)
    res["approx_used_equal"] = bool(int(ga[8]) == int(used_ref))
    res["approx_nan_pattern_equal"] = bool(np.array_equal(np.isnan(got_a[:, 0]), ~oka))
    res["approx_se_err"] = float(np.max(np.abs(got_a[oka, 1] - ref_a[oka, 1]) / ref_a[oka, 1]))
    res["approx_beta_err"] = float(np.max(np.abs(got_a[oka, 0] - ref_a[oka, 0]) / np.maximum(np.abs(ref_a[oka, 0]), ref_a[oka, 1])))
    z2a = (ref_a[oka, 0] / ref_a[oka, 1]) ** 2
    res["approx_p_err"] = float(max(np.max(np.abs(got_a[oka, 2] - ref_a[oka, 2]) / ref_a[oka, 2] / np.maximum(1.0, z2a)),
                                    np.max(np.abs(np.log10(got_a[oka, 2]) - np.log10(ref_a[oka, 2])) /
                                           np.maximum(1.0, -np.log10(ref_a[oka, 2])))))
    res["peak_hbm_gib"] = torch.cuda.max_memory_allocated() / 2**30
    rss["checked"] = _rss_gib()
    res["host_rss_gib"] = rss
    res["host_maxrss_gib"] = rss["checked"][1]
    # what the run added on top of the process baseline (interpreter + torch + HIP runtime with its code objects loaded)
    res["host_rss_growth_gib"] = res["host_maxrss_gib"] - rss["init"][1]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
