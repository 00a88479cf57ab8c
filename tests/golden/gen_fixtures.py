#!/usr/bin/env python3
"""Generates the committed golden fixtures (inputs + expected outputs) for the mixed-model hot path.

Run in the BUILD container only:  python tests/golden/gen_fixtures.py
  * expected outputs come from the oracle (oracle/jx_oracle.py + liboracle.so), which restates the reference;
  * where the reference's own Python can run here it is used to cross-check the oracle BEFORE the fixture is
    written: /root/reference/python/janusx/pyBLUP/assoc.py is imported with a stub `janusx.janusx` module (the
    native extension cannot be built: no Rust toolchain) and its pure-numpy helpers `_lmm_profile_exact_vc` and
    `_chi2_sf_df1` are evaluated on the same inputs; /root/reference/python/janusx/pyBLUP/QK2.py (legacy numpy GRM)
    is used as a structural sanity check of ZZ^T / sum(2pq);
  * round 3: further reference-produced values are STORED in the fixture (and asserted by the CPU and GPU tests):
    `pyBLUP/blup.py::REML` (dense-Cholesky restricted likelihood at 5 lambda), `pyBLUP/mlm.py::BLUP._REML` (spectral GBLUP
    likelihood, incl. its v_floor branch), `pyBLUP/assoc.py::_lm_plrt_from_beta_se` and `_lm_precompute_ixx_qr` (full-rank
    and rank-deficient design).
Nothing under /root/reference is copied: the fixtures hold data only.
"""
import importlib
import io
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import jx_oracle as O  # noqa: E402
from oracle import jx_oracle_c as OC  # noqa: E402
from janusx_amd import bed  # noqa: E402

REF_PY = "/root/reference/python"


def reference_helpers():
    """Import the reference's pyBLUP.assoc with a stubbed native module; returns the module or None."""
    if not os.path.isdir(REF_PY):
        return None
    sys.path.insert(0, REF_PY)

    class _Stub(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return None

    sys.modules["janusx.janusx"] = _Stub("janusx.janusx")
    try:
        return importlib.import_module("janusx.pyBLUP.assoc")
    except Exception as e:  # pragma: no cover
        print("reference python import failed:", repr(e))
        return None


def main():
    rng = np.random.default_rng(20260609)
    n, m = 61, 160
    packed, g = bed.synth_panel_numpy(n, m, seed=42, missing_rate=0.03)
    g[0, :] = -9          # all missing
    g[1, :] = 0           # monomorphic ref
    g[2, :] = 2           # monomorphic alt
    g[3, :] = 1           # all het
    g[4, : n // 2] = -9   # high missingness
    g[5, :] = np.where(rng.random(n) < 0.9, 2, 1)  # alt_freq > 0.5 -> flip in the stream GRM
    packed = bed.pack_dosage(g)
    y = bed.synth_phenotype(g, n_causal=8, pve=0.6, seed=42)
    x = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, 1))], axis=1)

    mi, he, ho = O.row_counts(packed, n)
    counts = np.stack([mi, he, ho], 1).astype(np.int32)
    keep, af, miss, flip = O.gwas_scan_row_stats(mi, he, ho, n, 0.02, 0.05, 1.0)
    gkeep, gmean, gscale, gflip, gvar = O.stream_grm_row_prepare(mi, he, ho, n, 1, 0.02, 0.05, 0.0)
    k1, eff1, keep1 = O.grm_stream_bed(packed, n, 1, 0.02, 0.05, 0.0)
    k2, eff2, keep2 = O.grm_stream_bed(packed, n, 2, 0.02, 0.05, 0.0)
    pk = np.ascontiguousarray(packed[keep])
    sub = np.sort(rng.permutation(n)[:40])
    kp1, d1 = O.grm_packed(pk, n, flip[keep], af[keep], None, 1)
    kp1s, d1s = O.grm_packed(pk, n, flip[keep], af[keep], sub, 1)
    kp2, d2 = O.grm_packed(pk, n, flip[keep], af[keep], None, 2)

    s, u = O.gwas_eigh_from_grm(k1)
    nm = O.spectral_null_model(y, x, s, u)
    rows = np.nonzero(keep)[0]
    gd = O.decode_centered_block_f32(packed, n, flip, af, rows=rows)
    grot = O.rotate_block_f32(gd, nm.Dh)
    lmm, evals = OC.lmm_scan_rotated_block(grot, nm.S, nm.Xcov, nm.y, nm.bounds[0], nm.bounds[1], 30, 1e-2,
                                           return_evals=True)
    lmm_py = O.lmm_scan_rotated_block(grot, nm.S, nm.Xcov, nm.y, nm.bounds[0], nm.bounds[1], 30, 1e-2)
    assert np.nanmax(np.abs(lmm - lmm_py) / np.maximum(np.abs(lmm_py), 1e-300)) < 1e-7
    fv = O.fvlmm_assoc_rotated_block(grot, O.fvlmm_prepare_cache(nm.S, nm.Xcov, nm.y, nm.lbd_null))

    # ---- cross-checks against the reference's own Python (this container only) -------------------
    ref = reference_helpers()
    ref_used = []
    ref_lams = [0.01, 0.3, 1.0, 7.5, 200.0]
    ref_nullreml = []
    if ref is not None:
        sg2, se2 = ref._lmm_profile_exact_vc(nm.S, nm.Xcov, nm.y, nm.lbd_null)
        assert abs(sg2 - nm.sigma_g2) <= 1e-12 * abs(sg2) and abs(se2 - nm.sigma_e2) <= 1e-12 * abs(se2)
        ref_used.append("_lmm_profile_exact_vc")
        for st in (0.5, 3.84, 30.0, 81.8):
            assert abs(ref._chi2_sf_df1(st) - O.chi2_sf_df1(st)) <= 1e-15 + 1e-12 * O.chi2_sf_df1(st)
        ref_used.append("_chi2_sf_df1")
        # the reference's own pure-numpy null REML (assoc.py:1917, no 1e-6 ridge): golden values for the
        # likelihood formula (constants, log-determinant terms) produced by the reference itself
        obj = ref.LMM.__new__(ref.LMM)
        obj.Xcov, obj.S, obj.y = nm.Xcov, nm.S, nm.y.reshape(-1, 1)
        for lam in ref_lams:
            val = float(obj._NULLREML(lam))
            ref_nullreml.append(val)
            mine = O.reml_loglike(math.log10(lam), nm.S, nm.Xcov, nm.y, None)
            assert abs(val - mine) < 1e-6 * max(1.0, abs(val)), (lam, val, mine)
        ref_used.append("LMM._NULLREML")
        try:
            qk = importlib.import_module("janusx.pyBLUP.QK2")
            gg = np.where(g[keep1] < 0, np.nan, g[keep1]).astype(float)
            # structural sanity only (different MAF estimator / imputation): correlation of entries
            kq = qk.GRM(np.nan_to_num(gg, nan=0.0)) if hasattr(qk, "GRM") else None
            if kq is not None and kq.shape == k1.shape:
                c = np.corrcoef(kq.ravel(), k1.ravel())[0, 1]
                assert c > 0.98, c
                ref_used.append(f"QK2.GRM corr={c:.4f}")
        except Exception as e:
            print("QK2 sanity skipped:", repr(e))
    # ---- round 3: more reference-produced values, stored -----------------------------------------
    extra = {}
    lm_x = x
    lm_ixx = O.lm_precompute_ixx_qr(lm_x)
    flip0 = np.zeros(int(keep.sum()), dtype=bool)
    lm_out = O.lm_block_assoc_packed(y, lm_x, lm_ixx, pk, n, flip0, af[keep])
    if ref is not None:
        blup_mod = importlib.import_module("janusx.pyBLUP.blup")
        mlm_mod = importlib.import_module("janusx.pyBLUP.mlm")
        # (a) dense restricted likelihood of blup.py (scale invariant in theta: V = K + lambda I), on the ridged GRM the
        #     spectral null model decomposes: must equal -reml_loglike of the spectral form up to its 1e-6 ridge on X'V^-1X
        kr = k1.astype(np.float64) + 1e-6 * np.eye(n)
        dense = []
        for lam in ref_lams:
            val = float(blup_mod.REML(np.array([1.0, lam]), y.reshape(-1, 1), x, [kr]))
            dense.append(val)
            mine = -O.reml_loglike(math.log10(lam), nm.S, nm.Xcov, nm.y, None)
            assert abs(val - mine) < 1e-6 * max(1.0, abs(val)), (lam, val, mine)
        extra["ref_dense_reml"] = np.array(dense)
        ref_used.append("blup.REML")
        # (b) spectral GBLUP likelihood of mlm.py on the training spectrum of the GBLUP fit (K + g_eps I), incl. a lambda
        #     where an (artificially negative) eigenvalue hits the v_floor branch
        g_eps = 1e-8
        sg, ug = O.eigh_sym(k1.astype(np.float64) + g_eps * np.eye(n))
        yc = y - float(np.sum(y) / n)
        x_rot, y_rot = ug.sum(axis=0), ug.T @ yc
        obj = mlm_mod.BLUP.__new__(mlm_mod.BLUP)
        obj._reml_calls, obj._debug_stage, obj._reml_v_floor = 0, False, 1e-12
        obj.X, obj.y, obj.p = x_rot.reshape(-1, 1), y_rot.reshape(-1, 1), 1
        gb = []
        for lam in ref_lams:
            obj.S = sg
            val = float(obj._REML(lam))
            gb.append(val)
            mine = O.gblup_reml_eval(sg, x_rot, y_rot, n, math.log10(lam))[0]
            assert abs(val - mine) < 1e-10 * max(1.0, abs(val)), (lam, val, mine)
        s_floor = sg.copy()
        s_floor[0] = -0.3                                   # s + lambda <= 1e-12 at lambda = 0.3: clipped to the floor
        obj.S = s_floor
        val = float(obj._REML(0.3))
        mine = O.gblup_reml_eval(s_floor, x_rot, y_rot, n, math.log10(0.3))[0]
        assert abs(val - mine) < 1e-10 * max(1.0, abs(val)), (val, mine)
        extra.update(ref_gblup_reml=np.array(gb), ref_gblup_reml_floor=np.array(val), gblup_s=sg, gblup_s_floor=s_floor,
                     gblup_xrot=x_rot, gblup_yrot=y_rot)
        ref_used.append("mlm.BLUP._REML")
        # (c) LM helpers of assoc.py
        df_lm = n - lm_x.shape[1] - 1
        plrt_ref = ref._lm_plrt_from_beta_se(lm_out[:, 0], lm_out[:, 1], n_obs=n, df=df_lm)
        okr = np.isfinite(plrt_ref)
        assert np.array_equal(okr, np.isfinite(lm_out[:, 3]))
        assert np.max(np.abs(plrt_ref[okr] - lm_out[okr, 3]) / plrt_ref[okr]) < 1e-12
        ixx_ref = ref._lm_precompute_ixx_qr(lm_x)
        assert np.max(np.abs(ixx_ref - lm_ixx)) < 1e-14 * np.max(np.abs(ixx_ref))
        x_def = np.concatenate([lm_x, lm_x[:, 1:2] * 2.0], axis=1)          # rank deficient: pseudo-inverse branch
        ixx_def_ref = ref._lm_precompute_ixx_qr(x_def)
        assert np.max(np.abs(ixx_def_ref - O.lm_precompute_ixx_qr(x_def))) < 1e-12 * np.max(np.abs(ixx_def_ref))
        extra.update(ref_lm_plrt=plrt_ref, ref_lm_ixx=ixx_ref, ref_lm_ixx_deficient=ixx_def_ref, lm_x_deficient=x_def)
        ref_used += ["_lm_plrt_from_beta_se", "_lm_precompute_ixx_qr"]
    extra.update(lm_out=lm_out, lm_pk=pk, lm_maf=af[keep])
    print("reference helpers cross-checked:", ref_used)

    # ---- TSV text ---------------------------------------------------------------------------------
    buf = io.StringIO()
    buf.write(O.TSV_HEADER)
    for i, j in enumerate(rows[:12]):
        buf.write(O.format_assoc_row("1", 1000 + int(j), f"snp{j}" if j % 3 else ".", "A", "G", af[j], miss[j],
                                     lmm[i, 0], lmm[i, 1], lmm[i, 2]))

    np.savez_compressed(
        os.path.join(HERE, "panel_small.npz"),
        n=n, m=m, packed=packed, y=y, x=x, sub=sub, counts=counts,
        keep=keep, af=af, miss=miss,
        gkeep=gkeep, gmean=gmean, gscale=gscale, gflip=gflip, gvar=gvar,
        k_stream_m1=k1, k_stream_m2=k2, eff_m=np.array([eff1, eff2]),
        k_packed_m1=kp1, k_packed_m1_sub=kp1s, k_packed_m2=kp2, denom=np.array([d1, d1s, d2]),
        S=nm.S, lbd=nm.lbd_null, ml0=nm.ML0, reml0=nm.LL0, sg2=nm.sigma_g2, se2=nm.sigma_e2, pve=nm.pve,
        bounds=np.array(nm.bounds), Dh=nm.Dh, Xcov=nm.Xcov, yrot=nm.y,
        grot=grot, lmm=lmm, lmm_evals=evals, fvlmm=fv, tsv=np.array(buf.getvalue()),
        reference_checked=np.array(";".join(ref_used)),
        ref_lams=np.array(ref_lams), ref_nullreml=np.array(ref_nullreml),
        **extra,
    )
    sz = os.path.getsize(os.path.join(HERE, "panel_small.npz"))
    print("wrote panel_small.npz", sz, "bytes; mean Brent evals", evals[evals > 0].mean())


if __name__ == "__main__":
    main()
