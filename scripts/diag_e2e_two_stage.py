#!/usr/bin/env python3
"""Where the end-to-end disagreement at n = 5000 comes from (tests/test_gpu_round6.py::test_end_to_end_two_stage): per-SNP error
distribution of beta against the oracle that does its own GRM / eigh / rotation, split by whether the GPU's Brent search took the
oracle's number of objective evaluations (same trajectory) or not (a branch of the search decided differently)."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from janusx_amd import bed, pipeline, stats as st  # noqa: E402
from oracle import jx_oracle as O, jx_oracle_c as OC  # noqa: E402

OC.build()
n, m = 5000, 20000
MISSING = float(os.environ.get("DIAG_MISSING", "0.002"))
packed, g = bed.synth_panel_numpy(n, m, seed=61, missing_rate=MISSING)
y = bed.synth_phenotype(g, n_causal=40, pve=0.5, seed=61)
del g
mi, he, ho = O.row_counts(packed, n)
k_ref, eff, _ = O.grm_stream_bed(packed, n, 1, 0.02, 0.05, 0.0)
s, u = O.gwas_eigh_from_grm(k_ref)
nm = O.spectral_null_model(y, np.ones((n, 1)), s, u)
keep, maf, miss, flip = O.gwas_scan_row_stats(mi, he, ho, n, 0.02, 0.05, 1.0)
rows = np.nonzero(keep)[0]
gd = O.decode_centered_block_f32(packed, n, flip, maf, rows=rows)
grot = O.rotate_block_f32(gd, nm.Dh)
ref, ev_ref = OC.lmm_scan_rotated_block(grot, nm.S, nm.Xcov, nm.y, nm.bounds[0], nm.bounds[1], 30, 1e-2, threads=os.cpu_count(),
                                        return_evals=True)
# the same oracle with the GRM's block products in exact arithmetic (no f32 SSYRK rounding)
k_x, _, _ = O.grm_stream_bed(packed, n, 1, 0.02, 0.05, 0.0, exact_syrk=True)
s_x, u_x = O.gwas_eigh_from_grm(k_x)
nm_x = O.spectral_null_model(y, np.ones((n, 1)), s_x, u_x)
ref_x = OC.lmm_scan_rotated_block(O.rotate_block_f32(gd, nm_x.Dh), nm_x.S, nm_x.Xcov, nm_x.y, nm_x.bounds[0], nm_x.bounds[1], 30, 1e-2,
                                  threads=os.cpu_count())
dev = torch.device("cuda", 0)
pk = torch.from_numpy(packed).to(dev)
out = {}
okx = ~np.isnan(ref[:, 0])
bxx = np.abs(ref_x[okx, 0] - ref[okx, 0]) / np.maximum(np.abs(ref[okx, 0]), ref[okx, 1])
out["oracle_f32_syrk_vs_oracle_exact_syrk"] = {"beta_err_quantiles_50_90_99_999_max": [float(v) for v in np.quantile(bxx, [0.5, 0.9, 0.99, 0.999, 1.0])],
                                               "grm_max_rel": float(np.max(np.abs(k_x.astype(np.float64) - k_ref) / np.maximum(np.abs(k_ref), np.mean(np.diag(k_ref))))),
                                               "lbd": [nm.lbd_null, nm_x.lbd_null]}
for planes in (5,):
    k32, geff, panel = pipeline.build_grm(pk, n, 1, 0.02, 0.05)
    s_d, ut64 = pipeline.eigh_from_grm(k32, 1e-6, f32_consumer=(planes == 5))
    model = pipeline.SpectralModel(s_d, ut64, np.ones((n, 1)), y)
    counts = panel.counts()
    lut = st.scan_lut_from_counts(maf[rows], np.zeros(len(rows), bool), counts[rows], n)
    got, ev = pipeline.scan_rows(panel, model, rows, lut, "lmm", max_iter=30, tol=1e-2, return_evals=True)
    got, ev = got.cpu().numpy(), ev.cpu().numpy()
    ok = ~np.isnan(ref[:, 0])
    be = np.abs(got[ok, 0] - ref[ok, 0]) / np.maximum(np.abs(ref[ok, 0]), ref[ok, 1])
    se = np.abs(got[ok, 1] - ref[ok, 1]) / ref[ok, 1]
    same = ev[ok] == ev_ref[ok]
    # the same scan given the ORACLE's spectral inputs (isolates rotation + scan from GRM + eigh)
    m2 = pipeline.SpectralModel(torch.from_numpy(nm.S).to(dev), torch.from_numpy(nm.Dh.astype(np.float64)).to(dev), np.ones((n, 1)), y)
    got2, ev2 = pipeline.scan_rows(panel, m2, rows, lut, "lmm", low=nm.bounds[0], high=nm.bounds[1], max_iter=30, tol=1e-2,
                                   return_evals=True)
    got2, ev2 = got2.cpu().numpy(), ev2.cpu().numpy()
    be2 = np.abs(got2[ok, 0] - ref[ok, 0]) / np.maximum(np.abs(ref[ok, 0]), ref[ok, 1])
    same2 = ev2[ok] == ev_ref[ok]
    q = lambda v: [float(x) for x in np.quantile(v, [0.5, 0.9, 0.99, 0.999, 1.0])]   # noqa: E731
    bex = np.abs(got[ok, 0] - ref_x[ok, 0]) / np.maximum(np.abs(ref_x[ok, 0]), ref_x[ok, 1])
    k_gpu = k32.double().cpu().numpy()
    out[f"planes{planes}"] = {
        "vs_oracle_exact_syrk_beta_err_quantiles": q(bex),
        "grm_gpu_vs_f32_syrk_oracle": float(np.max(np.abs(k_gpu - k_ref) / np.maximum(np.abs(k_ref), np.mean(np.diag(k_ref))))),
        "grm_gpu_vs_exact_syrk_oracle": float(np.max(np.abs(k_gpu - k_x) / np.maximum(np.abs(k_x), np.mean(np.diag(k_x))))),
        "lbd_gpu": model.null.lbd, "lbd_oracle": nm.lbd_null, "bounds_gpu": model.null.bounds, "bounds_oracle": nm.bounds,
        "beta_err_quantiles_50_90_99_999_max": q(be), "se_err_quantiles": q(se),
        "share_same_evals": float(same.mean()), "beta_err_max_same_evals": float(be[same].max()),
        "beta_err_max_other": float(be[~same].max()) if (~same).any() else 0.0,
        "n_above_1e-5": int((be > 1e-5).sum()), "n_above_1e-5_same_evals": int((be[same] > 1e-5).sum()),
        "oracle_spectral_inputs": {"beta_err_quantiles": q(be2), "share_same_evals": float(same2.mean()),
                                   "beta_err_max_same_evals": float(be2[same2].max()),
                                   "n_above_1e-5": int((be2 > 1e-5).sum())}}
    del ut64, model, m2
tag = os.environ.get("DIAG_TAG", "default")
print(json.dumps(out, indent=1))
v = out["planes5"]
print(f"SUMMARY {tag}: missing={MISSING} beta e2e max {v['beta_err_quantiles_50_90_99_999_max'][-1]:.3e} (p99 {v['beta_err_quantiles_50_90_99_999_max'][2]:.3e}) "
      f"GRM gpu vs f32-syrk oracle {v['grm_gpu_vs_f32_syrk_oracle']:.3e} vs exact {v['grm_gpu_vs_exact_syrk_oracle']:.3e}; "
      f"oracle f32 vs exact syrk: beta {out['oracle_f32_syrk_vs_oracle_exact_syrk']['beta_err_quantiles_50_90_99_999_max'][-1]:.3e}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"diag_e2e_two_stage_{tag}.json"), "w"), indent=1)
