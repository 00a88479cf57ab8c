"""Exact per-SNP scan with covariates: time of the rotation + association stages for several covariate counts (the
BASELINE shapes are intercept-only; a GWAS with principal components as covariates is not).  GPU box only.
usage: time_scan_covariates.py [n] [m] [q1 q2 ...]   (q = covariates beside the intercept; default 0 2 5 10)
JX_SCAN_MODE=lmm|lmm2|fvlmm selects the scan (default lmm)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from janusx_amd import pipeline as pl, stats as st  # noqa: E402
from janusx_amd._lib import lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    qs = [int(a) for a in sys.argv[3:]] or [0, 2, 5, 10]
    dev = torch.device("cuda", 0)
    packed, dos = bench.synth_panel_gpu(n, m, 11, dev)
    y = bench.make_phenotype(dos, n, 7, dev)
    k32, _eff, panel = pl.build_grm(packed, n)
    s, ut = pl.eigh_from_grm(k32)
    del k32
    counts = panel.counts()
    keep, af, _miss = st.gwas_scan_row_stats(counts, n, 0.02, 0.05, 1.0)
    rows = np.nonzero(keep)[0]
    lut = st.scan_lut_from_counts(af[rows], np.zeros(len(rows), dtype=bool), counts[rows], n)
    rng = np.random.default_rng(5)
    for q in qs:
        x = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, q))], axis=1)
        model = pl.SpectralModel(s, ut, x, y)
        lo, hi = model.null.bounds
        for rep in range(2):
            tm = pl.StageTimes()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mode = os.environ.get("JX_SCAN_MODE", "lmm")
            kw = {"nullml": model.null.ml0} if mode == "lmm2" else {}
            out = pl.scan_rows(panel, model, rows, lut, mode=mode, low=lo, high=hi, max_iter=30, tol=1e-2, times=tm, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        form = {0.0: "lds-resident", 1.0: "tiled", 2.0: "one wave per SNP (global)"}.get(float(lib().jxg_last_kernel_ms(11)), "?")
        d = tm.as_dict() if hasattr(tm, "as_dict") else getattr(tm, "t", {})
        print(f"n={n} m_kept={len(rows)} covariates={q} (dim {q + 2}): scan {dt * 1e3:.1f} ms  stages {d}  form {form}  "
              f"finite rows {int(torch.isfinite(out[:, 0]).sum())}", flush=True)


main()
