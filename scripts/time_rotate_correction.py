"""Where the time of a scan with a few missing calls per row goes (GPU box): usamp transpose, row classification, scan_rows.
usage: time_rotate_correction.py [n] [m] [missing_rate]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from janusx_amd import pipeline as pl, stats as st  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
rate = float(sys.argv[3]) if len(sys.argv) > 3 else 0.002
dev = torch.device("cuda", 0)
packed, dos = bench.synth_panel_gpu(n, m, 11, dev, missing_rate=rate)
y = bench.make_phenotype(dos, n, 7, dev)
k32, _eff, panel = pl.build_grm(packed, n)
s, ut = pl.eigh_from_grm(k32)
del k32
counts = panel.counts()
keep, af, _miss = st.gwas_scan_row_stats(counts, n, 0.02, 0.05, 1.0)
rows = np.nonzero(keep)[0]
lut = st.scan_lut_from_counts(af[rows], np.zeros(len(rows), dtype=bool), counts[rows], n)
x = np.ones((n, 1))
for mm in ("default", "0", "default", "0"):
    if mm == "0":
        os.environ["JXGPU_ROT_MISS_MAX"] = "0"
    else:
        os.environ.pop("JXGPU_ROT_MISS_MAX", None)
    model = pl.SpectralModel(s, ut, x, y)
    lo, hi = model.null.bounds
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mm != "0":
        model.usamp()
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    tm = pl.StageTimes()
    out = pl.scan_rows(panel, model, rows, lut, mode="lmm", low=lo, high=hi, max_iter=30, tol=1e-2, times=tm)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    d = tm.as_dict() if hasattr(tm, "as_dict") else getattr(tm, "t", {})
    print(f"rot_miss_max={mm}: usamp {1e3 * (t1 - t0):.1f} ms, scan_rows {1e3 * (t2 - t1):.1f} ms, stages {d}", flush=True)
