"""Timing of the plain LM scan (`lm_dots_kernel` + `lm_stats_kernel`, csrc/k_lm.hip) on a resident synthetic panel (GPU box).
python scripts/time_lm_scan.py [n] [m] [covariates]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from janusx_amd import pipeline as pl          # noqa: E402
from janusx_amd import stats as st             # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    ncov = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dev = torch.device("cuda", 0)
    packed, dos = bench.synth_panel_gpu(n, m, 11, dev, missing_rate=0.01)
    y = bench.make_phenotype(dos, n, 7, dev)
    x = np.concatenate([np.ones((n, 1)), np.random.default_rng(1).standard_normal((n, ncov))], axis=1)
    panel = pl.Panel(packed, n)
    counts = panel.counts()
    keep, af, _miss = st.gwas_scan_row_stats(counts, n, 0.02, 0.05, 1.0)
    rows = np.nonzero(keep)[0]
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pl.scan_rows_lm(panel, rows, af[rows], x, y)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"n={n} rows={len(rows)} q0={x.shape[1]}: {dt * 1e3:.1f} ms ({len(rows) / dt / 1e6:.2f} M SNPs/s, packed stream "
              f"{len(rows) * ((n + 127) // 128) * 32 / dt / 1e9:.0f} GB/s); min p {float(out[:, 2].min()):.2e}")


if __name__ == "__main__":
    main()
