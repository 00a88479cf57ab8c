"""GRM kernel timing for a synthetic panel: python scripts/time_grm.py n m  (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from janusx_amd import pipeline as pl, stats as st
from janusx_amd._lib import lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n, m = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
packed, _ = bench.synth_panel_gpu(n, m, 1, dev)
panel = pl.Panel(packed, n)
counts = panel.counts()
keep, mean_g, scale, flip, var = st.stream_grm_row_prepare(counts, n, 1, 0.02, 0.05, 0.0)
rows = np.nonzero(keep)[0]
lut = st.grm_lut_from_mean_scale(mean_g[rows], scale[rows], flip[rows])
for rep in range(3):
    acc = pl.grm_accumulate(panel, rows, lut)
    ms = lib().jxg_last_kernel_ms(0)
print(f"n={n} m={len(rows)} tile={os.environ.get('JXGPU_GRM_TILE','auto')} grm {ms:.3f} ms  {n*(n+1.0)*len(rows)/ms/1e9:.1f} TFLOP/s algorithmic")
