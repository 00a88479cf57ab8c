"""How many design rows of the 1 %-missing C3 panel are NOT on the int8 rotation, and why (LUT of such a row)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from janusx_amd import pipeline as pl, stats as st
from janusx_amd._lib import lib, check
n, m = 20000, 65536
dev = torch.device("cuda", 0)
packed, _ = bench.synth_panel_gpu(n, m, 20260609, dev, m_offset=0, missing_rate=0.01)
panel = pl.Panel(packed, n, None)
counts = panel.counts()
keep, af, miss = st.gwas_scan_row_stats(counts, n, 0.02, 0.05, 1.0)
rows = np.nonzero(keep)[0].astype(np.int32)
lut = st.scan_lut_from_counts(af[rows], np.zeros(len(rows), bool), counts[rows], n)
rows_t = torch.from_numpy(rows).to(dev); lut_t = torch.from_numpy(lut).to(dev)
mk = len(rows)
lut16 = torch.empty((mk, 16), dtype=torch.uint8, device=dev); rowoff = torch.empty(mk, dtype=torch.float32, device=dev)
mm = int(lib().jxg_rot_miss_max(n, panel.mean_missing()))
rowmiss = torch.zeros(mk, dtype=torch.float32, device=dev)
check(lib().jxg_lut_split_rows_m(panel.p32.data_ptr(), panel.m, n, rows_t.data_ptr(), lut_t.data_ptr(), mk, lut16.data_ptr(),
                                 rowoff.data_ptr(), rowmiss.data_ptr(), mm, 0))
ro = rowoff.cpu().numpy()
bad = np.nonzero(np.isnan(ro))[0]
print("rows", mk, "miss_max", mm, "inexact", len(bad))
for i in bad[:6]:
    print(i, lut[i], counts[rows[i]], af[rows[i]])
