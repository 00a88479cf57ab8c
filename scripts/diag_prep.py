"""Where a slow `prep` of bench.py's step goes (seen in about one process out of three: 10 - 29 ms per step instead of 1.5):
re-tiling kernel, count kernel, the device -> host copy of the counts, each bracketed by synchronisations, over 12 repeats.
usage: diag_prep.py [n] [m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from janusx_amd import pipeline as pl
from janusx_amd._lib import lib, check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
dev = torch.device("cuda", 0)
packed, _ = bench.synth_panel_gpu(n, m, 20260609, dev, m_offset=0, missing_rate=0.0)
big = torch.empty((n, n), dtype=torch.float64, device=dev)          # some HBM in use, as in a step
rows = []
for rep in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    panel = pl.Panel(packed, n)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    c = torch.empty((panel.m, 3), dtype=torch.int32, device=dev)
    check(lib().jxg_row_counts_p32(panel.p32.data_ptr(), panel.m, panel.n, c.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    h = c.cpu()
    t3 = time.perf_counter()
    a = h.numpy().copy()
    t4 = time.perf_counter()
    rows.append([round((b - a_) * 1e3, 3) for a_, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4))])
    del panel
print("ms per repeat [re-tile incl. allocation, count kernel, D2H copy, numpy copy]:")
for r in rows: print("  ", r)
