"""Stage-by-stage run of the pipeline at a given n (small m) with a sync + message after every stage: locates the
stage that fails at a new problem size.  usage: stage_probe.py n m"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from janusx_amd import pipeline as pl, stats as st
from janusx_amd._lib import lib

def say(msg):
    torch.cuda.synchronize()
    print(f"[{time.perf_counter() - T0:7.1f}s] {msg}  (alloc {torch.cuda.memory_allocated() / 2**30:.1f} GiB)", flush=True)

n, m = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
T0 = time.perf_counter()
packed, dos = bench.synth_panel_gpu(n, m, 1, dev)
y = bench.make_phenotype(dos, n, 1, dev)
say("panel generated")
panel = pl.Panel(packed, n); counts = panel.counts(); say("repack + counts")
gkeep, mean_g, scale, flip, var = st.stream_grm_row_prepare(counts, n, 1, 0.02, 0.05, 0.0)
grows = np.nonzero(gkeep)[0]
glut = st.grm_lut_from_mean_scale(mean_g[grows], scale[grows], flip[grows])
acc = pl.grm_accumulate(panel, grows, glut); say("grm accumulate")
k32 = pl.grm_finalize(acc, n, float(np.sum(var[grows])), torch.float32); del acc; say("grm finalize")
s, ut64 = pl.eigh_from_grm(k32, 1e-6); say("eigh")
model = pl.SpectralModel(s, ut64, np.ones((n, 1)), y); del ut64; say(f"null model lbd={model.null.lbd:.4g}")
keep, af, miss = st.gwas_scan_row_stats(counts, n, 0.02, 0.05, 1.0)
rows = np.nonzero(keep)[0]
lut = st.scan_lut_from_counts(af[rows], np.zeros(len(rows), dtype=bool), counts[rows], n)
out = pl.scan_rows(panel, model, rows, lut, "lmm", max_iter=30, tol=1e-2); say("lmm scan")
out2 = pl.scan_rows(panel, model, rows, lut, "fvlmm"); say("fvlmm scan")
print("finite rows", int(torch.isfinite(out[:, 0]).sum()), "of", len(rows))
