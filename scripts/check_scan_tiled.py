"""GPU check: the tiled exact scan (n beyond the LDS-resident limit) against the one-wave-per-SNP form, bit for bit,
and its timing.  python scripts/check_scan_tiled.py [n] [m]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from janusx_amd import pipeline, stats
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
dev = torch.device("cuda:0")
packed, dos = bench.synth_panel_gpu(n, m, 20260609, dev, missing_rate=0.01)
y = bench.make_phenotype(dos, n, 20260609, dev)
k, eff, panel = pipeline.build_grm(packed, n, 1, 0.02, 0.05)
s, ut = pipeline.eigh_from_grm(k, 1e-6)
model = pipeline.SpectralModel(s, ut, np.ones((n, 1)), y)
del ut
counts = panel.counts()
keep, af, miss = stats.gwas_scan_row_stats(counts, n, 0.02, 0.05, 1.0)
rows = np.nonzero(keep)[0]
lut = stats.scan_lut_from_counts(af[rows], np.zeros(len(rows), bool), counts[rows], n)
res = {}
for tag, env in (("tiled", None), ("plain", "1")):
    if env: os.environ["JXGPU_SCAN_NOTILE"] = env
    else: os.environ.pop("JXGPU_SCAN_NOTILE", None)
    for rep in range(2):
        tm = pipeline.StageTimes()
        out, ev = pipeline.scan_rows(panel, model, rows, lut, "lmm", times=tm, return_evals=True)
    res[tag] = (out.cpu().numpy(), ev.cpu().numpy(), tm.t)
    print(tag, {k2: round(v * 1e3, 1) for k2, v in tm.t.items()}, "mean evals", float(ev.float().mean()))
a, b = res["tiled"], res["plain"]
print("identical stats:", np.array_equal(a[0], b[0], equal_nan=True), "identical evals:", np.array_equal(a[1], b[1]))
ok = ~np.isnan(b[0][:, 0])
rel = np.abs(a[0][ok, :2] - b[0][ok, :2]) / np.maximum(np.abs(b[0][ok, :2]), b[0][ok, 1:2])
print("NaN pattern equal:", np.array_equal(np.isnan(a[0]), np.isnan(b[0])), "max rel diff beta/se", float(rel.max()),
      "evals differing:", int((a[1] != b[1]).sum()), "of", len(a[1]))
