"""Where the beta disagreement of the C1 (mouse) leg comes from: device scan vs the oracle with the f32 SGEMM rotation of the
reference and vs the oracle with an exact (f64) rotation, Brent evaluation counts beside it."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from janusx_amd import pipeline, stats
from oracle import jx_oracle as O
from oracle import jx_oracle_c as OC

d = np.load("tests/golden/mouse_hs1940.npz")
packed, n = np.ascontiguousarray(d["packed"]), len(d["ids"])
ph = d["pheno"][:, 0]
pos = {s: i for i, s in enumerate(d["pheno_ids"])}
yfull = np.array([ph[pos[s]] if s in pos else np.nan for s in d["ids"]])
keep_idx = np.nonzero(np.isfinite(yfull))[0]
y = yfull[keep_idx]
k_ref, eff_ref, _ = O.grm_stream_bed(packed, n, 1, 0.02, 0.05, 0.0)
x = np.ones((len(keep_idx), 1))
s, u = O.gwas_eigh_from_grm(k_ref, 1e-6, keep_idx)
nm = O.spectral_null_model(y, x, s, u)
mi, he, ho = O.row_counts(packed, n, keep_idx)
keep, maf, miss, flip = O.gwas_scan_row_stats(mi, he, ho, len(keep_idx), 0.02, 0.05, 1.0)
rows = np.nonzero(keep)[0]
gd = O.decode_centered_block_f32(packed, n, flip, maf, sample_idx=keep_idx, rows=rows)
g32 = O.rotate_block_f32(gd, nm.Dh)
g64 = (gd.astype(np.float64) @ nm.Dh.astype(np.float64).T).astype(np.float32)
lo, hi = nm.bounds
ref32, ev32 = O.lmm_scan_rotated_block(g32[:1500], nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2, count_evals=True)
ref64, ev64 = O.lmm_scan_rotated_block(g64[:1500], nm.S, nm.Xcov, nm.y, lo, hi, 30, 1e-2, count_evals=True)
pt = torch.from_numpy(packed).cuda()
kt = torch.from_numpy(np.ascontiguousarray(k_ref)).cuda()
sg, ut = pipeline.eigh_from_grm(kt, 1e-6, keep_idx)
model = pipeline.SpectralModel(sg, ut, x, y)
panel = pipeline.Panel(pt, n, keep_idx)
counts = panel.counts()
lut = stats.scan_lut_from_counts(maf[rows], np.zeros(len(rows), bool), counts[rows], len(keep_idx))
out, evg = pipeline.scan_rows(panel, model, rows[:1500], lut[:1500], "lmm", max_iter=30, tol=1e-2, return_evals=True)
out, evg = out.cpu().numpy(), evg.cpu().numpy()
grot_gpu = pipeline.rotate_rows(panel, model, rows[:1500], lut[:1500]).cpu().numpy()
sgn = np.sign(np.sum(model.ut.cpu().numpy() * nm.Dh, axis=1))       # eigenvector signs of the two decompositions
rng = np.abs(g64[:1500]).max(axis=1, keepdims=True)
print("rotation: device vs exact  max |d| / row range", float(np.max(np.abs(grot_gpu * sgn[None, :] - g64[:1500]) / rng)),
      " f32 SGEMM vs exact", float(np.max(np.abs(g32[:1500] - g64[:1500]) / rng)))


def cmp(a, b, name, ea, eb):
    ok = ~np.isnan(b[:, 0])
    be = np.abs(a[ok, 0] - b[ok, 0]) / np.maximum(np.abs(b[ok, 0]), b[ok, 1])
    se = np.abs(a[ok, 1] - b[ok, 1]) / b[ok, 1]
    same = (np.asarray(ea)[ok] == np.asarray(eb)[ok])
    print(f"{name}: beta max {be.max():.2e} (same Brent count: {be[same].max():.2e}, different: {be[~same].max() if (~same).any() else 0:.2e}; "
          f"{int((~same).sum())} of {int(ok.sum())} differ)  median {np.median(be):.2e}  se max {se.max():.2e}")


cmp(out, ref32, "device vs oracle f32-rotation", evg, ev32 + 0)
cmp(out, ref64, "device vs oracle exact rotation", evg, ev64 + 0)
cmp(ref32, ref64, "oracle f32-rotation vs oracle exact rotation", ev32, ev64)
print("eval count convention check (device - oracle):", np.unique(evg.astype(int) - ev64.astype(int), return_counts=True))
