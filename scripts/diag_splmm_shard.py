"""Diagnostic: is the SparseLMM exact scan of a row independent of the other rows in the call (and repeatable)?"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from janusx_amd import bed, janusx as jxrs
import test_gpu_parity as T

n, m = 900, 2400
packed, g = T._related_panel(n, m, 47, 0.01)
y = bed.synth_phenotype(g, n_causal=12, pve=0.5, seed=7)
tmp = tempfile.mkdtemp()
prefix = os.path.join(tmp, "p")
ids = [f"s{i}" for i in range(n)]
bim = bed.Bim(["1"] * m, [f"rs{j}" for j in range(m)], list(range(1, m + 1)), ["A"] * m, ["G"] * m)
bed.write_bed(prefix, packed, ids, bim)
path, _, nnz = jxrs.spgrm_bed_to_jxgrm(prefix, out_prefix=prefix, method=1, threshold=0.05, maf_threshold=0.02, max_missing_rate=0.05)
keep_idx = np.array([i for i in range(n) if i % 7 != 0], dtype=np.int64)
rng = np.random.default_rng(47)
yy = np.array([float(repr(float(y[i] + rng.normal()))) for i in keep_idx])
counts = jxrs.bed_row_counts(packed, n, keep_idx)
from janusx_amd import stats as st
keep, af, miss = st.gwas_scan_row_stats(counts, len(keep_idx), 0.02, 0.05, 1.0)
kept = np.nonzero(keep)[0]
maf_all = np.zeros(m, dtype=np.float32); maf_all[kept] = af[kept]
def scan(rows):
    s, l10, null = jxrs.splmm_exact_scan_from_jxgrm(path, yy, packed, n, maf_all, np.zeros(m, bool), None, keep_idx, rows, grid_size=17, tol=1e-3, max_iter=20)
    return s, l10
a, la = scan(kept)
b, lb = scan(kept)
h = len(kept) // 2
c, lc = scan(kept[:h])
d, ld = scan(kept[h:])
print("l10", la, lb, lc, ld)
print("repeat differs rows:", np.nonzero((a != b).any(1))[0][:10])
cd = np.concatenate([c, d])
bad = np.nonzero((a != cd).any(1) & ~np.isnan(a).any(1))[0]
print("half differs rows:", bad[:20], len(bad))
for i in bad[:5]:
    print(i, a[i], cd[i], "miss", miss[kept[i]])
# repeatability of the eigendecomposition behind the model (fresh each time)
outs = []
for rep in range(4):
    jxrs.spectral_cache_clear()
    mdl = jxrs._SpectralSparseReml(path, yy, None, keep_idx)
    outs.append((mdl.s_dev.cpu().numpy().copy(), mdl.ut_dev.cpu().numpy().copy()))
for rep in range(1, 4):
    print("eigh rep", rep, "s differs", int((outs[0][0] != outs[rep][0]).sum()), "ut differs", int((outs[0][1] != outs[rep][1]).sum()),
          "max", float(np.abs(outs[0][1] - outs[rep][1]).max()))
jxrs.spectral_cache_clear()
e, le = scan(kept)
print("fresh-model scan differs rows:", int((a != e).any(1).sum()), la, le)
