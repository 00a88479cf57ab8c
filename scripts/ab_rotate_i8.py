"""A/B of the int8 rotation kernels on one block of BASELINE configs[2] rows: prints the time per launch and a digest of the
rotated block (the two forms of k_rotate_i8.hip must give the same bits).  Run once per value of JXGPU_ROT_I8_DMA (the switch is
read once per process).  usage: ab_rotate_i8.py [n] [rows] [missing_rate]"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from janusx_amd import pipeline as pl, stats as st
from janusx_amd._lib import lib, check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
miss = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dev = torch.device("cuda", 0)
packed, _ = bench.synth_panel_gpu(n, m, 20260609, dev, m_offset=0, missing_rate=miss)
panel = pl.Panel(packed, n, None)
counts = panel.counts()
keep, af, ms = st.gwas_scan_row_stats(counts, n, 0.02, 0.05, 1.0)
rows = np.nonzero(keep)[0].astype(np.int32)
mk = len(rows)
lut = st.scan_lut_from_counts(af[rows], np.zeros(mk, bool), counts[rows], n)
rows_t = torch.from_numpy(rows).to(dev)
lut_t = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float32)).to(dev)
g = torch.Generator(device=dev); g.manual_seed(7)
ut = torch.randn((n, n), generator=g, device=dev, dtype=torch.float32) / np.sqrt(n)
npad = lib().jxg_num_tiles(n) * 128
q = torch.empty((3, npad, npad), dtype=torch.int8, device=dev); umax = torch.empty(npad, dtype=torch.float32, device=dev)
check(lib().jxg_ut_quant3(ut.data_ptr(), n, q.data_ptr(), umax.data_ptr(), None))
usum = torch.empty(npad, dtype=torch.float32, device=dev)
check(lib().jxg_ut_rowsum(ut.data_ptr(), n, usum.data_ptr(), None))
lut16 = torch.empty((mk, 16), dtype=torch.uint8, device=dev); rowoff = torch.empty(mk, dtype=torch.float32, device=dev)
rowmiss = torch.zeros(mk, dtype=torch.float32, device=dev)
check(lib().jxg_lut_split_rows_m(panel.p32.data_ptr(), panel.m, n, rows_t.data_ptr(), lut_t.data_ptr(), mk, lut16.data_ptr(),
                                 rowoff.data_ptr(), rowmiss.data_ptr(), 1 << 30 if miss > 0 else 0, None))
ex = np.flatnonzero(~np.isnan(rowoff.cpu().numpy())).astype(np.int32)
sel = torch.from_numpy(ex).to(dev)
out = torch.zeros((mk, n), dtype=torch.float32, device=dev)
selm = torch.from_numpy(np.flatnonzero(rowmiss.cpu().numpy() != 0).astype(np.int32)).to(dev)

def run():
    check(lib().jxg_rotate_packed16x_q(panel.p32.data_ptr(), panel.m, n, rows_t.data_ptr(), mk, lut16.data_ptr(), rowoff.data_ptr(),
                                       usum.data_ptr(), None, None, 10, q.data_ptr(), umax.data_ptr(), sel.data_ptr(), len(ex),
                                       None, 0, out.data_ptr(), None))
    if len(selm):
        check(lib().jxg_rotate_missing_dense(panel.p32.data_ptr(), panel.m, n, rows_t.data_ptr(), selm.data_ptr(), len(selm),
                                             rowmiss.data_ptr(), q.data_ptr(), umax.data_ptr(), out.data_ptr(), n, None))
run(); torch.cuda.synchronize()
dig = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
ops = 2.0 * len(ex) * n * n * 3 + 2.0 * len(selm) * n * n * 3
print(f"JXGPU_ROT_I8_DMA={os.environ.get('JXGPU_ROT_I8_DMA', '(default)')} n={n} rows={mk} exact={len(ex)} with_missing_term={len(selm)} "
      f"ms={min(ts):.2f} (median {sorted(ts)[2]:.2f}) int8 POP/s={ops / min(ts) / 1e9:.1f} digest={dig} "
      f"|out|max={float(out.abs().max()):.4f}")
