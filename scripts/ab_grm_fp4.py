"""A/B of the count Gram on the fp4 pipes (JXGPU_GRM_FP4=1, k_grm_fp4.hip) against the int8 kernel: time per accumulate and a digest of
the lower triangle of the accumulator (both are exact integer sums + the same f64 merge: the same bits).  usage: ab_grm_fp4.py n m [miss]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from janusx_amd import pipeline, stats as st
from janusx_amd._lib import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
m = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
miss = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dev = torch.device("cuda:0")
packed, _ = bench.synth_panel_gpu(n, m, 20260609, dev, missing_rate=miss)
p = pipeline.Panel(packed, n)
keep, mean_g, scale, flip, var = st.stream_grm_row_prepare(p.counts(), n, 1, 0.02, 0.05, 0.0)
rows = np.nonzero(keep)[0]
lut = st.grm_lut_from_mean_scale(mean_g[rows], scale[rows], flip[rows])
acc = torch.zeros((p.npad, p.npad), dtype=torch.float64, device=dev)
out = {}
for form in ("0", "0", "1", "1"):
    os.environ["JXGPU_GRM_FP4"] = form
    ms = []
    for rep in range(4):
        acc.zero_()
        pipeline.grm_accumulate(p, rows, lut, acc=acc)
        torch.cuda.synchronize()
        ms.append(float(lib().jxg_last_kernel_ms(0)))
    tri = torch.tril(acc[:n, :n]).cpu().numpy()
    out[form] = tri
    print(f"JXGPU_GRM_FP4={form} n={n} m_kept={len(rows)} miss={miss}: {min(ms[1:]):.3f} ms  {n * (n + 1.0) * len(rows) / min(ms[1:]) / 1e9:.0f} TOP/s algorithmic "
          f"digest {hashlib.sha256(tri.tobytes()).hexdigest()[:16]}", flush=True)
d = np.abs(out["0"] - out["1"])
print("max |difference| of the lower triangle:", float(d.max()), "equal bits:", bool(np.array_equal(out["0"], out["1"])))
