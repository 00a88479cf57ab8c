"""Times the GRM accumulation (jxg_grm_accumulate) on a synthetic panel. GPU box only.
usage: time_grm.py n m [missing_rate]   (env JXGPU_GRM_EXACT=0 / JXGPU_GRM_TILE=256 select variants)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from janusx_amd import pipeline, stats as st
from janusx_amd._lib import lib


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
    miss = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    dev = torch.device("cuda:0")
    packed, _ = bench.synth_panel_gpu(n, m, 20260609, dev, missing_rate=miss)
    p = pipeline.Panel(packed, n)
    keep, mean_g, scale, flip, var = st.stream_grm_row_prepare(p.counts(), n, 1, 0.02, 0.05, 0.0)
    rows = np.nonzero(keep)[0]
    lut = st.grm_lut_from_mean_scale(mean_g[rows], scale[rows], flip[rows])
    acc = torch.zeros((p.npad, p.npad), dtype=torch.float64, device=dev)
    ms = []
    for rep in range(5):
        acc.zero_()
        pipeline.grm_accumulate(p, rows, lut, acc=acc)
        torch.cuda.synchronize()
        ms.append(float(lib().jxg_last_kernel_ms(0)))
    best = min(ms[1:])
    print(f"n={n} m_kept={len(rows)} miss={miss} exact={os.environ.get('JXGPU_GRM_EXACT','1')} "
          f"tile={os.environ.get('JXGPU_GRM_TILE','128')}: {best:.3f} ms  "
          f"{n * (n + 1.0) * len(rows) / best / 1e9:.0f} TFLOP/s algorithmic  (all: {[round(x, 3) for x in ms]})", flush=True)


main()
