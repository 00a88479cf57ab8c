"""Times the device eigendecomposition without validation (experiments with JXGPU_SYTRD_* switches). GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from janusx_amd._lib import lib, check

for n in [int(a) for a in sys.argv[1:]] or [5000]:
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    z = torch.randn((n, n + 64), generator=g, device=dev, dtype=torch.float32)
    k = (z @ z.T / (n + 64)).to(torch.float64)
    del z
    best = 1e9
    for rep in range(3):
        a = k.clone()
        w = torch.empty(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = lib().jxg_eigh_f64(a.data_ptr(), n, 1e-6, w.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"n={n} dbg={os.environ.get('JXGPU_SYTRD_DBG','0')} kt={os.environ.get('JXGPU_SYTRD_KT','-')} rc={rc} eigh {best*1e3:.1f} ms", flush=True)
