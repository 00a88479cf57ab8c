"""Times the device eigendecomposition (jxg_eigh_f64) for a few n. GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from janusx_amd._lib import lib, check

def main():
    ns = [int(a) for a in sys.argv[1:]] or [2000, 5000, 10000]
    dev = torch.device("cuda:0")
    for n in ns:
        g = torch.Generator(device=dev); g.manual_seed(1)
        z = torch.randn((n, n + 64), generator=g, device=dev, dtype=torch.float32)
        k = (z @ z.T / (n + 64)).to(torch.float64)
        del z
        for rep in range(2):
            a = k.clone()
            w = torch.empty(n, dtype=torch.float64, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            check(lib().jxg_eigh_f64(a.data_ptr(), n, 1e-6, w.data_ptr(), torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        # residual check on a few vectors
        v = a[-3:]  # rows = eigenvectors
        kk = k + 1e-6 * torch.eye(n, device=dev, dtype=torch.float64)
        res = float((v @ kk - w[-3:, None] * v).abs().max())
        print(f"n={n} eigh {dt*1e3:.1f} ms  resid {res:.2e}", flush=True)
        del a, k, kk

main()
