"""Times / validates the device eigendecomposition (jxg_eigh_f64) for a few n. GPU box only.
usage: time_eigh.py n1 n2 ...   (env JXGPU_EIGH=rocsolver selects the library path)"""
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
        k = 0.5 * (k + k.T)
        del z
        for rep in range(2):
            a = k.clone()
            w = torch.empty(n, dtype=torch.float64, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            check(lib().jxg_eigh_f64(a.data_ptr(), n, 1e-6, w.data_ptr(), torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        kk = k + 1e-6 * torch.eye(n, device=dev, dtype=torch.float64)
        v = a  # rows = eigenvectors
        res = float((v @ kk - w[:, None] * v).abs().max())
        orth = float((v @ v.T - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
        wref = torch.linalg.eigvalsh(kk) if n <= 6000 else None
        werr = float((w - wref).abs().max()) if wref is not None else float("nan")
        asc = bool((w[1:] >= w[:-1]).all())
        st = {nm: round(float(lib().jxg_last_kernel_ms(i)), 1) for i, nm in ((6, "band"), (7, "chase"), (8, "dc"), (9, "q1"), (4, "q2_kernel"))}
        print(f"n={n} mode={os.environ.get('JXGPU_EIGH','custom')} eigh {dt*1e3:.1f} ms  resid {res:.2e} orth {orth:.2e} "
              f"eval_err {werr:.2e} ascending {asc} stages_ms {st}", flush=True)
        del a, k, kk, v

main()
