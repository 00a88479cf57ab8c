"""Times the rank-2k update of the band reduction alone (jxg_dsyr2k_lower_nt_f64, K = 128) at a few trailing sizes.
usage: time_syr2k.py [sizes ...]   (JXGPU_SYR2K_PIPE / JXGPU_SYR2K_PIPE_WGS select the form)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from janusx_amd._lib import lib, check
sizes = [int(a) for a in sys.argv[1:]] or [19872, 15000, 10000, 6000]
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for m in sizes:
    g = torch.Generator(device=dev); g.manual_seed(1)
    a = torch.randn((128, m), device=dev, dtype=torch.float64, generator=g)      # column-major (m, 128): stored as its transpose
    b = torch.randn((128, m), device=dev, dtype=torch.float64, generator=g)
    c = torch.randn((m, m), device=dev, dtype=torch.float64, generator=g)
    c0 = c.clone()
    fn = lambda: check(lib().jxg_dsyr2k_lower_nt_f64(m, 128, -1.0, a.data_ptr(), m, b.data_ptr(), m, 1.0, c.data_ptr(), m, st))
    fn(); torch.cuda.synchronize()
    ref = c0.T - (a.T @ b)          # column-major C = C - A B'  <->  row-major view: C^T
    err = float((torch.tril(c.T) - torch.tril(ref)).abs().max())
    dig = int(torch.tril(c.T).view(torch.int64).sum().item()) & 0xffffffff
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"m={m} K=128: {ms * 1e3:.0f} us  {m * (m + 1) * 128 / ms / 1e9:.1f} TFLOP/s  C traffic {m * (m + 1) / 2 * 16 / ms / 1e9:.2f} TB/s  err {err:.2e} digest {dig:08x}")
