"""dsyr2k (lower tiles) at several K for one size: separates the per-tile cost (C read-modify-write) from the K loop.
usage: time_syr2k_k.py [m]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from janusx_amd._lib import lib, check
m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for K in (128, 256, 512, 1024):
    a = torch.randn((K, m), device=dev, dtype=torch.float64)
    b = torch.randn((K, m), device=dev, dtype=torch.float64)
    c = torch.randn((m, m), device=dev, dtype=torch.float64)
    fn = lambda: check(lib().jxg_dsyr2k_lower_nt_f64(m, K, -1.0, a.data_ptr(), m, b.data_ptr(), m, 1.0, c.data_ptr(), m, st))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"m={m} K={K}: {ms * 1e3:.0f} us  {m * (m + 1) * K / ms / 1e9:.1f} TFLOP/s")
