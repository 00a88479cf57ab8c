"""dsymm_lower (Z = A V, A symmetric lower-stored, m x m) at several column counts: separates the A traffic (read once per call, from
both sides of the diagonal) from the products.  usage: time_symm.py [m]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from janusx_amd._lib import lib, check
m = int(sys.argv[1]) if len(sys.argv) > 1 else 17500
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
a = torch.randn((m, m), device=dev, dtype=torch.float64)
for n in (16, 32, 64, 128, 256):
    b = torch.randn((n, m), device=dev, dtype=torch.float64)
    c = torch.zeros((n, m), device=dev, dtype=torch.float64)
    fn = lambda: check(lib().jxg_dsymm_lower_f64(m, n, 1.0, a.data_ptr(), m, b.data_ptr(), m, 0.0, c.data_ptr(), m, st))
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"m={m} n={n}: {ms * 1e3:.0f} us  {2.0 * m * m * n / ms / 1e9:.1f} TFLOP/s  A read from both sides {8.0 * m * m / ms / 1e9:.2f} TB/s")
