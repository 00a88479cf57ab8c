"""Times jxg_repack_p32 (identity order and a sorted 80 % sample subset) and jxg_p32_transpose.  usage: time_repack.py [n_src] [m]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from janusx_amd._lib import lib, check
n_src = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
dev = torch.device("cuda", 0)
packed, _ = bench.synth_panel_gpu(n_src, m, 3, dev, m_offset=0, missing_rate=0.01)
bps = packed.shape[1]
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, idx in (("identity", None), ("subset80", np.sort(np.random.default_rng(1).choice(n_src, int(0.8 * n_src), replace=False)).astype(np.int32))):
    n = n_src if idx is None else len(idx)
    nt = lib().jxg_num_tiles(n)
    p32 = torch.empty((nt, m, 32), dtype=torch.uint8, device=dev)
    it = None if idx is None else torch.from_numpy(idx).to(dev)
    ms = timeit(lambda: check(lib().jxg_repack_p32(packed.data_ptr(), bps, n_src, m, None if it is None else it.data_ptr(), n, None, m, p32.data_ptr(), st)))
    gb = (m * bps + p32.numel()) / 1e9
    print(f"repack {name}: n={n} m={m} {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s (in + out {gb:.2f} GB)  digest {int(p32.view(torch.int64).sum().item()) & 0xffffffff:08x}")
    t32 = torch.empty(int(lib().jxg_t32_bytes(n, m)), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: check(lib().jxg_p32_transpose(p32.data_ptr(), m, n, None, m, t32.data_ptr(), st)))
    print(f"transpose {name}: {ms:.3f} ms  {2 * p32.numel() / 1e9 / ms * 1e3:.0f} GB/s  digest {int(t32.view(torch.int64).sum().item()) & 0xffffffff:08x}")
