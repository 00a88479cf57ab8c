"""GPU check of the position-owned bulge-chasing kernel against the sweep-owned one (bit-identical d, e) + timing:
python scripts/check_chase_owned.py [n ...]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from janusx_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda:0")


def make(n, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    z = torch.randn((n, 2 * n if n <= 6000 else n + 64), generator=g, device=dev, dtype=torch.float64)
    a = z @ z.T / z.shape[1]
    a.diagonal().add_(1e-6)
    return a


def run(a, owned):
    os.environ["JXGPU_BC_OWNED"] = os.environ.get("OWNED_MODE", "1") if owned else "0"
    n = a.shape[0]
    w = a.clone()
    d = torch.zeros(n, device=dev, dtype=torch.float64)
    e = torch.zeros(n, device=dev, dtype=torch.float64)
    hf = (ctypes.c_int * 4)()
    torch.cuda.synchronize()
    check(lib().jxg_sy2st_f64(w.data_ptr(), n, d.data_ptr(), e.data_ptr(), None, hf, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return d.cpu().numpy(), e.cpu().numpy(), list(hf)[:2]


ok = True
sizes = [int(x) for x in sys.argv[1:]] or [131, 194, 259, 300, 1000, 2049, 5000]
for n in sizes:
    a = make(n, n)
    d0, e0, f0 = run(a, False)
    d1, e1, f1 = run(a, True)
    same = bool(np.array_equal(d0, d1) and np.array_equal(e0, e1))
    dd = float(np.abs(d0 - d1).max()), float(np.abs(e0 - e1).max())
    print(f"n={n}: flags {f0} {f1} identical {same} max diff d {dd[0]:.2e} e {dd[1]:.2e}", flush=True)
    ok = ok and same and f1 == [0, 0]
    if n >= 5000:
        os.environ["JXGPU_EIGH_TRACE"] = "1"
        for owned in (False, True):
            print("owned" if owned else "sweep-owned", flush=True)
            run(a, owned)
        del os.environ["JXGPU_EIGH_TRACE"]
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
