"""GPU check of jxg_eigh_f64 (two-stage path): invariants at several sizes + stage timing.
python scripts/check_eigh2.py [sizes...]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from janusx_amd import pipeline  # noqa: E402

dev = torch.device("cuda:0")


def grm_like(n, seed, rank_frac):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    m = max(1, int(n * rank_frac))
    z = torch.randn((n, m), generator=g, device=dev, dtype=torch.float32)
    k = z @ z.T / m
    return 0.5 * (k + k.T)        # an f32 GEMM result is not exactly symmetric


def run(n, rank_frac=2.0, reps=1):
    k = grm_like(n, n, rank_frac)
    kk = k.double()
    kk.diagonal().add_(1e-6)
    for it in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s, ut = pipeline.eigh_from_grm(k, 1e-6)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    smax = float(s.abs().max())
    res = float((ut @ kk - s[:, None] * ut).abs().max()) / smax
    orth = float((ut @ ut.T - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
    asc = bool((s[1:] >= s[:-1]).all())
    ref = torch.linalg.eigvalsh(kk) if n <= 6000 else None
    everr = float((s - ref).abs().max()) / smax if ref is not None else float("nan")
    print(f"n={n} rank_frac={rank_frac}: {dt * 1e3:.1f} ms  residual {res:.2e} orth {orth:.2e} ascending {asc} eigval err {everr:.2e}", flush=True)
    return res < 1e-12 and orth < 1e-12 and asc


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]] or [300, 1000, 2049, 5000]
    ok = True
    for n in sizes:
        ok = run(n, 2.0, 2 if n >= 5000 else 1) and ok
    if len(sys.argv) <= 1:
        ok = run(1000, 0.5) and ok
    print("ALL OK" if ok else "FAILED")
