"""GPU check + timing of the two-stage reduction building blocks (run on the GPU box):
python scripts/check_twostage.py [gemm] [st] [time]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from janusx_amd._lib import check, lib  # noqa: E402

dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731


def colmajor(t):
    """torch (m, n) -> column-major buffer (stored as the transposed contiguous tensor)."""
    return t.T.contiguous()


def gemm_checks():
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    worst = 0.0
    for (m, n, k, ta, tb, ks) in [(300, 200, 150, 0, 0, 1), (257, 64, 1000, 1, 0, 0), (64, 64, 5000, 1, 0, 0),
                                  (1000, 64, 128, 0, 0, 1), (513, 300, 77, 0, 1, 1), (130, 129, 65, 1, 1, 1),
                                  (2000, 2000, 512, 0, 0, 0), (37, 5, 1000, 1, 0, 0), (900, 64, 64, 0, 0, 1)]:
        a = torch.randn((k, m) if ta else (m, k), generator=g, device=dev, dtype=torch.float64)
        b = torch.randn((n, k) if tb else (k, n), generator=g, device=dev, dtype=torch.float64)
        c = torch.randn((m, n), generator=g, device=dev, dtype=torch.float64)
        ref = 0.7 * (a.T if ta else a) @ (b.T if tb else b) + 0.3 * c
        ac, bc, cc = colmajor(a), colmajor(b), colmajor(c)
        check(lib().jxg_dgemm_f64(ta, tb, m, n, k, 0.7, ac.data_ptr(), a.shape[0], bc.data_ptr(), b.shape[0], 0.3,
                                  cc.data_ptr(), m, ks, st()))
        err = float((cc.T - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        worst = max(worst, err)
        print(f"dgemm m={m} n={n} k={k} ta={ta} tb={tb} ksplit={ks}: rel err {err:.2e}")
    for (m, n) in [(300, 64), (1000, 64), (129, 30), (2500, 64), (700, 128)]:
        a = torch.randn((m, m), generator=g, device=dev, dtype=torch.float64)
        a = a + a.T
        b = torch.randn((m, n), generator=g, device=dev, dtype=torch.float64)
        ref = a @ b
        al = torch.tril(a) + torch.triu(torch.full_like(a, float("nan")), 1)     # strict upper part must not be read
        ac, bc = colmajor(al), colmajor(b)
        cc = torch.zeros((n, m), device=dev, dtype=torch.float64)
        check(lib().jxg_dsymm_lower_f64(m, n, 1.0, ac.data_ptr(), m, bc.data_ptr(), m, 0.0, cc.data_ptr(), m, st()))
        err = float((cc.T - ref).abs().max()) / float(ref.abs().max())
        worst = max(worst, err)
        print(f"dsymm m={m} n={n}: rel err {err:.2e}")
    for (m, k) in [(300, 128), (1000, 128), (129, 40)]:
        a = torch.randn((m, k), generator=g, device=dev, dtype=torch.float64)
        b = torch.randn((m, k), generator=g, device=dev, dtype=torch.float64)
        c = torch.randn((m, m), generator=g, device=dev, dtype=torch.float64)
        ref = torch.tril(c - a @ b.T)
        ac, bc, cc = colmajor(a), colmajor(b), colmajor(c)
        check(lib().jxg_dsyr2k_lower_nt_f64(m, k, -1.0, ac.data_ptr(), m, bc.data_ptr(), m, 1.0, cc.data_ptr(), m, st()))
        got = cc.T
        err = float((torch.tril(got) - ref).abs().max()) / float(ref.abs().max())
        upper_untouched = bool(torch.equal(torch.triu(got, 1), torch.triu(c, 1)))
        worst = max(worst, err)
        print(f"dsyr2k m={m} k={k}: rel err {err:.2e} upper untouched {upper_untouched}")
    print("gemm worst", worst)
    return worst < 1e-13


def make_grm_like(n, seed, rank_frac=2.0):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    m = int(n * rank_frac)
    z = torch.randn((n, m), generator=g, device=dev, dtype=torch.float64)
    a = z @ z.T / m
    a.diagonal().add_(1e-6)
    return a


def st_checks():
    from scipy.linalg import eigvalsh_tridiagonal
    ok = True
    for n, rf in [(70, 2.0), (300, 2.0), (1000, 0.5), (2049, 2.0), (5000, 2.0)]:
        a = make_grm_like(n, n, rf)
        ev = torch.linalg.eigvalsh(a).cpu().numpy()
        w = a.clone()
        d = torch.empty(n, device=dev, dtype=torch.float64)
        e = torch.zeros(n, device=dev, dtype=torch.float64)
        ab = torch.zeros((n, 128), device=dev, dtype=torch.float64)
        fl = np.zeros(4, dtype=np.int32)
        check(lib().jxg_sy2st_f64(w.data_ptr(), n, d.data_ptr(), e.data_ptr(), ab.data_ptr(), fl.ctypes.data, st()))
        # band matrix -> dense
        abh = ab.cpu().numpy()       # [j][d]
        band = np.zeros((n, n))
        for dd in range(0, min(65, n)):
            idx = np.arange(n - dd)
            band[idx + dd, idx] = abh[idx, dd]
            band[idx, idx + dd] = abh[idx, dd]
        evb = np.linalg.eigvalsh(band)
        evt = eigvalsh_tridiagonal(d.cpu().numpy(), e.cpu().numpy()[: n - 1])
        eb = np.abs(evb - ev).max() / np.abs(ev).max()
        et = np.abs(evt - ev).max() / np.abs(ev).max()
        print(f"n={n}: flags {fl[:2]}, band eig rel err {eb:.2e}, tridiagonal eig rel err {et:.2e}")
        ok = ok and fl[0] == 0 and fl[1] == 0 and eb < 1e-12 and et < 1e-12
    return ok


def timing():
    for n in (5000, 20000):
        a = make_grm_like(n, 3, 2.0)
        d = torch.empty(n, device=dev, dtype=torch.float64)
        e = torch.zeros(n, device=dev, dtype=torch.float64)
        fl = np.zeros(4, dtype=np.int32)
        for it in range(2):
            w = a.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            check(lib().jxg_sy2st_f64(w.data_ptr(), n, d.data_ptr(), e.data_ptr(), None, fl.ctypes.data, st()))
            torch.cuda.synchronize()
            print(f"n={n}: sy2st {1e3 * (time.perf_counter() - t0):.1f} ms flags {fl[:2]}")


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "st", "time"]
    good = True
    if "gemm" in what:
        good = gemm_checks() and good
    if "st" in what:
        good = st_checks() and good
    if "time" in what:
        timing()
    print("ALL OK" if good else "FAILED")
