"""GPU check + timing of the sliced int8 GEMM (k_ozgemm.hip) at the shapes the eigensolver uses."""
import ctypes
import sys
import torch
sys.path.insert(0, ".")
from janusx_amd._lib import check, lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def main():
    L = lib()
    print("planes", L.jxg_oz_planes())
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    shapes = [(300, 200, 177, 0, 0), (257, 129, 1000, 1, 0), (130, 390, 64, 0, 1), (515, 77, 333, 1, 1), (2048, 4096, 3000, 1, 0)]
    if len(sys.argv) > 1:
        shapes += [(8192, 8192, 8192, 0, 0), (2048, 20000, 10000, 1, 0), (10000, 20000, 2048, 0, 0), (10000, 10000, 10000, 0, 0),
                   (2048, 20000, 2048, 0, 0), (2048, 2048, 10000, 1, 0), (4096, 4096, 30000, 1, 0)]
    for (m, n, k, ta, tb) in shapes:
        a = torch.randn((k, m) if not ta else (m, k), device=dev, dtype=torch.float64, generator=g)
        b = torch.randn((n, k) if not tb else (k, n), device=dev, dtype=torch.float64, generator=g)
        # rows / columns of very different magnitude
        a *= torch.exp(3.0 * torch.randn((1, a.shape[1]) if not ta else (a.shape[0], 1), device=dev, dtype=torch.float64, generator=g))
        c0 = torch.randn((n, m), device=dev, dtype=torch.float64, generator=g)
        c = c0.clone()
        ms = (ctypes.c_float * 3)()
        lda = k if ta else m
        ldb = n if tb else k
        best = None
        for rep in range(3):
            c.copy_(c0)
            check(L.jxg_oz_dgemm_f64(ta, tb, m, n, k, 1.5, a.data_ptr(), lda, b.data_ptr(), ldb, 0.5, c.data_ptr(), m, ms, st))
            if best is None or ms[2] < best[2]:
                best = list(ms)
        opa = a.T if not ta else a
        opb = b.T if not tb else b
        err = None
        if m * n * k < 3e12:
            ref = 1.5 * (opa @ opb) + 0.5 * c0.T
            # scale of an entry: |row of op(A)| . |column of op(B)|
            den = (opa.abs() @ opb.abs()).clamp_min(1e-300)
            err = float(((c.T - ref).abs() / den).max())
            nrm = float((c.T - ref).norm() / ref.norm())
        tf = 2.0 * m * n * k / best[2] / 1e9
        print(f"oz m={m} n={n} k={k} ta={ta} tb={tb}: slice {best[0]:.3f} + {best[1]:.3f} ms, product {best[2]:.3f} ms = {tf:.1f} TFLOP/s-equivalent"
              + (f", max err / (|a|.|b|) {err:.2e}, Frobenius {nrm:.2e}" if err is not None else ""))


main()
