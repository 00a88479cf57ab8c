"""GPU timing of the f64 MFMA GEMM family (k_dgemm.hip) at the shapes the eigensolver uses."""
import sys
import torch
sys.path.insert(0, ".")
from janusx_amd._lib import check, lib
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    L = lib()
    for (m, n, k, ta, tb) in [(8192, 8192, 8192, 0, 0), (8192, 8192, 1024, 0, 0), (1024, 8192, 8192, 1, 0), (20000, 64, 128, 0, 0),
                              (64, 64, 20000, 1, 0), (20000, 20000, 1024, 0, 0)]:
        a = torch.randn((k, m) if not ta else (m, k), device=dev, dtype=torch.float64)   # column-major buffers
        b = torch.randn((n, k) if not tb else (k, n), device=dev, dtype=torch.float64)
        c = torch.zeros((n, m), device=dev, dtype=torch.float64)
        lda = k if ta else m
        ldb = n if tb else k
        ms = timeit(lambda: check(L.jxg_dgemm_f64(ta, tb, m, n, k, 1.0, a.data_ptr(), lda, b.data_ptr(), ldb, 0.0, c.data_ptr(), m, 0, st)))
        # torch reference (rocBLAS): same math in row-major terms
        ms_t = timeit(lambda: torch.matmul(a.T, b.T)) if (not ta and not tb and m * n * k < 2e12) else float("nan")
        print(f"dgemm m={m} n={n} k={k} ta={ta} tb={tb}: {ms:.3f} ms = {2.0 * m * n * k / ms / 1e9:.1f} TFLOP/s   (torch f64 matmul of the same size: {ms_t:.3f} ms)")
    for nt in (5000, 20000):
        a = torch.randn((nt, nt), device=dev, dtype=torch.float64)
        v = torch.randn((64, nt), device=dev, dtype=torch.float64)
        z = torch.zeros((64, nt), device=dev, dtype=torch.float64)
        ms = timeit(lambda: check(L.jxg_dsymm_lower_f64(nt, 64, 1.0, a.data_ptr(), nt, v.data_ptr(), nt, 0.0, z.data_ptr(), nt, st)))
        print(f"dsymm nt={nt} n=64: {ms:.3f} ms = {2.0 * nt * nt * 64 / ms / 1e9:.1f} TFLOP/s, {8.0 * nt * nt / ms / 1e6:.0f} GB/s of A (full square)")
        p1 = torch.randn((128, nt), device=dev, dtype=torch.float64)
        p2 = torch.randn((128, nt), device=dev, dtype=torch.float64)
        ms = timeit(lambda: check(L.jxg_dsyr2k_lower_nt_f64(nt, 128, -1.0, p1.data_ptr(), nt, p2.data_ptr(), nt, 1.0, a.data_ptr(), nt, st)))
        print(f"dsyr2k nt={nt} k=128: {ms:.3f} ms = {nt * (nt + 1.0) * 128 / ms / 1e9:.1f} TFLOP/s, {8.0 * nt * nt / ms / 1e6:.0f} GB/s of C (read + write of the triangle)")


main()
