"""Timing of the sparse-GRM threshold / compaction kernels on a synthetic accumulator (GPU box):
    python scripts/time_spgrm.py [n] [threshold]
Prints ms and GB/s of the count and fill passes (algorithmic bytes = 8 B x lower triangle per pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from janusx_amd._lib import check, lib   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
    dev = torch.device("cuda", 0)
    L = lib()
    ld = int(L.jxg_num_tiles(n)) * 128
    g = torch.Generator(device=dev).manual_seed(1)
    acc = torch.randn((ld, ld), dtype=torch.float64, device=dev, generator=g) * 0.02    # ~0.6 % above 0.05
    acc.diagonal().fill_(1.0)
    work = torch.empty(int(L.jxg_spgrm_work_bytes(n)), dtype=torch.uint8, device=dev)
    colptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    p = lambda t: t.data_ptr()   # noqa: E731
    for rep in range(3):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        check(L.jxg_spgrm_count(p(acc), n, 1.0, thr, 0, p(work), p(colptr), None))
        e[1].record()
        nnz = int(colptr[n].item())
        rows = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        vals = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        e[1].record()
        check(L.jxg_spgrm_fill(p(acc), n, 1.0, thr, 0, p(work), p(colptr), p(rows), p(vals), None))
        e[2].record()
        torch.cuda.synchronize()
        tri = 8.0 * n * (n + 1) / 2
        t_fill = e[1].elapsed_time(e[2])
        print(f"n={n} thr={thr} nnz={nnz} ({nnz / (n * (n + 1) / 2):.4f} of the triangle): "
              f"fill {t_fill:.3f} ms = {tri / t_fill / 1e6:.0f} GB/s (+ {12.0 * nnz / 1e6:.1f} MB written)")
    # count pass alone (includes its two small scan kernels and the flag read-back)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        check(L.jxg_spgrm_count(p(acc), n, 1.0, thr, 0, p(work), p(colptr), None))
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5 * 1e3
    print(f"count (+ band prefix, column scan, synchronise): {t:.3f} ms = {8.0 * n * (n + 1) / 2 / t / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
