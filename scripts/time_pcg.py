"""Times the two matrix-free halves of the PCG operator (jxg_packed_dot: Z'v over SNPs -> samples, jxg_packed_tdot:
Z u over samples -> SNPs) and a whole rrblup_pcg_bed solve. GPU box only.
usage: time_pcg.py n m"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from janusx_amd import pipeline
from janusx_amd._lib import lib, check


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    dev = torch.device("cuda:0")
    packed, _ = bench.synth_panel_gpu(n, m, 20260609, dev, missing_rate=0.0)
    p = pipeline.Panel(packed, n)
    st = torch.cuda.current_stream().cuda_stream
    lut = torch.randn((m, 4), device=dev, dtype=torch.float32)
    lut[:, 1] = 0
    beta = torch.randn(m, device=dev, dtype=torch.float64)
    alpha = torch.randn(n, device=dev, dtype=torch.float64)
    on = torch.empty(n, device=dev, dtype=torch.float64)
    om = torch.empty(m, device=dev, dtype=torch.float64)
    payload = n * m / 4.0
    t32 = torch.empty(int(lib().jxg_t32_bytes(n, m)), dtype=torch.uint8, device=dev)
    work = torch.empty(16 * m + 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().jxg_p32_transpose(p.p32.data_ptr(), p.m, n, None, m, t32.data_ptr(), st))
    e1.record()
    torch.cuda.synchronize()
    print(f"p32 -> sample-major transpose: {e0.elapsed_time(e1):.3f} ms", flush=True)
    for name, fn in (("packed_dot  (Z'v -> n)", lambda: lib().jxg_packed_dot(p.p32.data_ptr(), p.m, n, None, m, lut.data_ptr(), beta.data_ptr(), on.data_ptr(), st)),
                     ("packed_dot_t32 (Z'v, bit-plane tables)", lambda: lib().jxg_packed_dot_t32(t32.data_ptr(), n, m, lut.data_ptr(), beta.data_ptr(), work.data_ptr(), on.data_ptr(), st)),
                     ("packed_tdot (Z u -> m)", lambda: lib().jxg_packed_tdot(p.p32.data_ptr(), p.m, n, None, m, lut.data_ptr(), alpha.data_ptr(), om.data_ptr(), st)),
                     ("packed_tdot_f32 (Z u, bit-plane tables)", lambda: lib().jxg_packed_tdot_f32(p.p32.data_ptr(), p.m, n, None, m, lut.data_ptr(), alpha.data_ptr(), om.data_ptr(), st))):
        for _ in range(2):
            check(fn())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 10
        for _ in range(reps):
            check(fn())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"n={n} m={m} {name}: {ms:.3f} ms  {payload / ms / 1e6:.0f} GB/s of payload ({payload / 1e6:.0f} MB)", flush=True)
    # reference values for the two products from a dense decode of a slice
    codes = torch.stack([(packed[:256].to(torch.int64) >> (2 * k)) & 3 for k in range(4)], dim=2).reshape(256, -1)[:, :n]
    z = torch.gather(lut[:256].to(torch.float64), 1, codes)
    check(lib().jxg_packed_tdot(p.p32.data_ptr(), p.m, n, None, m, lut.data_ptr(), alpha.data_ptr(), om.data_ptr(), st))
    torch.cuda.synchronize()
    print("tdot slice err", float((z @ alpha - om[:256]).abs().max()))
    # Z'v: full reference on the first 64 samples through a dense decode in chunks
    b32 = beta.to(torch.float32)
    ref = torch.zeros(64, dtype=torch.float64, device=dev)
    for r0 in range(0, m, 8192):
        pk = packed[r0:r0 + 8192, :16].to(torch.int64)
        cd = torch.stack([(pk >> (2 * k)) & 3 for k in range(4)], dim=2).reshape(pk.shape[0], -1)[:, :64]
        wv = (lut[r0:r0 + 8192] * b32[r0:r0 + 8192, None]).to(torch.float64)
        ref += torch.gather(wv, 1, cd).sum(0)
    check(lib().jxg_packed_dot_t32(t32.data_ptr(), n, m, lut.data_ptr(), beta.data_ptr(), work.data_ptr(), on.data_ptr(), st))
    torch.cuda.synchronize()
    print("dot_t32 rel err (64 samples)", float((ref - on[:64]).abs().max() / ref.abs().max()))
    a32 = alpha.to(torch.float32).to(torch.float64)
    check(lib().jxg_packed_tdot_f32(p.p32.data_ptr(), p.m, n, None, m, lut.data_ptr(), alpha.data_ptr(), om.data_ptr(), st))
    torch.cuda.synchronize()
    ref = z @ a32
    print("tdot_f32 slice rel err", float((ref - om[:256]).abs().max() / ref.abs().max()))
    if len(sys.argv) > 3:
        from janusx_amd import janusx as jxrs
        pk = packed.cpu().numpy()
        maf = np.full(m, 0.25, dtype=np.float32)
        cnt = p.counts()
        nm = n - cnt[:, 0]
        pf = (cnt[:, 1] + 2 * cnt[:, 2]) / (2.0 * nm)
        maf = np.minimum(pf, 1 - pf).astype(np.float32)
        flip = pf > 0.5
        y = np.random.default_rng(0).standard_normal(n)
        t0 = time.perf_counter()
        r = jxrs.rrblup_pcg_bed("", np.arange(n), y, packed=pk, packed_n_samples=n, maf=maf, row_flip=flip,
                                lambda_value=float(m), tol=1e-4, max_iter=100)
        dt = time.perf_counter() - t0
        print(f"rrblup_pcg_bed n={n} m={m}: {dt:.3f} s, iters={r[4]} converged={r[3]} rel_res={r[5]:.3g}")


main()
