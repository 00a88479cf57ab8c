"""Debug helper: eigenvectors of one matrix under two slab widths of the balanced Q2 form; which columns differ."""
import os, sys, subprocess, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "child":
    sys.path.insert(0, root)
    import torch
    from janusx_amd._lib import lib, check
    n = int(sys.argv[2])
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    z = torch.randn((n, n + 64), generator=g, device="cuda", dtype=torch.float32)
    k = (z @ z.T / (n + 64)).to(torch.float64); k = 0.5 * (k + k.T)
    a = k.clone(); w = torch.empty(n, dtype=torch.float64, device="cuda")
    check(lib().jxg_eigh_f64(a.data_ptr(), n, 1e-6, w.data_ptr(), torch.cuda.current_stream().cuda_stream))
    np.save(sys.argv[3], a.cpu().numpy())
    sys.exit(0)
n = int(sys.argv[1]); pa, pb = sys.argv[2], sys.argv[3]
outs = []
for per in (pa, pb):
    f = f"/tmp/u_{per}.npy"
    env = dict(os.environ, JXGPU_SBBACK_BAL_MIN="0", JXGPU_SBBACK_BAL_PER=per)
    subprocess.run([sys.executable, __file__, "child", str(n), f], env=env, check=True)
    outs.append(np.load(f))
ua, ub = outs           # rows = eigenvectors = columns of C
d = np.abs(ua - ub).max(axis=1)      # per eigenvector (column of C)
bad = np.nonzero(d > 1e-9)[0]
print("n", n, "bad columns", len(bad), "of", n)
if len(bad):
    units = np.unique(bad // 16)
    per = int(pb)
    print("bad units (index, slab, position in slab):", [(int(u), int(u) // per, int(u) % per) for u in units[:40]])
    print("bad column offsets within unit:", np.unique(bad % 16))
    rows = np.nonzero(np.abs(ua[bad[0]] - ub[bad[0]]) > 1e-9)[0]
    print("first bad column", bad[0], "bad rows from", rows[:5], "to", rows[-5:], "count", len(rows))
