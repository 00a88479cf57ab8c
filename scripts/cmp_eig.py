import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, subprocess
from janusx_amd._lib import lib, check
n = int(sys.argv[1])
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
z = torch.randn((n, n + 64), generator=g, device=dev, dtype=torch.float32)
k = (z @ z.T / (n + 64)).to(torch.float64); k = 0.5 * (k + k.T); del z
def run():
    a = k.clone(); w = torch.empty(n, dtype=torch.float64, device=dev)
    check(lib().jxg_eigh_f64(a.data_ptr(), n, 1e-6, w.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return w, a
w1, a1 = run()
os.environ["JXGPU_STEDC"] = "rocsolver"; os.environ["JXGPU_ORMTR"] = "rocsolver"
w2, a2 = run()
print("n", n, "max |dw|", float((w1 - w2).abs().max()), "rel", float(((w1 - w2).abs() / w2.abs().clamp_min(1e-300)).max()))
# subspace agreement: |<u1_i, u2_i>| for well separated eigenvalues (top 50)
d = (a1[-50:] * a2[-50:]).sum(dim=1).abs()
print("min |<u_own, u_rocsolver>| over the top 50:", float(d.min()))
