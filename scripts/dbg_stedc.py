import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from janusx_amd._lib import check, lib
from scipy.linalg import eigvalsh_tridiagonal
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
g = torch.Generator(device=dev); g.manual_seed(n)
z = torch.randn((n, 2 * n), generator=g, device=dev, dtype=torch.float32)
k = (z @ z.T / (2 * n)).double(); k.diagonal().add_(1e-6)
w = k.clone()
d = torch.empty(n, device=dev, dtype=torch.float64); e = torch.zeros(n, device=dev, dtype=torch.float64)
fl = np.zeros(4, dtype=np.int32)
st = torch.cuda.current_stream().cuda_stream
check(lib().jxg_sy2st_f64(w.data_ptr(), n, d.data_ptr(), e.data_ptr(), None, fl.ctypes.data, st))
dh, eh = d.cpu().numpy(), e.cpu().numpy()[: n - 1]
ref = eigvalsh_tridiagonal(dh, eh)
print("e: min|e|", np.abs(eh).min(), "max|e|", np.abs(eh).max(), "n negative", (eh < 0).sum())
t = torch.diag(d) + torch.diag(e[: n - 1], 1) + torch.diag(e[: n - 1], -1)
for mode in ("default", "rocsolver"):
    if mode == "rocsolver":
        os.environ["JXGPU_STEDC"] = "rocsolver"
    os.environ["JXGPU_EIGH"] = "onestage"
    a = t.clone(); wv = torch.empty(n, device=dev, dtype=torch.float64)
    check(lib().jxg_eigh_f64(a.data_ptr(), n, 0.0, wv.data_ptr(), st))
    err = np.abs(wv.cpu().numpy() - ref)
    i = int(err.argmax())
    print(mode, "stedc on the chased tridiagonal: max err", err.max(), "at index", i, "of", n, "eig", ref[i], " #err>1e-12:", int((err > 1e-12).sum()))
