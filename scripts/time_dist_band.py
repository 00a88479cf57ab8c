"""Stage times of the eigendecomposition with the SHARDED band reduction's launch sequence and collectives on ONE rank (RCCL,
JXGPU_DIST_EIGH_FORCE=1 JXGPU_EIGH=twostage): what the per-block-row launches, the staging copies and the two ncclAllReduce
calls per panel cost before any wire time.  usage: time_dist_band.py n"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
import torch
import torch.distributed as dist
from janusx_amd import pipeline as jp
from janusx_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
z = torch.randn((n, n + 64), generator=g, device=dev, dtype=torch.float32)
k = (z @ z.T / (n + 64)).to(torch.float64); k = 0.5 * (k + k.T); del z
on = jp.enable_distributed_eigh(min_n=1 << 30)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    w, u = jp.eigh_from_grm(k, 1e-6)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = {nm: round(float(lib().jxg_last_kernel_ms(i)), 1) for i, nm in ((6, "band"), (7, "chase"), (8, "dc"), (9, "q1"), (4, "q2_kernel"))}
kk = k + 1e-6 * torch.eye(n, device=dev, dtype=torch.float64)
res = float((u[:256] @ kk - w[:256, None] * u[:256]).abs().max())
print(f"n={n} dist_enabled={on} band_sharded={int(lib().jxg_eigh_last_band_sharded())} eigh {dt*1e3:.1f} ms resid(256 rows) {res:.2e} stages_ms {st}", flush=True)
dist.destroy_process_group()
