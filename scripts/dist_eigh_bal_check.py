"""Two (or more) ranks on the one device: the column-sharded back-transformations at a size where a rank's share takes the
balanced Q2 form (n = 20 000, two ranks: 10 000 columns = 625 units -> slabs of four).  Device-side invariants + bit equality of
the ranks' results.  JXGPU_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... scripts/dist_eigh_bal_check.py 20000"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from janusx_amd import pipeline as jp   # noqa: E402
from janusx_amd._lib import lib          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
backend = os.environ.get("JXGPU_BENCH_BACKEND", "nccl")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
dist.init_process_group(backend=backend, rank=rank, world_size=world)
dev = torch.device("cuda", torch.cuda.current_device())
g = torch.Generator(device=dev); g.manual_seed(5)
z = torch.randn((n, n + 40), generator=g, device=dev, dtype=torch.float32)
k = (z @ z.T / (n + 40)).to(torch.float64); k = 0.5 * (k + k.T); del z
assert jp.enable_distributed_eigh(min_n=64)
w, u = jp.eigh_from_grm(k, ridge=0.0)
torch.cuda.synchronize()
form = int(round(lib().jxg_last_kernel_ms(17)))
sc = float(w.abs().max())
res = float((u @ k - w[:, None] * u).abs().max()) / sc
orth = float((u @ u.T - torch.eye(n, device=dev, dtype=torch.float64)).abs().max())
chk = torch.tensor([float(u.sum()), float((u * u).sum()), float(w.sum())], dtype=torch.float64)
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
same = bool((lo == hi).all())
ok = res < 1e-10 and orth < 1e-10 and same
if rank == 0:
    print(f"DIST_EIGH_BAL {'OK' if ok else 'FAILED'} n={n} world={world} q2_form={form} resid={res:.2e} orth={orth:.2e} ranks_identical={same}", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
