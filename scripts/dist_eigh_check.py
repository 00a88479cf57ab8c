"""Functional check of the distributed eigensolver (run under torch.distributed.run, any world size).  Default: the
rank-sharded one-stage tridiagonalisation; with JXGPU_EIGH=twostage in the environment: the two-stage path with
column-sharded back-transformations (replicated reduction + divide and conquer, one broadcast per rank at the end).

    JXGPU_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29531 scripts/dist_eigh_check.py 600

Every rank decomposes the same seeded SPD matrix with the symv tiles dealt over the ranks (one all-reduce per column),
checks residual / orthogonality / eigenvalues against a host LAPACK decomposition, and the ranks then compare their
results bit for bit (the replicated part of every column must not diverge).  Prints one "DIST_EIGH_OK ..." line on
rank 0; a non-zero exit code otherwise.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from janusx_amd import pipeline as jp   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    backend = os.environ.get("JXGPU_BENCH_BACKEND", "nccl")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(20251)
    z = rng.standard_normal((n, n + 37))
    k = z @ z.T / z.shape[1] + np.diag(rng.uniform(0.0, 0.3, n))
    kd = torch.from_numpy(k).to(dev)
    assert jp.enable_distributed_eigh(min_n=64)
    w, u = jp.eigh_from_grm(kd, ridge=0.0)            # u: row j = eigenvector j
    torch.cuda.synchronize()
    wh, uh = w.cpu().numpy(), u.cpu().numpy()
    wref = np.linalg.eigvalsh(k)
    scale = np.abs(wref).max()
    eval_err = np.abs(np.sort(wh) - wref).max() / scale
    resid = np.abs(k @ uh.T - uh.T * wh[None, :]).max() / scale
    orth = np.abs(uh @ uh.T - np.eye(n)).max()
    # replicas: bit-identical eigenvalues and eigenvectors on every rank
    digest = torch.tensor([float(np.frombuffer(wh.tobytes(), dtype=np.uint64).sum() % (1 << 52)),
                           float(np.frombuffer(uh.tobytes(), dtype=np.uint64).sum() % (1 << 52))],
                          dtype=torch.float64)
    if backend == "nccl":
        digest = digest.to(dev)
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool((lo == hi).all())
    ok = eval_err < 1e-11 and resid < 1e-11 and orth < 1e-11 and same
    flag = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device=dev if backend == "nccl" else None)
    dist.all_reduce(flag)
    if rank == 0:
        from janusx_amd._lib import lib
        tag = "DIST_EIGH_OK" if flag.item() == 0 else "DIST_EIGH_FAIL"
        # agree: the replicas' checksum comparison in front of the sharded back-transformations of the two-stage path
        # (-1 not run, 1 agreed, 0 differed -> unsharded fallback; JXGPU_DIST_EIGH_TEST_DISAGREE=<rank> forces a difference)
        print(f"{tag} n={n} world={world} eval_err={eval_err:.2e} resid={resid:.2e} orth={orth:.2e} "
              f"replicas_identical={same} agree={int(lib().jxg_eigh_last_dist_agree())} "
              f"band_sharded={int(lib().jxg_eigh_last_band_sharded())} dc_windowed={int(lib().jxg_eigh_last_dc_windowed())}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 0 else 1)


if __name__ == "__main__":
    main()
