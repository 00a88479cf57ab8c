"""Functional check of the marker-sharded rrBLUP PCG (run under torch.distributed.run, any world size; ranks may share a GPU
with JXGPU_BENCH_BACKEND=gloo):

    JXGPU_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29533 scripts/dist_pcg_check.py 1500 6000

Every rank solves the same seeded system first alone (all markers) and then with the markers dealt over the ranks
(janusx_amd.dist.enable_distributed_pcg: one all-reduce of an n_train-vector per iteration); the sharded solve must take the
same number of iterations (+-1) and return the same beta / predictions on every rank.  Prints "DIST_PCG_OK ..." on rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from janusx_amd import bed, dist as jd                       # noqa: E402
from janusx_amd import janusx as jxrs                        # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    backend = os.environ.get("JXGPU_BENCH_BACKEND", "nccl")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    packed, g = bed.synth_panel_numpy(n, m, seed=13, missing_rate=0.01)
    y = bed.synth_phenotype(g, n_causal=25, pve=0.5, seed=13)
    counts = jxrs.bed_row_counts(packed, n)
    nm = n - counts[:, 0]
    af = ((counts[:, 1] + 2.0 * counts[:, 2]) / np.maximum(2.0 * nm, 1.0)).astype(np.float32)
    maf = np.minimum(af, 1.0 - af).astype(np.float32)
    flip = af > 0.5
    keep = maf >= 0.02
    idx = np.random.default_rng(2).permutation(n)
    tr, te = np.sort(idx[: n - 200]), np.sort(idx[n - 200:])
    kw = dict(site_keep=keep, lambda_value=float(keep.sum()) * 0.8, tol=1e-7, max_iter=300, packed=packed, packed_n_samples=n,
              maf=maf, row_flip=flip)
    single = jxrs.rrblup_pcg_bed("", tr, y[tr], te, **kw)
    assert jd.enable_distributed_pcg(max_samples=n)
    if len(sys.argv) > 3 and sys.argv[3] == "fail":
        # one rank fails on its own (JXGPU_PCG_TEST_FAIL=<rank>:<where>): EVERY rank must come back with an error from the same
        # collective -- none may stay behind in an all-reduce (the launcher's timeout would end the test)
        raised = 0.0
        try:
            jxrs.rrblup_pcg_bed("", tr, y[tr], te, **kw)
        except RuntimeError as e:
            raised = 1.0
            print(f"rank {rank}: {e}", flush=True)
        cnt = torch.tensor([raised], dtype=torch.float64)
        dist.all_reduce(cnt)
        if rank == 0:
            print(f"{'DIST_PCG_FAIL_TOGETHER_OK' if cnt.item() == world else 'DIST_PCG_FAIL_TOGETHER_BAD'} raised={int(cnt.item())} world={world}",
                  flush=True)
        # the group is still usable: the healthy path runs afterwards
        os.environ.pop("JXGPU_PCG_TEST_FAIL", None)
        dist.destroy_process_group()
        sys.exit(0 if cnt.item() == world else 1)
    shard = jxrs.rrblup_pcg_bed("", tr, y[tr], te, **kw)
    b1, b2 = single[9], shard[9]
    scale = float(np.max(np.abs(b1)))
    beta_err = float(np.max(np.abs(b2 - b1))) / scale
    pred_err = float(np.max(np.abs(shard[1] - single[1]))) / float(np.max(np.abs(single[1])))
    ok = bool(single[3] and shard[3]) and abs(single[4] - shard[4]) <= 1 and beta_err < 2e-5 and pred_err < 2e-5
    # every rank must hold the same result
    dig = torch.tensor([float(np.frombuffer(b2.tobytes(), dtype=np.uint32).sum() % (1 << 40)), float(shard[4])],
                       dtype=torch.float64)
    lo, hi = dig.clone(), dig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool((lo == hi).all())
    flag = torch.tensor([0.0 if (ok and same) else 1.0], dtype=torch.float64)
    dist.all_reduce(flag)
    if rank == 0:
        tag = "DIST_PCG_OK" if flag.item() == 0 else "DIST_PCG_FAIL"
        print(f"{tag} n={n} m_kept={int(keep.sum())} world={world} iters single={single[4]} sharded={shard[4]} "
              f"beta_err={beta_err:.2e} pred_err={pred_err:.2e} ranks_identical={same}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 0 else 1)


if __name__ == "__main__":
    main()
