"""SNP-sharded multi-GPU helpers (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).  The reference has no distributed layer; the partition
follows SURVEY.md §8(e): contiguous SNP ranges per rank, one sum-reduction of the partial n x n GRM
accumulators (+ the scalar denominators), SNP-independent scan afterwards, results gathered in BED order.
"""
from __future__ import annotations

import numpy as np


def shard_range(m: int, rank: int, world: int):
    """Contiguous SNP range [lo, hi) of `rank`; ranges tile [0, m) in BED order."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return (m * rank) // world, (m * (rank + 1)) // world


def allreduce_sum_(t):
    """In-place sum over ranks (no-op when torch.distributed is not initialised)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if t.is_cuda and dist.get_backend() != "nccl":
            # functional multi-rank mode on shared GPUs (gloo): reduce through host memory
            h = t.detach().cpu()
            dist.all_reduce(h)
            t.copy_(h)
        else:
            dist.all_reduce(t)
    return t


def broadcast_(t, src):
    """In-place broadcast from rank `src` (no-op without an initialised multi-rank group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if t.is_cuda and dist.get_backend() != "nccl":
            h = t.detach().cpu()          # functional multi-rank mode on shared GPUs (gloo): through host memory
            dist.broadcast(h, src)
            if dist.get_rank() != src:
                t.copy_(h)
        else:
            dist.broadcast(t, src)
    return t


def gather_rows(local, counts=None):
    """Concatenate per-rank (rows_r, c) tensors in rank order on every rank (BED order of the SNP shards)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def allreduce_grm_accumulator_(acc, chunk_bytes=2 << 30):
    """Sum the (npad, npad) f64 GRM accumulators of the ranks in place, moving only the lower-triangle tiles the GRM kernel
    writes: pack (ti >= tj) tiles -> all-reduce of n (n + 1) / 2 values (in `chunk_bytes` pieces, so the staging buffer
    stays small next to the 20 GB square of BASELINE configs[3]) -> unpack.  Falls back to the plain all-reduce off the
    GPU (gloo CPU tests)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return acc
    if not acc.is_cuda:
        return allreduce_sum_(acc)
    from ._lib import check, lib
    npad = int(acc.shape[0])
    total = int(lib().jxg_tri_tiles_doubles(npad))
    buf = torch.empty(total, dtype=torch.float64, device=acc.device)
    st = torch.cuda.current_stream().cuda_stream
    check(lib().jxg_tri_tiles_pack_f64(acc.data_ptr(), npad, buf.data_ptr(), 0, st))
    step = max(1, int(chunk_bytes) // 8)
    for o in range(0, total, step):
        allreduce_sum_(buf[o:o + step])
    check(lib().jxg_tri_tiles_pack_f64(acc.data_ptr(), npad, buf.data_ptr(), 1, st))
    return acc


_DIST_PCG = {}


def enable_distributed_pcg(max_samples: int = 0):
    """Marker-sharded rrBLUP PCG over the ranks of the initialised torch.distributed group (last row): after
    this call `janusx.rrblup_pcg_bed` deals the kept SNP rows over the ranks in contiguous ranges, every rank streams its own
    range of the payload and ONE all-reduce of an n_train-vector per iteration (RCCL over xGMI with the nccl backend; gloo
    through host memory in the functional tests) completes Z'p; every rank returns the full result.  `max_samples`: largest
    n_train / n_test to be solved (sizes the staging buffer; 0: 2^20).  Returns False (and switches the mode off) without a
    multi-rank group."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from ._lib import check, lib
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        check(lib().jx_pcg_set_dist(0, 1, None, None, None, 0))
        _DIST_PCG.clear()
        return False
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    cap = max(int(max_samples), 1 << 20) + 16
    staging = torch.zeros(cap, dtype=torch.float64, device=dev)

    def _cb(_user):
        try:
            allreduce_sum_(staging[: int(lib().jx_pcg_dist_count())])
            return 0
        except Exception as e:   # noqa: BLE001 - reported through the C status
            import sys
            print(f"distributed PCG: all-reduce failed on rank {rank}: {e!r}", file=sys.stderr, flush=True)
            return 1

    cb = C.CFUNCTYPE(C.c_int, C.c_void_p)(_cb)
    check(lib().jx_pcg_set_dist(rank, world, C.cast(cb, C.c_void_p), None, staging.data_ptr(), cap))
    _DIST_PCG.update(cb=cb, staging=staging, rank=rank, world=world)
    return True


def distributed_pcg():
    """(rank, world) of the marker-sharded PCG mode, or None when it is off."""
    return (_DIST_PCG["rank"], _DIST_PCG["world"]) if _DIST_PCG else None


_COLLECTIVE_SIZE_CHECKS = {"on": False}


def enable_collective_size_checks(on: bool = True):
    """State that the sparse-GRM routes of `janusx` (sparse REML, SparseLMM) are called by EVERY rank of the process group with the
    same arguments.  Their HBM size check then takes the minimum of the free HBM over the ranks (one MIN all-reduce), so that all
    ranks accept or refuse together.  Off by default: the routes may equally be called on one rank only (`if rank == 0:` in a
    torchrun job) or be answered from the spectral cache on some ranks, where a hidden collective would leave the other ranks
    waiting for ever; the check then uses the local figure.  -> the previous setting."""
    prev = _COLLECTIVE_SIZE_CHECKS["on"]
    _COLLECTIVE_SIZE_CHECKS["on"] = bool(on)
    return prev


def collective_size_checks():
    return _COLLECTIVE_SIZE_CHECKS["on"]
