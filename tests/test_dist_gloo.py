"""world_size-2 gloo test of the SNP-sharded path (CPU): partial GRM accumulators summed by all-reduce equal
the single-process GRM; scan shards gathered in rank order reproduce BED order."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from janusx_amd import bed, dist as jd, stats
    from oracle import jx_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m = 96, 501
    packed, g = bed.synth_panel_numpy(n, m, seed=77, missing_rate=0.02)
    lo, hi = jd.shard_range(m, rank, world)
    sub = packed[lo:hi]
    mi, he, ho = O.row_counts(sub, n)
    cnt = np.stack([mi, he, ho], 1)
    keep, mean_g, scale, flip, var = stats.stream_grm_row_prepare(cnt, n, 1, 0.02, 0.05, 0.0)
    rows = np.nonzero(keep)[0]
    lut = stats.grm_lut_from_mean_scale(mean_g[rows], scale[rows], flip[rows])
    codes = O.unpack_codes(sub[rows], n)
    z = np.stack([lut[k][codes[k]] for k in range(len(rows))]) if len(rows) else np.zeros((0, n), np.float32)
    acc = torch.from_numpy((z.T @ z).astype(np.float64))
    den = torch.tensor([float(var[rows].sum()), float(len(rows))], dtype=torch.float64)
    jd.allreduce_sum_(acc)
    jd.allreduce_sum_(den)
    k = (acc.numpy() / den[0].item()).astype(np.float32)
    # shard-local "scan result": row ids, gathered in rank order
    local = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1)
    allrows = jd.gather_rows(local).numpy().ravel()
    if rank == 0:
        q.put((k, float(den[1]), allrows))
    dist.destroy_process_group()


def test_snp_sharded_grm_matches_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from janusx_amd import bed
    from oracle import jx_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    k, eff, allrows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, m = 96, 501
    packed, g = bed.synth_panel_numpy(n, m, seed=77, missing_rate=0.02)
    ref, eff_ref, _ = O.grm_stream_bed(packed, n, 1, 0.02, 0.05, 0.0)
    assert eff == eff_ref
    assert np.max(np.abs(k - ref)) < 1e-5
    assert np.array_equal(allrows, np.arange(m, dtype=np.float64))


def test_shard_ranges_tile_the_panel():
    from janusx_amd import dist as jd
    for m in (0, 1, 7, 50000, 500001):
        for world in (1, 2, 3, 8):
            edges = [jd.shard_range(m, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        jd.shard_range(10, 2, 2)
