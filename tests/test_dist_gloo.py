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


def _pcg_problem():
    from janusx_amd import bed
    from oracle import jx_oracle as O
    n, m = 80, 260
    packed, g = bed.synth_panel_numpy(n, m, seed=5, missing_rate=0.01)
    mi, he, ho = O.row_counts(packed, n)
    af = ((he + 2.0 * ho) / np.maximum(2.0 * (n - mi), 1.0)).astype(np.float32)
    maf = np.minimum(af, 1 - af).astype(np.float32)
    y = bed.synth_phenotype(g, n_causal=12, pve=0.6, seed=5)
    rm, ri, _m_eff = O.rrblup_row_standardization(np.clip(maf, 0, 0.5), np.float32(1e-12))
    lut = O.rrblup_value_lut(rm, ri, np.zeros(m, dtype=bool))
    codes = O.unpack_codes(packed, n)
    z = np.take_along_axis(lut, codes.astype(np.int64), axis=1).astype(np.float32)      # (m, n)
    return n, m, z, y


def _pcg_worker(rank, world, port, q):
    """Marker-sharded PCG of the rrBLUP system (last row; the decomposition csrc/k_pcg.hip uses under
    jx_pcg_set_dist): every rank holds a contiguous range of the markers, Z'p and the dot products are all-reduced."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from janusx_amd import dist as jd
    from oracle import jx_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m, z, y = _pcg_problem()
    lo, hi = jd.shard_range(m, rank, world)
    zs = z[lo:hi]
    lam = np.float32(35.0)
    yc = (y - y.mean()).astype(np.float32)
    b = (zs @ yc).astype(np.float32)
    z64 = zs.astype(np.float64)
    mean = z64.sum(axis=1) / n
    ss = np.maximum((z64 * z64).sum(axis=1) - n * mean * mean, 0.0)
    mu = mean.astype(np.float32)
    dinv = (np.float32(1.0) / np.maximum(ss.astype(np.float32) + lam, np.float32(1e-12))).astype(np.float32)
    ncoll = [0]

    def allsum(a):
        t = torch.from_numpy(np.asarray(a, dtype=np.float64).copy())
        jd.allreduce_sum_(t)
        ncoll[0] += 1
        return t.numpy()

    def dot(u, v):
        return float(allsum([np.dot(u.astype(np.float64), v.astype(np.float64))])[0])

    def apply_a(p):
        part = np.concatenate([zs.T.astype(np.float64) @ p.astype(np.float64),
                               [np.dot(mu.astype(np.float64), p.astype(np.float64))]])
        tot = allsum(part)                                     # ONE collective: Z'p (n values) and mu'p
        xp = tot[:n].astype(np.float32)
        ap = (zs @ xp).astype(np.float32)
        ap = (ap - np.float32(n) * mu * np.float32(tot[n])).astype(np.float32)
        return (ap + lam * p).astype(np.float32)

    beta, conv, iters, rel = O.pcg_solve_f32(b, apply_a, dinv, 200, 1e-7, dot=dot)
    full = torch.zeros(m, dtype=torch.float64)
    full[lo:hi] = torch.from_numpy(beta.astype(np.float64))
    jd.allreduce_sum_(full)
    if rank == 0:
        q.put((full.numpy(), bool(conv), int(iters), ncoll[0]))
    dist.destroy_process_group()


def test_marker_sharded_pcg_matches_single_process():
    """world_size-2 gloo run of the marker-sharded PCG against the single-process solve of the same system: same
    iteration count (+-1), same beta; per iteration one vector collective (Z'p with mu'p) and three scalar ones."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from oracle import jx_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pcg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    beta2, conv2, it2, ncoll = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, m, z, y = _pcg_problem()
    lam = np.float32(35.0)
    yc = (y - y.mean()).astype(np.float32)
    b = (z @ yc).astype(np.float32)
    z64 = z.astype(np.float64)
    mean = z64.sum(axis=1) / n
    ss = np.maximum((z64 * z64).sum(axis=1) - n * mean * mean, 0.0)
    mu = mean.astype(np.float32)
    dinv = (np.float32(1.0) / np.maximum(ss.astype(np.float32) + lam, np.float32(1e-12))).astype(np.float32)

    def apply_a(p):
        xp = (z.T @ p).astype(np.float32)
        ap = (z @ xp).astype(np.float32)
        md = np.float32(float(np.dot(mu.astype(np.float64), p.astype(np.float64))))
        return ((ap - np.float32(n) * mu * md).astype(np.float32) + lam * p).astype(np.float32)

    beta1, conv1, it1, _rel = O.pcg_solve_f32(b, apply_a, dinv, 200, 1e-7)
    assert conv1 and conv2 and abs(it1 - it2) <= 1
    assert np.max(np.abs(beta2 - beta1)) < 2e-5 * np.max(np.abs(beta1))
    assert ncoll <= 4 * it2 + 6            # 1 vector + 3 scalar collectives per iteration, a few before the loop


def test_shard_ranges_tile_the_panel():
    from janusx_amd import dist as jd
    for m in (0, 1, 7, 50000, 500001):
        for world in (1, 2, 3, 8):
            edges = [jd.shard_range(m, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        jd.shard_range(10, 2, 2)
