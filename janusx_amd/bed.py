"""PLINK BED/BIM/FAM payload IO and the synthetic panel generator used by tests and bench.py.

BED layout (src/stats/lmm.rs:1050-1061, src/math/bedmath.rs:20-27): 3 magic bytes ``6c 1b 01`` then m rows of
``ceil(n/4)`` bytes, SNP-major; sample j sits in byte j>>2 at bits 2*(j&3); 00 -> 0, 10 -> 1, 11 -> 2,
01 -> missing; dosage counts the BIM column-6 allele (src/io/gfcore.rs:1470-1476).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

BED_MAGIC = bytes([0x6C, 0x1B, 0x01])


@dataclass
class Bim:
    chrom: list
    snp: list
    pos: list
    a0: list  # column 5 (ref_allele -> allele0)
    a1: list  # column 6 (alt_allele -> allele1)


def read_fam_ids(prefix):
    """Sample IDs = FAM column 2 (src/io/gfcore.rs:307-323)."""
    ids = []
    with open(f"{prefix}.fam") as fh:
        for line in fh:
            parts = line.split()
            if parts:
                ids.append(parts[1] if len(parts) > 1 else parts[0])
    return ids


def read_bim(prefix) -> Bim:
    chrom, snp, pos, a0, a1 = [], [], [], [], []
    with open(f"{prefix}.bim") as fh:
        for line in fh:
            parts = line.split()
            if len(parts) < 6:
                continue
            chrom.append(parts[0])
            snp.append(parts[1])
            pos.append(int(parts[3]))
            a0.append(parts[4])
            a1.append(parts[5])
    return Bim(chrom, snp, pos, a0, a1)


def read_bed_payload(prefix):
    """-> (packed (m, bps) uint8 memmap view, n_samples, Bim)."""
    ids = read_fam_ids(prefix)
    n = len(ids)
    bim = read_bim(prefix)
    m = len(bim.snp)
    bps = (n + 3) // 4
    path = f"{prefix}.bed"
    with open(path, "rb") as fh:
        magic = fh.read(3)
    if magic != BED_MAGIC:
        raise RuntimeError(f"{path}: not a SNP-major PLINK .bed (bad magic)")
    size = os.path.getsize(path)
    if size != 3 + m * bps:
        raise RuntimeError(f"{path}: size {size} != 3 + {m}*{bps}")
    packed = np.memmap(path, dtype=np.uint8, mode="r", offset=3, shape=(m, bps))
    return packed, n, bim


def stage_bed_payload(prefix, mmap_window_mb=None, device=None):
    """PLINK .bed payload -> (m, bps) uint8 tensor in HBM, staged window by window: the counterpart of the reference's
    `WindowedBedMatrix` (src/io/gload.rs:523-640; `mmap_window_mb` of src/stats/lmm.rs:2488-2520, :1046-1049: windows of
    that many MiB of the file instead of one mapping of all of it).  The file is read `mmap_window_mb` MiB at a time (default
    256) into two pinned staging buffers that alternate, each window copied to its rows of the device tensor while the next
    one is read: no host copy of the payload exists beyond the two windows (BASELINE configs[4] is 50 GB packed), and the
    host-layer entry points take the device tensor in place.  -> (tensor, n_samples, Bim)."""
    import torch
    ids = read_fam_ids(prefix)
    n = len(ids)
    bim = read_bim(prefix)
    m = len(bim.snp)
    bps = (n + 3) // 4
    path = f"{prefix}.bed"
    if mmap_window_mb is not None and int(mmap_window_mb) <= 0:
        raise RuntimeError("mmap_window_mb must be > 0")
    with open(path, "rb") as fh:
        if fh.read(3) != BED_MAGIC:
            raise RuntimeError(f"{path}: not a SNP-major PLINK .bed (bad magic)")
        size = os.path.getsize(path)
        if size != 3 + m * bps:
            raise RuntimeError(f"{path}: size {size} != 3 + {m}*{bps}")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        out = torch.empty((m, bps), dtype=torch.uint8, device=dev)
        if m == 0:
            return out, n, bim
        win_rows = max(1, (int(mmap_window_mb or 256) << 20) // max(bps, 1))
        win_rows = min(win_rows, m)
        bufs = [torch.empty((win_rows, bps), dtype=torch.uint8).pin_memory() for _ in range(2 if m > win_rows else 1)]
        events = [None] * len(bufs)
        stream = torch.cuda.current_stream(dev)
        for w, r0 in enumerate(range(0, m, win_rows)):
            r1 = min(m, r0 + win_rows)
            b = bufs[w % len(bufs)]
            if events[w % len(bufs)] is not None:
                events[w % len(bufs)].synchronize()           # the copy out of this buffer has finished
            view = b[: r1 - r0].numpy().reshape(-1)
            got = fh.readinto(memoryview(view))
            if got != (r1 - r0) * bps:
                raise RuntimeError(f"{path}: short read ({got} of {(r1 - r0) * bps} bytes)")
            out[r0:r1].copy_(b[: r1 - r0], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            events[w % len(bufs)] = ev
        stream.synchronize()
    return out, n, bim


def snps_only_mask(bim: Bim):
    """Both alleles single A/C/G/T (src/io/gfreader.rs:7013-7019)."""
    ok = set("ACGTacgt")
    return np.array([len(a) == 1 and len(b) == 1 and a in ok and b in ok for a, b in zip(bim.a0, bim.a1)], dtype=bool)


def write_bed(prefix, packed, sample_ids, bim: Bim):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    with open(f"{prefix}.bed", "wb") as fh:
        fh.write(BED_MAGIC)
        fh.write(packed.tobytes())
    with open(f"{prefix}.fam", "w") as fh:
        for s in sample_ids:
            fh.write(f"{s} {s} 0 0 0 -9\n")
    with open(f"{prefix}.bim", "w") as fh:
        for c, s, p, a0, a1 in zip(bim.chrom, bim.snp, bim.pos, bim.a0, bim.a1):
            fh.write(f"{c}\t{s}\t0\t{p}\t{a0}\t{a1}\n")


def pack_dosage(g: np.ndarray) -> np.ndarray:
    """(m, n) dosage in {0,1,2}, negative = missing -> (m, ceil(n/4)) uint8 payload."""
    g = np.asarray(g)
    m, n = g.shape
    codes = np.full((m, ((n + 3) // 4) * 4), 0, dtype=np.uint8)
    c = np.full(g.shape, 1, dtype=np.uint8)
    c[g == 0] = 0
    c[g == 1] = 2
    c[g == 2] = 3
    codes[:, :n] = c
    codes = codes.reshape(m, -1, 4)
    return (codes[:, :, 0] | (codes[:, :, 1] << 2) | (codes[:, :, 2] << 4) | (codes[:, :, 3] << 6)).astype(np.uint8)


def synth_panel_numpy(n, m, seed=20260609, missing_rate=0.0, maf_low=0.02, maf_high=0.45, family=False):
    """Synthetic HWE panel like the reference's simulator defaults (python/janusx/script/sim.py:15-19, 49-67):
    per-SNP MAF ~ U(maf_low, maf_high), g ~ Binomial(2, p) i.i.d.; optional sibships of 4 covering 80 % of the
    samples (each sib inherits one allele from each of two founder-like parents).  Returns (packed, dosage int8)."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(maf_low, maf_high, size=m)
    if not family:
        g = rng.binomial(2, p[:, None], size=(m, n)).astype(np.int8)
    else:
        g = rng.binomial(2, p[:, None], size=(m, n)).astype(np.int8)
        nfam = int(0.8 * n) // 4
        for f in range(nfam):
            pa = rng.binomial(2, p).astype(np.int8)
            ma = rng.binomial(2, p).astype(np.int8)
            for k in range(4):
                ta = (rng.random(m) < pa / 2.0).astype(np.int8)
                tb = (rng.random(m) < ma / 2.0).astype(np.int8)
                g[:, 4 * f + k] = ta + tb
    if missing_rate > 0:
        mask = rng.random((m, n)) < missing_rate
        g = g.copy()
        g[mask] = -9
    return pack_dosage(g), g


def synth_phenotype(g: np.ndarray, n_causal=100, pve=0.5, seed=20260609):
    """y = Z beta + e with `n_causal` causal SNPs and the requested PVE (scripts/benchmark.sh:32-37)."""
    rng = np.random.default_rng(seed + 1)
    m, n = g.shape
    idx = rng.choice(m, size=min(n_causal, m), replace=False)
    beta = rng.normal(size=idx.shape[0])
    z = np.where(g[idx] < 0, 0, g[idx]).astype(np.float64)
    z = z - z.mean(axis=1, keepdims=True)
    gv = beta @ z
    vg = float(np.var(gv))
    if vg <= 0:
        gv = np.zeros(n)
        vg = 1.0
    e = rng.normal(size=n) * np.sqrt(vg * (1.0 - pve) / pve)
    return gv + e
