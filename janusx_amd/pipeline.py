"""HBM-resident end-to-end mixed-model pipeline: GRM -> eigh -> null REML -> per-SNP scan.

This is the path `bench.py` times and the multi-GPU driver shards.  torch is used only as plumbing (device
buffers, streams, `torch.distributed`); every computation is a libjxgpu kernel called through the C ABI.

Mirrors the reference's call stack for `jx gwas -lmm/-fvlmm`:
  build_grm_streaming (python/janusx/assoc/workflow.py:2928) -> grm_stream_bed_f32 (src/stats/grm.rs:4690)
  _gwas_eigh_from_grm (workflow.py:5509, ridge 1e-6)      -> rust_eigh_* (src/math/eigh.rs:1422)
  LMM._initialize_from_spectral (python/janusx/pyBLUP/assoc.py:1726-1876)
  lmm_reml_assoc_bed_to_tsv_f32 / fvlmm_assoc_bed_to_tsv_f32 (src/stats/lmm.rs:2513, src/stats/fvlmm.rs:2502)
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import stats as st
from ._lib import check, lib

SCALE_EXP = 10  # U^T planes are scaled by 2^10 before the fp16 split (keeps the lo plane out of subnormals)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


@dataclass
class NullFit:
    lbd: float
    ml0: float
    reml0: float
    sigma_g2: float
    sigma_e2: float
    pve: float
    bounds: tuple


@dataclass
class StageTimes:
    t: dict = field(default_factory=dict)

    def add(self, key, dt):
        self.t[key] = self.t.get(key, 0.0) + dt


class Panel:
    """A 2-bit genotype payload resident in HBM in the internal P32 tiling."""

    def __init__(self, packed: torch.Tensor, n_samples: int, sample_idx=None, p32_buffer=None):
        # p32_buffer: a (tiles, m, 32) uint8 device tensor of a previous Panel of the same shape to re-tile into (a caller that
        # re-tiles the same payload repeatedly -- bench.py's steps -- keeps the allocation out of its loop)
        assert packed.is_cuda and packed.dtype == torch.uint8 and packed.dim() == 2
        self.device = packed.device
        self.m = int(packed.shape[0])
        self.n_src = int(n_samples)
        bps = int(packed.shape[1])
        if bps != (self.n_src + 3) // 4:
            raise RuntimeError(f"packed length mismatch: got {bps} bytes per SNP, expected {(self.n_src + 3) // 4}")
        idx_t = None
        if sample_idx is not None:
            idx = np.asarray(sample_idx, dtype=np.int64)
            if idx.size == 0:
                raise RuntimeError("sample_indices must not be empty")
            if idx.min() < 0 or idx.max() >= self.n_src:
                raise RuntimeError(f"sample index out of range: {int(idx.max())} >= {self.n_src}")
            if not (idx.size == self.n_src and np.array_equal(idx, np.arange(self.n_src))):
                idx_t = torch.from_numpy(idx.astype(np.int32)).to(self.device)
            self.n = int(idx.size)
        else:
            self.n = self.n_src
        self.nt = lib().jxg_num_tiles(self.n)
        self.npad = self.nt * 128
        if (p32_buffer is not None and tuple(p32_buffer.shape) == (self.nt, self.m, 32) and p32_buffer.dtype == torch.uint8
                and p32_buffer.device == self.device and p32_buffer.is_contiguous()):
            self.p32 = p32_buffer
        else:
            self.p32 = torch.empty((self.nt, self.m, 32), dtype=torch.uint8, device=self.device)
        check(lib().jxg_repack_p32(_ptr(packed), bps, self.n_src, self.m, _ptr(idx_t), self.n, None, self.m,
                                   _ptr(self.p32), _stream()))
        self._counts = None
        self._mean_missing = None
        self.sharded = False            # True: this payload is one rank's SNP shard of a larger panel (`payload_sharded`)

    def mean_missing(self) -> float:
        """Mean number of missing calls per row over the rows of the payload that CAN pass a quality filter (at most n / 10
        missing calls: a few percent of mostly-missing junk rows must not push every kept row into the dense missing-call form;
        ADVICE r4), cached: the call-independent statistic behind the choice of the rotation path for rows with a few missing
        calls (`jxg_rot_miss_max`; api.cpp applies the same rule to the rows it is handed).  Under a distributed run the sum and
        the row count are added over the ranks, so that every rank of a payload-sharded scan decides from the same number (with
        a replicated payload every rank already holds the same rows)."""
        if self._mean_missing is None:
            c = self.counts()
            mi = c[:, 0].astype(np.float64) if len(c) else np.zeros(0)
            ok = mi <= self.n / 10.0
            tot = np.array([float(mi[ok].sum()), float(np.count_nonzero(ok))], dtype=np.float64)
            if self.sharded:
                from . import dist as jd
                rank, world = dist_info()
                if world > 1:
                    t = torch.from_numpy(tot)
                    if torch.distributed.get_backend() == "nccl":
                        t = t.to(self.device)
                    jd.allreduce_sum_(t)
                    tot = t.cpu().numpy()
            self._mean_missing = float(tot[0] / tot[1]) if tot[1] > 0 else 0.0
        return self._mean_missing

    def counts(self) -> np.ndarray:
        """(m,3) int32 (missing, het, hom_alt) over the selected samples (host copy, cached)."""
        if self._counts is None:
            c = torch.empty((self.m, 3), dtype=torch.int32, device=self.device)
            check(lib().jxg_row_counts_p32(_ptr(self.p32), self.m, self.n, _ptr(c), _stream()))
            self._counts = c.cpu().numpy()
        return self._counts


def grm_accumulate(panel: Panel, rows: np.ndarray, lut: np.ndarray, acc: torch.Tensor = None, kchunk=0):
    """acc (npad,npad) f64 += Z Z^T over the given SNP rows with their (len(rows),4) f32 value LUT."""
    dev = panel.device
    if acc is None:
        acc = torch.zeros((panel.npad, panel.npad), dtype=torch.float64, device=dev)
    if len(rows) == 0:
        return acc
    rows_t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(dev)
    lut_t = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float32)).to(dev)
    check(lib().jxg_grm_accumulate(_ptr(panel.p32), panel.m, panel.n, _ptr(rows_t), _ptr(lut_t), len(rows),
                                   _ptr(acc), int(kchunk), 0, _stream()))
    return acc


def grm_finalize(acc: torch.Tensor, n: int, scale: float, dtype=torch.float32):
    out = torch.empty((n, n), dtype=dtype, device=acc.device)
    check(lib().jxg_grm_finalize(_ptr(acc), n, 1.0 / float(scale), _ptr(out), 1 if dtype == torch.float64 else 0,
                                 _stream()))
    return out


_DIST_EIGH = {}   # keeps the ctypes callback and the staging tensor alive


def dist_info():
    """(rank, world) of the initialised torch.distributed group; (0, 1) without one (or with a single rank)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_distributed(eigh_min_n: int = 16384):
    """Call once per process after `torch.distributed.init_process_group` (one process per GPU; backend "nccl" = RCCL over
    xGMI, "gloo" in the functional tests): switches the rank-aware forms of `build_grm` / `run_trait` / `run_gwas` on (they
    look at the process group themselves) and distributes the eigendecomposition (`enable_distributed_eigh`).
    SURVEY.md 8(e): SNP ranges per rank for the GRM, one sum-reduction of the partial accumulators, eigenvectors sharded over
    the ranks, SNP-sharded scan, results gathered in BED order.  -> (rank, world)."""
    rank, world = dist_info()
    if world > 1 and os.environ.get("JXGPU_DIST_EIGH", "1") != "0":
        enable_distributed_eigh(eigh_min_n)
    return rank, world


def _my_slice(count: int, payload_sharded: bool):
    """This rank's contiguous share [lo, hi) of `count` work items (all of them when the payload itself is the rank's shard)."""
    rank, world = dist_info()
    if world == 1 or payload_sharded:
        return 0, count
    from .dist import shard_range
    return shard_range(count, rank, world)


def _allgather_rows(t: torch.Tensor) -> torch.Tensor:
    """Per-rank (rows_r, ...) tensors concatenated in rank order on every rank (= BED order of contiguous SNP shards)."""
    from .dist import gather_rows
    import torch.distributed as dist
    if dist_info()[1] == 1:
        return t
    if t.is_cuda and dist.get_backend() != "nccl":
        return gather_rows(t.cpu()).to(t.device)         # functional multi-rank mode on shared GPUs (gloo): through host memory
    return gather_rows(t)


gather_results = _allgather_rows


def _allgather_np(a: np.ndarray) -> np.ndarray:
    if dist_info()[1] == 1:
        return a
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dist.get_backend() == "nccl":
        return _allgather_rows(t.to(torch.device("cuda", torch.cuda.current_device()))).cpu().numpy()
    from .dist import gather_rows
    return gather_rows(t).numpy()


def enable_distributed_eigh(min_n: int = 0):
    """Distribute the eigendecomposition over the ranks of the initialised torch.distributed group (every rank must then
    call `eigh_from_grm` on the same matrix).  Two-stage path (n >= 10000, the default there): the reduction stages and
    the divide and conquer run replicated -- their kernels are bit-reproducible, so the replicas agree --, every rank
    back-transforms its n / world eigenvectors (two thirds of the flops) and one broadcast per rank (RCCL over xGMI with
    the nccl backend) completes U on every rank.  One-stage path (JXGPU_DIST_EIGH_ONESTAGE=1, or n below the two-stage
    threshold): per column each rank streams 1 / world of the trailing-matrix tiles of the symv and one all-reduce sums
    the partial products, from `min_n` rows (default 16384: below that a column is launch-latency-bound and the
    collective costs more than it saves)."""
    import ctypes as C
    import torch.distributed as dist
    from .dist import allreduce_sum_, broadcast_
    import os
    force_single = os.environ.get("JXGPU_DIST_EIGH_FORCE", "0") not in ("", "0")   # one-rank run of the same code path
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_single):
        check(lib().jxg_eigh_set_dist(0, 1, None, None, None, 0, 0))
        check(lib().jxg_eigh_set_gather(None, None))
        check(lib().jxg_eigh_set_agree(None, None))
        check(lib().jxg_eigh_set_band_dist(0, 1, None, None, None, 0, 0, 0))
        _DIST_EIGH.clear()
        return False
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    state = {"staging": None}

    def _cb(_user):
        try:
            allreduce_sum_(state["staging"])
            return 0
        except Exception as e:   # noqa: BLE001 - reported through the C status
            import sys
            print(f"distributed eigh: all-reduce failed on rank {rank}: {e!r}", file=sys.stderr, flush=True)
            return 1

    cb = C.CFUNCTYPE(C.c_int, C.c_void_p)(_cb)

    def _gather(_user):
        # two-stage path: every rank has back-transformed the eigenvectors [n r / world, n (r + 1) / world) (rows of the
        # row-major result); one broadcast per rank completes the matrix on every rank (RCCL over xGMI with nccl)
        try:
            a = state["out"]
            n = int(a.shape[0])
            for r in range(world):
                r0, r1 = (n * r) // world, (n * (r + 1)) // world
                if r1 > r0:
                    broadcast_(a[r0:r1], r)
            return 0
        except Exception as e:   # noqa: BLE001 - reported through the C status
            import sys
            print(f"distributed eigh: gather failed on rank {rank}: {e!r}", file=sys.stderr, flush=True)
            return 1

    gcb = C.CFUNCTYPE(C.c_int, C.c_void_p)(_gather)
    check(lib().jxg_eigh_set_gather(C.cast(gcb, C.c_void_p) if world > 1 else None, None))

    def _agree(_user, checksum):
        # the replicated stages must have produced the same bits on every rank before rows of different ranks are mixed:
        # MIN and MAX of the 64-bit checksum (as two 32-bit halves in int64 lanes) over the ranks
        try:
            cs = int(checksum) & 0xFFFFFFFFFFFFFFFF
            # test hook: this rank pretends to hold different bits (the replicated results' checksum only -- the failure flags of
            # the reduction stages, compared through the same callback, are 0 or the constant below)
            if os.environ.get("JXGPU_DIST_EIGH_TEST_DISAGREE", "") == str(rank) and cs not in (0, 0x9E3779B97F4A7C15):
                cs ^= 1
            v = torch.tensor([cs >> 32, cs & 0xFFFFFFFF], dtype=torch.int64)
            lo, hi = v.clone(), v.clone()
            if dist.get_backend() == "nccl":
                lo, hi = lo.to(dev), hi.to(dev)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            return 1 if bool((lo == hi).all()) else 0
        except Exception as e:   # noqa: BLE001 - reported through the C status
            import sys
            print(f"distributed eigh: agreement check failed on rank {rank}: {e!r}", file=sys.stderr, flush=True)
            return -1

    acb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)(_agree)
    check(lib().jxg_eigh_set_agree(C.cast(acb, C.c_void_p) if world > 1 else None, None))

    def _band(_user, count):
        # band reduction with the trailing matrix sharded over the ranks (k_sy2sb.hip): per panel the partial products Z = A22 V
        # and the gather of the next panel's block column are summed over the ranks (RCCL over xGMI with nccl, on the stream
        # the eigensolver runs on)
        try:
            t = state["band_staging"][: int(count)]
            if world > 1:
                allreduce_sum_(t)
            elif dist.get_backend() == "nccl":
                dist.all_reduce(t)      # one-rank run of the same code path (JXGPU_DIST_EIGH_FORCE): the RCCL call itself
            return 0
        except Exception as e:   # noqa: BLE001 - reported through the C status
            import sys
            print(f"distributed eigh: band all-reduce failed on rank {rank}: {e!r}", file=sys.stderr, flush=True)
            return 1

    bcb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64)(_band)
    band_on = os.environ.get("JXGPU_DIST_BAND", "1") != "0"
    band_min_n = int(os.environ.get("JXGPU_DIST_BAND_MIN_N", "8192"))
    band_block = int(os.environ.get("JXGPU_DIST_BAND_BLOCK", "2048"))

    def prepare(n):
        need = int(lib().jxg_eigh_dist_staging_doubles(int(n)))
        if state["staging"] is None or state["staging"].numel() != need:
            state["staging"] = torch.zeros(need, dtype=torch.float64, device=dev)
            check(lib().jxg_eigh_set_dist(rank, world, C.cast(cb, C.c_void_p), None, _ptr(state["staging"]), need,
                                          int(min_n)))
        if band_on and int(n) >= band_min_n:
            # block size first (it sizes the staging buffer), then the buffer itself
            check(lib().jxg_eigh_set_band_dist(rank, world, None, None, None, 0, band_min_n, band_block))
            bneed = int(lib().jxg_eigh_band_staging_doubles(int(n)))
            if state.get("band_staging") is None or state["band_staging"].numel() < bneed:
                state["band_staging"] = torch.zeros(bneed, dtype=torch.float64, device=dev)
            check(lib().jxg_eigh_set_band_dist(rank, world, C.cast(bcb, C.c_void_p), None, _ptr(state["band_staging"]),
                                               int(state["band_staging"].numel()), band_min_n, band_block))
        else:
            check(lib().jxg_eigh_set_band_dist(rank, world, None, None, None, 0, band_min_n, band_block))

    _DIST_EIGH.update(cb=cb, gcb=gcb, acb=acb, bcb=bcb, state=state, prepare=prepare)
    return True


LAST_EIGH = {"planes": 0}      # digit planes of the sliced int8 products in the last eigh_from_grm call


def eigh_from_grm(k: torch.Tensor, ridge=1e-6, subset_idx=None, f32_consumer=False):
    """K (n,n) f32/f64 on device -> (S f64 (k), U^T f64 (k,k) row j = eigenvector j).
    f32_consumer: the caller keeps the eigenvectors only as the f32 U^T of the scan (`SpectralModel`, the reference's `Dh`,
    python/janusx/pyBLUP/assoc.py:1817; src/stats/reml.rs:109-198): the sliced int8 products of the Q1 back-transformation and of
    the divide-and-conquer merges then run with 5 digit planes (15 products) instead of 6 (21) -- orthogonality of U 4e-10
    instead of 2e-12 at n = 20 000, six times below the 2.6e-9 the f32 rounding of the kept copy leaves by itself; eigenvalues unchanged
    (JXGPU_OZ_PLANES, when set, decides alone; JXGPU_EIGH_F32_PLANES=0 keeps 6)."""
    dev = k.device
    n = int(k.shape[0])
    idx = np.arange(n, dtype=np.int32) if subset_idx is None else np.asarray(subset_idx, dtype=np.int32)
    kk = int(idx.shape[0])
    idx_t = torch.from_numpy(idx).to(dev)
    a = torch.empty((kk, kk), dtype=torch.float64, device=dev)
    check(lib().jxg_gather_sub_f64(_ptr(k), 1 if k.dtype == torch.float64 else 0, n, _ptr(idx_t), kk, _ptr(a),
                                   _stream()))
    check(lib().jxg_symmetrize_f64(_ptr(a), kk, _stream()))
    w = torch.empty(kk, dtype=torch.float64, device=dev)
    if _DIST_EIGH:
        _DIST_EIGH["prepare"](kk)   # staging buffer of the per-column all-reduce (sized by n)
        _DIST_EIGH["state"]["out"] = a   # the gather callback of the two-stage path completes this tensor
    five = bool(f32_consumer) and not os.environ.get("JXGPU_OZ_PLANES") and os.environ.get("JXGPU_EIGH_F32_PLANES", "5") == "5"
    prev = lib().jxg_oz_set_planes(5) if five else 0
    try:
        LAST_EIGH["planes"] = int(lib().jxg_oz_planes())      # what bench.py prices the Q1 stage on
        check(lib().jxg_eigh_f64(_ptr(a), kk, float(ridge), _ptr(w), _stream()))
    finally:
        if five:
            lib().jxg_oz_set_planes(prev)
    return w, a


class SpectralModel:
    """Device-resident counterpart of `LMM.from_spectral` (python/janusx/pyBLUP/assoc.py:1702-1876)."""

    def __init__(self, s: torch.Tensor, ut64: torch.Tensor, x: np.ndarray, y: np.ndarray, fit_null: bool = True):
        dev = s.device
        self.n = n = int(s.shape[0])
        self.S = s
        self.ut = torch.empty((n, n), dtype=torch.float32, device=dev)  # Dh = float32(U^T)
        check(lib().jxg_cast_f64_to_f32(_ptr(ut64), _ptr(self.ut), n * n, _stream()))
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64).ravel()
        if x.ndim != 2 or x.shape[0] != n or y.shape[0] != n:
            raise RuntimeError(f"design row mismatch: got {x.shape[0]}, expected {n}")
        self.p = p = int(x.shape[1])
        xy = torch.from_numpy(np.concatenate([x, y[:, None]], axis=1)).to(dev)
        rot = torch.empty_like(xy)
        check(lib().jxg_rotate_xy(_ptr(self.ut), n, _ptr(xy), p + 1, _ptr(rot), _stream()))
        self.xcov = rot[:, :p].contiguous()
        self.y = rot[:, p].contiguous()
        self._planes = None
        self._fv = None
        if not fit_null:      # spectrum of a sparse (possibly indefinite) K: lambda comes from the sparse REML search
            self.null = None
            return
        out3 = torch.empty(3, dtype=torch.float64, device=dev)
        check(lib().jxg_lmm_reml_null(_ptr(self.S), _ptr(self.xcov), _ptr(self.y), n, p, -5.0, 5.0, 50, 1e-3,
                                      _ptr(out3), _stream()))
        lbd, ml0, reml0 = [float(v) for v in out3.cpu().numpy()]
        self.null = self._profile(lbd, ml0, reml0)
        self._planes = None
        self._fv = None

    def _profile(self, lbd, ml0, reml0) -> NullFit:
        # assoc.py:907-951 `_lmm_profile_exact_vc` + :1845-1876 (host, O(n p^2))
        s = np.maximum(self.S.cpu().numpy(), 0.0)
        x = self.xcov.cpu().numpy()
        y = self.y.cpu().numpy()
        n, p = x.shape
        sg2 = se2 = float("nan")
        if math.isfinite(lbd) and lbd > 0.0 and n - p > 0:
            v_inv = 1.0 / np.maximum(s + lbd, 1e-30)
            a = (x.T * v_inv) @ x
            b = (x.T * v_inv) @ y
            try:
                beta = np.linalg.solve(a, b)
            except np.linalg.LinAlgError:
                beta = np.linalg.lstsq(a, b, rcond=None)[0]
            r = y - x @ beta
            q = float(np.dot(v_inv, r * r))
            if math.isfinite(q) and q > 0.0:
                sg2 = q / float(n - p)
                se2 = lbd * sg2
        trace_mean = float(np.sum(s) / float(max(1, n)))
        ssum = sg2 + se2
        if math.isfinite(ssum) and ssum > 0.0:
            vg = sg2 * max(trace_mean, 0.0)
            den = vg + se2
            pve = vg / den if (math.isfinite(den) and den > 0.0) else sg2 / ssum
        else:
            vg0 = float(np.mean(s))
            pve = vg0 / (vg0 + lbd) if (vg0 + lbd) > 0 else float("nan")
        if pve > 0.95 or pve < 0.05 or (not math.isfinite(lbd)) or lbd <= 0.0:
            bounds = (-5.0, 5.0)
        else:
            bounds = (math.log10(lbd) - 2.0, math.log10(lbd) + 2.0)
        return NullFit(lbd, ml0, reml0, sg2, se2, pve, bounds)

    def planes(self):
        if self._planes is None:
            nt = lib().jxg_num_tiles(self.n)
            npad = nt * 128
            hi = torch.empty((npad, npad), dtype=torch.float16, device=self.S.device)
            lo = torch.empty((npad, npad), dtype=torch.float16, device=self.S.device)
            check(lib().jxg_ut_split(_ptr(self.ut), self.n, _ptr(hi), _ptr(lo), SCALE_EXP, _stream()))
            usum = torch.empty(npad, dtype=torch.float32, device=self.S.device)   # sum_i u_t[j][i]: affine term of exact rows
            check(lib().jxg_ut_rowsum(_ptr(self.ut), self.n, _ptr(usum), _stream()))
            self._planes = (hi, lo, usum)
        return self._planes

    def qplanes(self):
        """Three int8 planes of U^T with one scale per eigenvector (jxg_ut_quant3): operands of the int8 rotation of exact
        design rows (csrc/k_rotate_i8.hip)."""
        if getattr(self, "_qplanes", None) is None:
            npad = lib().jxg_num_tiles(self.n) * 128
            q = torch.empty((3, npad, npad), dtype=torch.int8, device=self.S.device)
            umax = torch.empty(npad, dtype=torch.float32, device=self.S.device)
            check(lib().jxg_ut_quant3(_ptr(self.ut), self.n, _ptr(q), _ptr(umax), _stream()))
            self._qplanes = (q, umax)
        return self._qplanes

    def usamp(self):
        """U with one row per sample (the transpose of `ut`, f32): what the missing-call term of the rotation gathers from
        (jxg_rotate_missing_correct)."""
        if getattr(self, "_usamp", None) is None:
            us = torch.empty((self.n, self.n), dtype=torch.float32, device=self.S.device)
            check(lib().jxg_transpose_f32(_ptr(self.ut), self.n, _ptr(us), _stream()))
            self._usamp = us
        return self._usamp

    def fv_cache(self, log10_lbd=None):
        lbd = self.null.lbd if log10_lbd is None else 10.0 ** float(log10_lbd)
        if self._fv is None or self._fv[0] != lbd:
            dev = self.S.device
            n, p = self.n, self.p
            w = torch.empty(n, dtype=torch.float32, device=dev)
            py = torch.empty(n, dtype=torch.float32, device=dev)
            wx = torch.empty((n, p), dtype=torch.float32, device=dev)
            a = np.zeros((p, p), dtype=np.float64)
            sc = np.zeros(3, dtype=np.float64)
            torch.cuda.current_stream().synchronize()
            check(lib().jxg_fvlmm_prepare(_ptr(self.S), _ptr(self.xcov), _ptr(self.y), n, p, float(lbd), _ptr(w),
                                          _ptr(py), _ptr(wx), a.ctypes.data, sc.ctypes.data))
            self._fv = (lbd, w, py, wx, a, float(sc[0]), float(sc[1]), int(sc[2]))
        return self._fv


def _fused_fixed_lambda(p: int) -> bool:
    """The fixed-lambda scans (fvlmm, SparseLMM exact) reduce the rotated tile inside the rotation kernel's epilogue
    (`jxg_rotate_packed16x_fused` + `jxg_fvlmm_finish_dev`) instead of writing and re-reading G~ (4 m n bytes each way);
    JXGPU_FVLMM_FUSED=0 keeps the two-kernel form, more than 8 covariates always do."""
    return p <= 8 and os.environ.get("JXGPU_FVLMM_FUSED", "1").strip() != "0"


def scan_rows(panel: Panel, model: SpectralModel, rows: np.ndarray, lut: np.ndarray, mode="lmm", low=None,
              high=None, max_iter=30, tol=1e-2, init_log10_lbd=None, block_rows=8192, return_evals=False,
              times: StageTimes = None, nullml=None, fv_state=None, on_block=None, chain_off=None):
    """Rotate + scan the given SNP rows. mode: 'lmm' (exact per-SNP REML), 'fvlmm' (fixed lambda) or 'lmm2' (REML Wald +
    ML likelihood ratio, needs `nullml`).  Returns a (len(rows), 3) f64 device tensor [beta, se, p] -- 4 columns
    [.., plrt] when `nullml` is given, 6 columns [beta, se, pwald, lambda, ml, plrt] for 'lmm2' -- (and the per-SNP
    Brent evaluation counts).
    chain_off ('lmm' only): offsets into `rows` (int64, 0 ... len(rows), ascending; `stats.warm_chain_offsets`) of the
    reference's warm-start chains (src/stats/lmm.rs:134-161): inside a chain every SNP's Brent starts from the optimum of the
    valid SNP before it, the first one from `init_log10_lbd` or the interval midpoint.  The rotation and the per-SNP series stay
    parallel over all SNPs; the Brent searches run one wave per chain -- in ONE launch behind the last block where the series
    form exists (`on_block` then receives the whole table at the end), block by block with carried states otherwise."""
    dev = panel.device
    n = model.n
    if panel.n != n:
        raise RuntimeError(f"selected sample count {panel.n} != model n {n}")
    mk = len(rows)
    with_plrt = 1 if nullml is not None else 0
    nullml_v = float(nullml) if nullml is not None else 0.0
    if mode == "lmm2" and nullml is None:
        raise RuntimeError("lmm2 needs the null ML (nullml)")
    out = torch.empty((mk, 6 if mode == "lmm2" else (4 if with_plrt else 3)), dtype=torch.float64, device=dev)
    evals = torch.zeros(mk, dtype=torch.int32, device=dev) if return_evals else None
    if mk == 0:
        return (out, evals) if return_evals else out
    rows_t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(dev)
    lut_t = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float32)).to(dev)
    hi, lo, usum = model.planes()
    # int8 planes of the eigenvectors for the exact-row tiles of full-size blocks (k_rotate_i8.hip)
    # From n = 4096 the int8 / 256-tile rotation kernels write G~ and the fixed-lambda scans read it back (10 ms per 200 000
    # SNPs at n = 20 000) instead of the fused epilogue of the 128-tile fp16 kernel: 225 vs 360 ms for the scan of `-fvlmm` at
    # BASELINE configs[2].  JXGPU_FVLMM_FUSED=2 keeps the fused form at every size.
    use_q = n >= 4096 and os.environ.get("JXGPU_ROT_I8", "1") != "0"
    if mode in ("fvlmm", "splmm") and os.environ.get("JXGPU_FVLMM_FUSED", "1").strip() == "2" and _fused_fixed_lambda(model.p):
        use_q = False
    qpl = model.qplanes() if use_q else None
    # one-off per call: fp16 hi/lo LUT records (range-checked; rows without missing calls as integer LUT + offset, see
    # jxg_lut_split_rows) and, for the exact scan, the Chebyshev tables of the lambda-only REML sums; the block loop
    # below then only launches kernels (no allocation, no host sync).
    lut16 = torch.empty((mk, 16), dtype=torch.uint8, device=dev)
    rowoff = torch.empty(mk, dtype=torch.float32, device=dev)
    # rows with a few missing calls keep the exact (int8) rotation and get their missing-call term added behind it
    # (jxg_rotate_missing_correct: d * the sum of the missing samples' rows of U); only where the int8 rotation runs
    # The limit is decided from a statistic of the PANEL (mean missing calls over all its rows, cached), never from the rows of
    # this call: a row takes the same path -- and gets the same bits -- in chunked, unchunked and rank-sharded scans, and the
    # host entry point (api.cpp, all rows of the payload) decides the same way.
    miss_max = 0
    if qpl is not None:
        miss_max = int(lib().jxg_rot_miss_max(n, panel.mean_missing()))
    rowmiss = torch.zeros(mk, dtype=torch.float32, device=dev) if miss_max > 0 else None
    check(lib().jxg_lut_split_rows_m(_ptr(panel.p32), panel.m, n, _ptr(rows_t), _ptr(lut_t), mk, _ptr(lut16),
                                     _ptr(rowoff), _ptr(rowmiss) if rowmiss is not None else None, miss_max, _stream()))
    if rowmiss is not None and not bool((rowmiss != 0).any().item()):
        rowmiss = None
    # the missing-call term as one more int8 product (jxg_rotate_missing_dense) when the limit is "none" (> 256: more than n / 300
    # missing calls per row on average), else as a gather per missing call over U with one row per sample
    miss_dense = rowmiss is not None and miss_max > 256
    usamp = model.usamp() if (rowmiss is not None and not miss_dense) else None
    sel_miss_t, sel_miss_bounds = None, []
    tables = None
    if mode == "splmm" and fv_state is None:
        raise RuntimeError("the SparseLMM exact scan needs its null state (fv_state = w, py, wx, a_chol, ypy)")
    if mode == "lmm2":
        lo_b, hi_b = model.null.bounds if low is None else (float(low), float(high))
        warm = 1 if init_log10_lbd is not None else 0
        init = float(init_log10_lbd) if init_log10_lbd is not None else 0.0
    elif mode == "lmm":
        lo_b, hi_b = model.null.bounds if low is None else (float(low), float(high))
        warm = 1 if init_log10_lbd is not None else 0
        init = float(init_log10_lbd) if init_log10_lbd is not None else 0.0
        nbytes = int(lib().jxg_lmm_tables_bytes(n, model.p, lo_b, hi_b))
        if nbytes > 0:
            tables = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            check(lib().jxg_lmm_tables_build(_ptr(model.S), _ptr(model.xcov), _ptr(model.y), n, model.p, lo_b, hi_b,
                                             _ptr(tables), _stream()))
        if chain_off is not None:
            co = np.ascontiguousarray(chain_off, dtype=np.int64)
            if co.ndim != 1 or len(co) < 2 or co[0] != 0 or co[-1] != mk or np.any(np.diff(co) < 0):
                raise RuntimeError("chain_off must ascend from 0 to len(rows)")
            nch = len(co) - 1
            carry = torch.full((nch,), init if warm else float("nan"), dtype=torch.float64, device=dev)
            sd = int(lib().jxg_lmm_series_doubles(model.p, lo_b, hi_b)) if tables is not None else 0
            if sd > 0:
                scoef = torch.empty((mk, sd), dtype=torch.float64, device=dev)
                sssq = torch.empty(mk, dtype=torch.float64, device=dev)
                co_t = torch.from_numpy(co.astype(np.int32)).to(dev)
    elif mode == "splmm":
        # null state on the K + lambda I scale, formed in f64 by the caller without the 1e-6 ridge jxg_fvlmm_prepare puts
        # on X'WX (fvlmm.rs): at the large lambda of a trait without polygenic signal that ridge is a 1e-4 relative error
        w, py, wx, a_chol, ypy = fv_state
        a_dev = torch.from_numpy(np.ascontiguousarray(a_chol, dtype=np.float64)).to(dev)
    else:
        lbd, w, py, wx, a_chol, ypy, log_det_v, df = model.fv_cache(init_log10_lbd)
        a_dev = torch.from_numpy(a_chol).to(dev)
    if mode == "lmm" and block_rows == 8192 and (8 * n * (2 + model.p) > 156 * 1024 or model.p + 1 > 4):
        # beyond the LDS-resident limit the exact scan runs its tiled form: one workgroup per CU walks a queue of SNPs in
        # lock step over LDS tiles of (s, X~, y~); longer blocks keep every wave's queue deep (8 SNPs per wave)
        block_rows = 32768
    br = int(min(block_rows, mk))
    if miss_dense:
        # per block: positions of the rows with a missing-call term (a row's path does not depend on the blocking)
        has = (rowmiss != 0).cpu().numpy()
        parts, nm_ = [], 0
        for b0 in range(0, mk, br):
            a = np.flatnonzero(has[b0:b0 + br]).astype(np.int32)
            sel_miss_bounds.append((nm_, nm_ + len(a)))
            nm_ += len(a)
            parts.append(a)
        sel_miss_t = torch.from_numpy(np.concatenate(parts) if nm_ else np.zeros(1, np.int32)).to(dev)
    chain_series = mode == "lmm" and chain_off is not None and sd > 0
    nbuf = 2 if mk > br else 1
    fused = mode in ("fvlmm", "splmm") and _fused_fixed_lambda(model.p) and qpl is None
    if qpl is not None:
        # per block: positions of the exact rows (finite row offset: int8 rotation) and of the others (fp16 rotation); a row's
        # path does not depend on the blocking, so chunked scans stay bit-identical to unchunked ones
        n_inexact = int(torch.isnan(rowoff).sum().item())
        se, sr, sel_bounds = [], [], []
        ne = nx = 0
        if n_inexact == 0:
            # every row exact (a panel without missing calls): identity lists, nothing to build
            for b0 in range(0, mk, br):
                sel_bounds.append((0, min(br, mk - b0), 0, 0))
            sel_exact_t = sel_rest_t = None
        else:
            ex = ~np.isnan(rowoff.cpu().numpy())
            for b0 in range(0, mk, br):
                blk = ex[b0:b0 + br]
                a, b = np.flatnonzero(blk).astype(np.int32), np.flatnonzero(~blk).astype(np.int32)
                sel_bounds.append((ne, ne + len(a), nx, nx + len(b)))
                ne, nx = ne + len(a), nx + len(b)
                se.append(a)
                sr.append(b)
            sel_exact_t = torch.from_numpy(np.concatenate(se) if ne else np.zeros(1, np.int32)).to(dev)
            sel_rest_t = torch.from_numpy(np.concatenate(sr) if nx else np.zeros(1, np.int32)).to(dev)
    if fused:
        sums = [torch.empty((panel.nt, br, model.p + 2), dtype=torch.float64, device=dev) for _ in range(nbuf)]
        grots = [None] * nbuf
    else:
        grots = [torch.empty((br, n), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    ev_rot = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range((mk + br - 1) // br)]
    ev_scan = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(len(ev_rot))]
    pending = None          # (i0, i1, event recorded behind the block's scan)
    copy_stream = torch.cuda.Stream(device=dev) if on_block is not None else None

    def hand_over(item):
        i0, i1, ev = item
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ev)
            host = out[i0:i1].to("cpu", non_blocking=False)
        on_block(i0, host.numpy())

    for bi, r0 in enumerate(range(0, mk, br)):
        nr = min(br, mk - r0)
        grot = grots[bi % nbuf]
        if times is not None:
            ev_rot[bi][0].record()
        if fused:
            sm = sums[bi % nbuf]
            check(lib().jxg_rotate_packed16x_fused(_ptr(panel.p32), panel.m, n, rows_t[r0:].data_ptr(), nr,
                                                   lut16[r0:].data_ptr(), rowoff[r0:].data_ptr(), _ptr(usum), _ptr(hi),
                                                   _ptr(lo), SCALE_EXP, _ptr(w), _ptr(py), _ptr(wx), model.p, _ptr(sm),
                                                   model.p + 2, 0, _stream()))
        else:
            if qpl is not None:
                e0, e1, x0, x1 = sel_bounds[bi]
                check(lib().jxg_rotate_packed16x_q(_ptr(panel.p32), panel.m, n, rows_t[r0:].data_ptr(), nr,
                                                   lut16[r0:].data_ptr(), rowoff[r0:].data_ptr(), _ptr(usum), _ptr(hi), _ptr(lo),
                                                   SCALE_EXP, _ptr(qpl[0]), _ptr(qpl[1]),
                                                   sel_exact_t[e0:].data_ptr() if (sel_exact_t is not None and e1 > e0) else None,
                                                   e1 - e0,
                                                   sel_rest_t[x0:].data_ptr() if (sel_rest_t is not None and x1 > x0) else None,
                                                   x1 - x0, _ptr(grot), _stream()))
                if miss_dense:
                    m0, m1 = sel_miss_bounds[bi]
                    if m1 > m0:
                        check(lib().jxg_rotate_missing_dense(_ptr(panel.p32), panel.m, n, rows_t[r0:].data_ptr(),
                                                             sel_miss_t[m0:].data_ptr(), m1 - m0, rowmiss[r0:].data_ptr(),
                                                             _ptr(qpl[0]), _ptr(qpl[1]), _ptr(grot), n, _stream()))
                elif rowmiss is not None:
                    check(lib().jxg_rotate_missing_correct(_ptr(panel.p32), panel.m, n, rows_t[r0:].data_ptr(), nr,
                                                           rowmiss[r0:].data_ptr(), _ptr(usamp), _ptr(grot), n, _stream()))
            else:
                check(lib().jxg_rotate_packed16x(_ptr(panel.p32), panel.m, n, rows_t[r0:].data_ptr(), nr,
                                                 lut16[r0:].data_ptr(), rowoff[r0:].data_ptr(), _ptr(usum), _ptr(hi), _ptr(lo),
                                                 SCALE_EXP, _ptr(grot), _stream()))
        if times is not None:
            ev_rot[bi][1].record()
            ev_scan[bi][0].record()
        o = out[r0:]
        if fused:
            if mode == "splmm":
                check(lib().jxg_fvlmm_finish_dev(_ptr(sm), panel.nt, model.p + 2, nr, n, model.p, _ptr(a_dev), ypy, n - model.p, 0,
                                                 0.0, 0.0, 1, o.data_ptr(), _stream()))
            else:
                check(lib().jxg_fvlmm_finish_dev(_ptr(sm), panel.nt, model.p + 2, nr, n, model.p, _ptr(a_dev), ypy, df, with_plrt,
                                                 nullml_v, log_det_v, 0, o.data_ptr(), _stream()))
        elif mode == "lmm2":
            check(lib().jxg_lmm2_scan(_ptr(grot), nr, n, _ptr(model.S), _ptr(model.xcov), _ptr(model.y), model.p, lo_b,
                                      hi_b, float(tol), int(max_iter), warm, init, nullml_v, o.data_ptr(), _stream()))
        elif mode == "lmm" and chain_off is not None:
            ev_p = evals[r0:].data_ptr() if evals is not None else None
            if sd > 0:
                # the series of this block; the Brent searches follow behind the last block, all chains in one launch
                check(lib().jxg_lmm_series_coef_tab(_ptr(grot), nr, n, _ptr(model.xcov), model.p, lo_b, hi_b, _ptr(tables),
                                                    scoef[r0:].data_ptr(), sssq[r0:].data_ptr(), _stream()))
            else:
                # no series form (wide bounds, many covariates): the chains that touch this block, states carried across blocks
                c0 = int(np.searchsorted(co, r0, side="right")) - 1
                c0 = min(c0, nch - 1)
                c1 = c0
                loc = []
                while c1 < nch and co[c1] < r0 + nr:
                    loc.append(max(int(co[c1]), r0) - r0)
                    c1 += 1
                loc.append(min(int(co[c1]), r0 + nr) - r0)
                loc_t = torch.from_numpy(np.asarray(loc, dtype=np.int32)).to(dev)
                check(lib().jxg_lmm_scan_chain(_ptr(grot), nr, n, _ptr(model.S), _ptr(model.xcov), _ptr(model.y), model.p, lo_b,
                                               hi_b, float(tol), int(max_iter), _ptr(loc_t), c1 - c0, carry[c0:].data_ptr(),
                                               with_plrt, nullml_v, o.data_ptr(), ev_p, _stream()))
        elif mode == "lmm":
            ev_p = evals[r0:].data_ptr() if evals is not None else None
            if tables is not None:
                check(lib().jxg_lmm_scan_tab(_ptr(grot), nr, n, _ptr(model.S), _ptr(model.xcov), model.p, lo_b, hi_b,
                                             _ptr(tables), float(tol), int(max_iter), warm, init, with_plrt, nullml_v,
                                             o.data_ptr(), ev_p, _stream()))
            else:
                check(lib().jxg_lmm_scan_exact(_ptr(grot), nr, n, _ptr(model.S), _ptr(model.xcov), _ptr(model.y),
                                               model.p, lo_b, hi_b, float(tol), int(max_iter), warm, init, with_plrt,
                                               nullml_v, o.data_ptr(), ev_p, _stream()))
        elif mode == "splmm":   # score-form test with the null sigma2 = yPy / (n - p) (src/stats/splmm.rs:2567-2880)
            check(lib().jxg_splmm_exact_scan_dev(_ptr(grot), nr, n, model.p, _ptr(w), _ptr(py), _ptr(wx), _ptr(a_dev),
                                                 ypy, n - model.p, o.data_ptr(), _stream()))
        else:
            check(lib().jxg_fvlmm_scan_dev(_ptr(grot), nr, n, model.p, _ptr(w), _ptr(py), _ptr(wx), _ptr(a_dev),
                                           ypy, df, with_plrt, nullml_v, log_det_v, o.data_ptr(), _stream()))
        if times is not None:
            ev_scan[bi][1].record()
        if on_block is not None and not chain_series:
            done = torch.cuda.Event()
            done.record()
            if pending is not None:
                hand_over(pending)       # the previous block, while this one runs
            pending = (r0, r0 + nr, done)
    if chain_series:
        ev_c = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev_c[0].record()
        check(lib().jxg_lmm_series_brent_tab(mk, n, _ptr(model.S), _ptr(model.xcov), model.p, lo_b, hi_b, _ptr(tables), float(tol),
                                             int(max_iter), 0, 0.0, _ptr(scoef), _ptr(sssq), _ptr(co_t), nch, _ptr(carry),
                                             with_plrt, nullml_v, _ptr(out), _ptr(evals) if evals is not None else None,
                                             _stream()))
        ev_c[1].record()
        ev_scan.append(ev_c)
        if on_block is not None:
            done = torch.cuda.Event()
            done.record()
            pending = (0, mk, done)
    if pending is not None:
        hand_over(pending)
    if times is not None:
        torch.cuda.synchronize()
        times.add("rotate", sum(a.elapsed_time(b) for a, b in ev_rot) * 1e-3)
        times.add("scan", sum(a.elapsed_time(b) for a, b in ev_scan) * 1e-3)
    return (out, evals) if return_evals else out


class BlockRotation:
    """Block-diagonal eigenbasis of a sparse GRM whose graph falls into small connected components (thresholded GRMs are
    block diagonal by family): the samples are ordered so that every diagonal block holds whole components, each block has
    its own dense eigendecomposition, and a SNP is rotated block by block with the MFMA rotation kernel writing its columns
    of the rotated row (`jxg_rotate_packed16x_ld`).  Memory and time are O(n B) / O(n B) per SNP for blocks of B samples
    instead of O(n^2): the form the SparseLMM routes take beyond the reach of one dense n x n eigenproblem
    (reference: sparse LLT of K + lambda I, src/stats/spreml.rs:384-512, src/math/cholesky.rs:776-1075)."""

    def __init__(self, packed: torch.Tensor, n_samples: int, sample_idx: np.ndarray, blocks):
        # blocks: list of (offset, nb, ut64 device tensor (nb, nb), rows = eigenvectors); sample_idx: panel sample of every
        # position of the block order
        self.n = int(sum(b[1] for b in blocks))
        self.parts = []
        dev = packed.device
        for off, nb, ut64 in blocks:
            ut32 = torch.empty((nb, nb), dtype=torch.float32, device=dev)
            check(lib().jxg_cast_f64_to_f32(_ptr(ut64), _ptr(ut32), nb * nb, _stream()))
            npad = lib().jxg_num_tiles(nb) * 128
            hi = torch.empty((npad, npad), dtype=torch.float16, device=dev)
            lo = torch.empty((npad, npad), dtype=torch.float16, device=dev)
            check(lib().jxg_ut_split(_ptr(ut32), nb, _ptr(hi), _ptr(lo), SCALE_EXP, _stream()))
            del ut32
            panel = Panel(packed, n_samples, np.asarray(sample_idx[off:off + nb], dtype=np.int64))
            self.parts.append((off, nb, hi, lo, panel))
        self.m = int(packed.shape[0])
        self.device = dev


def scan_rows_splmm_blocks(rot: BlockRotation, p: int, rows: np.ndarray, lut: np.ndarray, fv_state, block_rows=8192):
    """SparseLMM exact scan (`jxg_splmm_exact_scan_dev`, src/stats/splmm.rs:2567-2880) over rows rotated by a block-diagonal
    eigenbasis: per block of SNP rows one rotation launch per diagonal block, then the score-form scan over the full
    rotated rows.  Returns (len(rows), 3) f64 [beta, se, p] on the device."""
    dev = rot.device
    n, mk = rot.n, len(rows)
    out = torch.empty((mk, 3), dtype=torch.float64, device=dev)
    if mk == 0:
        return out
    rows_t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(dev)
    lut_t = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float32)).to(dev)
    lut16 = torch.empty((mk, 16), dtype=torch.uint8, device=dev)
    check(lib().jxg_lut_split(_ptr(lut_t), mk, _ptr(lut16), _stream()))      # general rows: hi / lo split of every LUT
    w, py, wx, a_chol, ypy = fv_state
    a_dev = torch.from_numpy(np.ascontiguousarray(a_chol, dtype=np.float64)).to(dev)
    br = int(min(block_rows, mk))
    if _fused_fixed_lambda(p):
        # every column tile of every diagonal block writes its share of the three weighted sums; G~ is never written
        tiles = int(sum(panel.nt for _o, _nb, _h, _l, panel in rot.parts))
        sums = torch.empty((tiles, br, p + 2), dtype=torch.float64, device=dev)
        for r0 in range(0, mk, br):
            nr = min(br, mk - r0)
            t0 = 0
            for off, nb, hi, lo, panel in rot.parts:
                check(lib().jxg_rotate_packed16x_fused(_ptr(panel.p32), panel.m, nb, rows_t[r0:].data_ptr(), nr,
                                                       lut16[r0:].data_ptr(), None, None, _ptr(hi), _ptr(lo), SCALE_EXP,
                                                       w[off:].data_ptr(), py[off:].data_ptr(), wx[off:].data_ptr(), p,
                                                       _ptr(sums), p + 2, t0, _stream()))
                t0 += panel.nt
            check(lib().jxg_fvlmm_finish_dev(_ptr(sums), tiles, p + 2, nr, n, p, _ptr(a_dev), ypy, n - p, 0, 0.0, 0.0, 1,
                                             out[r0:].data_ptr(), _stream()))
        return out
    grot = torch.empty((br, n), dtype=torch.float32, device=dev)
    for r0 in range(0, mk, br):
        nr = min(br, mk - r0)
        for off, nb, hi, lo, panel in rot.parts:
            check(lib().jxg_rotate_packed16x_ld(_ptr(panel.p32), panel.m, nb, rows_t[r0:].data_ptr(), nr,
                                                lut16[r0:].data_ptr(), None, None, _ptr(hi), _ptr(lo), SCALE_EXP,
                                                grot[:, off:].data_ptr(), n, _stream()))
        check(lib().jxg_splmm_exact_scan_dev(_ptr(grot), nr, n, p, _ptr(w), _ptr(py), _ptr(wx), _ptr(a_dev), ypy, n - p,
                                             out[r0:].data_ptr(), _stream()))
    return out


def scan_rows_splmm_dense(g: np.ndarray, parts, n: int, p: int, fv_state, dev, block_rows=8192):
    """SparseLMM exact scan of DENSE f32 rows `g` (m, n) on the host (`splmm_assoc_pcg_dense_f32`, src/stats/splmm.rs:5464-5650):
    blocks of rows are staged to the device, rotated by the eigenbasis (`parts`: (offset, nb, U' (nb, nb) f32, sample positions of
    the block or None for the identity) -- one part for a dense decomposition, the diagonal blocks otherwise) with the f32
    rotation kernel and scanned (`jxg_splmm_exact_scan_dev`).  -> (m, 3) f64 on the device."""
    m = int(g.shape[0])
    out = torch.empty((m, 3), dtype=torch.float64, device=dev)
    w, py, wx, a_chol, ypy = fv_state
    a_dev = torch.from_numpy(np.ascontiguousarray(a_chol, dtype=np.float64)).to(dev)
    br = int(max(1, min(block_rows, m)))
    grot = torch.empty((br, n), dtype=torch.float32, device=dev)
    for r0 in range(0, m, br):
        nr = min(br, m - r0)
        gb = torch.from_numpy(g[r0:r0 + nr]).to(dev)
        for off, nb, ut32, cols in parts:
            if cols is None and nb == n:
                check(lib().jxg_rotate_dense_f32(_ptr(gb), nr, n, _ptr(ut32), _ptr(grot), _stream()))
            else:
                sub = gb[:, cols].contiguous()
                rb = torch.empty((nr, nb), dtype=torch.float32, device=dev)
                check(lib().jxg_rotate_dense_f32(_ptr(sub), nr, nb, _ptr(ut32), _ptr(rb), _stream()))
                grot[:nr, off:off + nb] = rb
        check(lib().jxg_splmm_exact_scan_dev(_ptr(grot), nr, n, p, _ptr(w), _ptr(py), _ptr(wx), _ptr(a_dev), ypy, n - p,
                                             out[r0:].data_ptr(), _stream()))
    return out


def scan_rows_splmm_factor(rows_f32, m: int, n: int, p: int, csr, diag: np.ndarray, lam: float, vinv_x: np.ndarray,
                           py: np.ndarray, a_chol: np.ndarray, ypy: float, dev, block_rows: int = 256, tol: float = 1e-11,
                           max_iter: int = 1000):
    """SparseLMM exact scan (`exact_scan_blocks_core`, src/stats/splmm.rs:2567-2880) WITHOUT a spectral form of K: the relatedness
    graph holds a connected component beyond one dense eigenproblem (`janusx._SparseFactorReml`).  The reference solves
    (K + lambda I) z = g per SNP with its sparse factor; here every block of `block_rows` decoded rows is the right-hand side
    of one multi-vector conjugate-gradient solve over the CSR image of K in HBM (`jxg_sps_solve_multi`, csrc/k_spsolve.hip),
    then g'V^-1 g, g.Py, g.(V^-1 X) in sample space (`jxg_sps_scan_sums`) and the score-form finish of the other exact scans
    (`jxg_fvlmm_finish_dev`, score_mode = 1).  rows_f32(r0, nr) -> device tensor (nr, n) f32 of the decoded rows r0 .. r0 + nr
    ([0, 2 maf, 1, 2] or flipped, missing = mean, not centred).  -> (m, 3) f64 [beta, se, p] on the device."""
    rowptr, col, val = csr
    out = torch.empty((m, 3), dtype=torch.float64, device=dev)
    if m == 0:
        return out
    br = int(max(64, min(block_rows, (m + 63) // 64 * 64)))
    ldr = int(lib().jxg_sps_ldr(br))
    dinv = torch.from_numpy(1.0 / (np.asarray(diag, dtype=np.float64) + float(lam))).to(dev)
    work = torch.empty(int(lib().jxg_sps_work_doubles(n, ldr)), dtype=torch.float64, device=dev)
    g = torch.empty((n, ldr), dtype=torch.float64, device=dev)
    z = torch.empty((n, ldr), dtype=torch.float64, device=dev)
    sums = torch.empty((br, p + 2), dtype=torch.float64, device=dev)
    py_t = torch.from_numpy(np.ascontiguousarray(py, dtype=np.float64)).to(dev)
    vx_t = torch.from_numpy(np.ascontiguousarray(vinv_x, dtype=np.float64)).to(dev)
    a_dev = torch.from_numpy(np.ascontiguousarray(a_chol, dtype=np.float64)).to(dev)
    info = np.zeros(2, dtype=np.float64)
    for r0 in range(0, m, br):
        nr = min(br, m - r0)
        rows = rows_f32(r0, nr)
        check(lib().jxg_sps_rows_to_cols_f64(_ptr(rows), nr, n, int(rows.stride(0)), _ptr(g), ldr, _stream()))
        check(lib().jxg_sps_solve_multi(n, _ptr(rowptr), _ptr(col), _ptr(val), float(lam), _ptr(dinv), _ptr(g), nr, ldr,
                                        float(tol), int(max_iter), _ptr(z), _ptr(work), info.ctypes.data, _stream()))
        check(lib().jxg_sps_scan_sums(n, _ptr(g), _ptr(z), nr, ldr, _ptr(py_t), _ptr(vx_t), p, _ptr(sums), _ptr(work), _stream()))
        check(lib().jxg_fvlmm_finish_dev(_ptr(sums), 1, p + 2, nr, n, p, _ptr(a_dev), float(ypy), n - p, 0, 0.0, 0.0, 1,
                                         out[r0:].data_ptr(), _stream()))
    return out


def rotate_rows(panel: Panel, model: SpectralModel, rows: np.ndarray, lut: np.ndarray) -> torch.Tensor:
    """G~ = G U for a (small) list of SNP rows, written out: (len(rows), n) f32 on the device (the fp16 hi / lo rotation of
    `scan_rows` without a scan behind it; used for the sampled markers of the SparseLMM gamma estimate)."""
    dev, n, mk = panel.device, model.n, len(rows)
    rows_t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(dev)
    lut_t = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float32)).to(dev)
    hi, lo, usum = model.planes()
    lut16 = torch.empty((mk, 16), dtype=torch.uint8, device=dev)
    rowoff = torch.empty(mk, dtype=torch.float32, device=dev)
    check(lib().jxg_lut_split_rows(_ptr(panel.p32), panel.m, n, _ptr(rows_t), _ptr(lut_t), mk, _ptr(lut16), _ptr(rowoff),
                                   _stream()))
    grot = torch.empty((mk, n), dtype=torch.float32, device=dev)
    check(lib().jxg_rotate_packed16x(_ptr(panel.p32), panel.m, n, _ptr(rows_t), mk, _ptr(lut16), _ptr(rowoff), _ptr(usum),
                                     _ptr(hi), _ptr(lo), SCALE_EXP, _ptr(grot), _stream()))
    return grot


def rotate_rows_blocks(rot: BlockRotation, rows: np.ndarray, lut: np.ndarray) -> torch.Tensor:
    """`rotate_rows` for a block-diagonal eigenbasis: every diagonal block writes its columns of the rotated rows."""
    dev, n, mk = rot.device, rot.n, len(rows)
    rows_t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(dev)
    lut_t = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float32)).to(dev)
    lut16 = torch.empty((mk, 16), dtype=torch.uint8, device=dev)
    check(lib().jxg_lut_split(_ptr(lut_t), mk, _ptr(lut16), _stream()))
    grot = torch.empty((mk, n), dtype=torch.float32, device=dev)
    for off, nb, hi, lo, panel in rot.parts:
        check(lib().jxg_rotate_packed16x_ld(_ptr(panel.p32), panel.m, nb, _ptr(rows_t), mk, _ptr(lut16), None, None, _ptr(hi),
                                            _ptr(lo), SCALE_EXP, grot[:, off:].data_ptr(), n, _stream()))
    return grot


def scan_rows_grammar(panel: Panel, rows: np.ndarray, lut: np.ndarray, x: np.ndarray, score_vec: np.ndarray, r_hat: float,
                      block_rows=65536, on_block=None):
    """SparseLMM approximate (GRAMMAR-gamma) scan over the payload in sample space (`jxg_splmm_grammar_scan_p32`,
    src/stats/splmm.rs:2935-3316 as called by `scan_with_py_and_rhat`, :3318): no rotation, (p + 1) dots per SNP.
    x (n, p) f64 design with intercept, score_vec (n) f64; both are rounded to f32 like `pack_score_design_rhs_f32`.
    Returns (len(rows), 3) f64 [beta, se, p] on the device."""
    dev, n, mk = panel.device, panel.n, len(rows)
    p = int(x.shape[1])
    out = torch.empty((mk, 3), dtype=torch.float64, device=dev)
    if mk == 0:
        return out
    if not (np.all(np.isfinite(score_vec)) and np.all(np.isfinite(x))):
        raise RuntimeError("SparseLMM scan received non-finite dense operand")
    xr = np.concatenate([x, np.asarray(score_vec, dtype=np.float64)[:, None]], axis=1).astype(np.float32).astype(np.float64)
    xtx = np.asarray(x, dtype=np.float64).T @ np.asarray(x, dtype=np.float64)
    ixx = np.linalg.inv(xtx)
    xr_t = torch.from_numpy(np.ascontiguousarray(xr)).to(dev)
    ixx_t = torch.from_numpy(np.ascontiguousarray(ixx)).to(dev)
    rows_t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(dev)
    lut_t = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float32)).to(dev)
    br = int(min(block_rows, mk))
    work = torch.empty(br * (p + 2), dtype=torch.float64, device=dev)
    for r0 in range(0, mk, br):
        nr = min(br, mk - r0)
        check(lib().jxg_splmm_grammar_scan_p32(_ptr(panel.p32), panel.m, n, rows_t[r0:].data_ptr(), nr, lut_t[r0:].data_ptr(),
                                               _ptr(xr_t), p, _ptr(ixx_t), 1.0, float(r_hat), 1.0, _ptr(work),
                                               out[r0:].data_ptr(), _stream()))
        if on_block is not None:
            on_block(r0, out[r0:r0 + nr].cpu().numpy())
    return out


def scan_rows_lm(panel: Panel, rows: np.ndarray, af: np.ndarray, x: np.ndarray, y: np.ndarray, on_block=None):
    """Plain LM scan of the given SNP rows of a resident panel (`lm_block_assoc_packed`, src/stats/glm.rs:3550-3860; the `LM`
    wrapper python/janusx/pyBLUP/assoc.py:2185-2228): mean-imputed additive decode with the rows' allele frequencies, `x`
    includes the intercept.  Returns a (len(rows), 4) f64 device tensor [beta, se, pwald (Student t), plrt]."""
    from .janusx import lm_precompute_ixx_qr
    dev = panel.device
    n = panel.n
    y = np.ascontiguousarray(y, dtype=np.float64).ravel()
    x = np.ascontiguousarray(x, dtype=np.float64)
    if y.shape[0] != n or x.shape[0] != n:
        raise RuntimeError(f"selected sample count {panel.n} != len(y) {y.shape[0]}")
    q0 = int(x.shape[1])
    mk = len(rows)
    out = torch.empty((mk, 4), dtype=torch.float64, device=dev)
    if mk == 0:
        return out
    ixx = lm_precompute_ixx_qr(x)
    xr = np.empty((n, q0 + 1), dtype=np.float64)
    yy = np.zeros(1, dtype=np.float64)
    check(lib().jx_lm_residualize(y.ctypes.data, x.ctypes.data, ixx.ctypes.data, n, q0, xr.ctypes.data, yy.ctypes.data))
    mean_g = np.clip(np.float32(2.0) * np.asarray(af, dtype=np.float32), np.float32(0.0), np.float32(2.0))
    lut = np.zeros((mk, 4), dtype=np.float32)
    lut[:, 1], lut[:, 2], lut[:, 3] = mean_g, 1.0, 2.0
    rows_t = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).to(dev)
    lut_t = torch.from_numpy(lut).to(dev)
    xr_t = torch.from_numpy(xr).to(dev)
    ixx_t = torch.from_numpy(ixx).to(dev)
    work = torch.empty(mk * (q0 + 2), dtype=torch.float64, device=dev)
    check(lib().jxg_lm_scan_p32(_ptr(panel.p32), panel.m, n, _ptr(rows_t), mk, _ptr(lut_t), _ptr(xr_t), q0, _ptr(ixx_t),
                                float(yy[0]), _ptr(work), _ptr(out), _stream()))
    if on_block is not None:
        on_block(0, out[:, :3].cpu().numpy())
    return out


@dataclass
class GwasResult:
    keep: np.ndarray        # (m,) bool, kept SNPs in BED order
    af: np.ndarray          # (m_kept,) f32
    miss: np.ndarray        # (m_kept,) f32
    stats: np.ndarray       # (m_kept, 3) f64 beta, se, p
    null: NullFit
    grm_eff_m: int
    times: dict
    model_tag: str = None   # route that produced `stats`: the requested mode, or "lm" after the null LRT fallback
    null_lrt: tuple = None  # (switch_to_lm, LRT statistic, p) of `gwas_lmm_lm_null_lrt_decision` when it was evaluated


def build_grm(packed: torch.Tensor, n_samples: int, method=1, maf=0.02, geno=0.05, panel: Panel = None,
              payload_sharded: bool = False):
    """`build_grm_streaming` -> `grm_stream_bed_f32` on an HBM-resident payload (all samples; het filter off,
    python/janusx/assoc/workflow.py:2928-3095). Returns (K f32 device tensor (n,n), eff_m, panel).
    Several ranks (torch.distributed initialised, SURVEY.md 8(e) / BASELINE north_star): every rank accumulates the partial
    Z Z' of ITS SNPs -- a contiguous share of the kept rows of a replicated payload, or all kept rows of its own payload shard
    (`payload_sharded`) --, ONE sum-reduction of the lower-triangle tiles of the f64 accumulators (RCCL over xGMI) and of the two
    scalar denominators follows, and every rank finalises the same K."""
    if panel is None:
        panel = Panel(packed, n_samples)
    panel.sharded = bool(payload_sharded)
    counts = panel.counts()
    n = panel.n
    gkeep, mean_g, scale, flip, var = st.stream_grm_row_prepare(counts, n, method, maf, geno, 0.0)
    grows = np.nonzero(gkeep)[0]
    lo, hi = _my_slice(len(grows), payload_sharded)
    grows = grows[lo:hi]
    panel.grm_rows_local = int(len(grows))          # this rank's share (bench.py prices its launch with it)
    totals = torch.tensor([float(np.sum(var[grows])), float(len(grows))], dtype=torch.float64, device=panel.device)
    world = dist_info()[1]
    if world > 1:
        from .dist import allreduce_sum_
        allreduce_sum_(totals)
    var_sum, eff = (float(v) for v in totals.cpu().numpy())
    if int(round(eff)) == 0:
        raise RuntimeError("No SNPs remained after filtering; GRM is empty.")
    denom = var_sum if method == 1 else eff
    if not (math.isfinite(denom) and denom > 0.0):
        raise RuntimeError("invalid centered GRM denominator: sum(2p(1-p)) <= 0")
    glut = st.grm_lut_from_mean_scale(mean_g[grows], scale[grows], flip[grows])
    acc = grm_accumulate(panel, grows, glut)
    if world > 1:
        from .dist import allreduce_grm_accumulator_
        allreduce_grm_accumulator_(acc)
    k32 = grm_finalize(acc, n, denom, torch.float32)
    return k32, int(round(eff)), panel


def run_trait(packed: torch.Tensor, n_samples: int, k: torch.Tensor, keep_idx, y: np.ndarray, x: np.ndarray,
              mode="lmm", maf=0.02, geno=0.05, het=1.0, max_iter=30, tol=1e-2, warm_start=False, on_rows=None,
              force_model=True, payload_sharded: bool = False, warm_chain=None):
    """One trait of `run_chunked_gwas_lmm_lm` (python/janusx/assoc/workflow_model_stream.py:464-1480):
    eigh of K[keep, keep] + 1e-6 I, spectral null model, QC on the trait's samples, rotate + scan.
    `x` includes the intercept column; `y`, `x` are already restricted to keep_idx (in that order).
    `on_rows(keep, af, miss, ncol, model_tag)` (optional) is called once the kept SNP set and the route are known and
    returns the `on_block` callback of `scan_rows` (or None): the streaming writer is opened there.
    force_model=False: the null likelihood-ratio test against the plain LM (`gwas_lmm_lm_null_lrt_decision`) runs after the
    null fit and, when it finds no polygenic variance (p >= 0.05), the trait is scanned with the LM instead
    (workflow_model_stream.py:930-963; result.model_tag == "lm", three columns beta, se, Student-t p).
    Several ranks (torch.distributed initialised; `init_distributed`): the eigendecomposition is shared out over the ranks, the
    null model is fitted on every rank (deterministic), every rank scans ITS kept SNPs -- a contiguous share of the kept rows of
    a replicated payload, or the kept rows of its own payload shard (`payload_sharded`) -- and the result rows are gathered in
    BED order on every rank; `on_rows` is called on rank 0 only, with the whole table as one block.
    warm_chain = (chunk_rows, pieces) ('lmm' only): the reference's default scan of this route (`lmm_reml_assoc_bed_to_tsv_f32`
    as workflow_model_stream.py:1449-1480 calls it: init_log10_lbd = log10 lambda0, rotate_block_rows = the CLI chunk size,
    warm-start chain on, src/stats/lmm.rs:2627): a chain is the kept rows of one chunk of `chunk_rows` SNP rows of the file
    (`stats.warm_chain_blocks_bed`), cut into `pieces` by halving.  On several ranks over a replicated payload the rows are dealt
    at chain boundaries (same table as one rank); with a payload shard per rank the chunks count from the shard's first row."""
    rank, world = dist_info()
    keep_idx = None if keep_idx is None else np.asarray(keep_idx, dtype=np.int64)
    s, ut64 = eigh_from_grm(k, 1e-6, keep_idx, f32_consumer=True)     # only the f32 U^T of SpectralModel is kept
    model = SpectralModel(s, ut64, x, y)
    del ut64
    panel = Panel(packed, n_samples, keep_idx)
    panel.sharded = bool(payload_sharded)
    counts = panel.counts()
    n = panel.n
    keep, af, miss = st.gwas_scan_row_stats(counts, n, maf, geno, het)
    rows = np.nonzero(keep)[0]
    chain_off = None
    if warm_chain is not None and mode == "lmm":
        chain_off = st.warm_chain_offsets(st.warm_chain_blocks_bed(rows, panel.m, int(warm_chain[0])), len(rows), int(warm_chain[1]))
        warm_start = True            # the route passes init_log10_lbd = log10 lambda0 (workflow_model_stream.py:1436-1440)
    if world > 1:
        return _run_trait_ranks(panel, model, counts, keep, af, miss, rows, y, x, mode, max_iter, tol, warm_start, on_rows,
                                force_model, payload_sharded, chain_off)
    lut = st.scan_lut_from_counts(af[rows], np.zeros(len(rows), dtype=bool), counts[rows], n)
    lrt = None
    if not force_model:
        from .janusx import gwas_lmm_lm_null_lrt_decision
        sw, stat, pv, _ml = gwas_lmm_lm_null_lrt_decision(y, np.asarray(x)[:, 1:], model.null.ml0)
        lrt = (sw, stat, pv)
        if sw:
            # the reference's LM routes print the COUNT of missing samples in the miss column (AssocMissValue::Count,
            # src/io/assoc2tsv.rs:452-458; the streaming writer formats rint(miss) as an integer)
            miss_cnt = counts[rows, 0].astype(np.float32)
            on_block = on_rows(keep, af[rows], miss_cnt, 3, "lm") if on_rows is not None else None
            out = scan_rows_lm(panel, rows, af[rows], x, y, on_block=on_block)
            res = GwasResult(keep, af[rows], miss_cnt, out[:, :3].cpu().numpy(), model.null, 0, {})
            res.model_tag, res.null_lrt = "lm", lrt
            return res
    on_block = on_rows(keep, af[rows], miss[rows], 6 if mode == "lmm2" else 3, mode) if on_rows is not None else None
    if mode == "lmm":
        init = math.log10(model.null.lbd) if (warm_start and model.null.lbd > 0) else None
        if init is not None:
            init = min(max(init, model.null.bounds[0]), model.null.bounds[1])
        out = scan_rows(panel, model, rows, lut, "lmm", max_iter=max_iter, tol=tol, init_log10_lbd=init, on_block=on_block,
                        chain_off=chain_off)
    elif mode == "lmm2":
        # null ML by Brent on -ml_loglike, seeded with the REML optimum (src/stats/lmm.rs:2902-2921; the workflow passes
        # bounds, max_iter = 30, tol = 1e-2 and init_log10_lbd_reml = log10 lambda0: workflow_model_stream.py:1499-1590)
        o2 = torch.empty(2, dtype=torch.float64, device=packed.device)
        lo_b, hi_b = model.null.bounds
        init = min(max(math.log10(model.null.lbd), lo_b), hi_b) if model.null.lbd > 0 else None
        check(lib().jxg_lmm2_null_ml(_ptr(model.S), _ptr(model.xcov), _ptr(model.y), n, model.p, lo_b, hi_b, int(max_iter),
                                     float(tol), 1 if init is not None else 0, float(init or 0.0), o2.data_ptr(),
                                     _stream()))
        ml0 = float(o2.cpu().numpy()[1])
        if not math.isfinite(ml0):
            raise RuntimeError("failed to optimize null ML for LMM2 unified scan")
        out = scan_rows(panel, model, rows, lut, "lmm2", max_iter=max_iter, tol=tol, init_log10_lbd=init, nullml=ml0,
                        on_block=on_block)
    else:
        out = scan_rows(panel, model, rows, lut, "fvlmm", on_block=on_block)
    res = GwasResult(keep, af[rows], miss[rows], out.cpu().numpy(), model.null, 0, {})
    res.model_tag, res.null_lrt = mode, lrt
    return res


def _run_trait_ranks(panel, model, counts, keep, af, miss, rows, y, x, mode, max_iter, tol, warm_start, on_rows, force_model,
                     payload_sharded, chain_off=None):
    """Scan stage of `run_trait` on several ranks: this rank's rows -> gather in BED order -> rank 0 hands the table on."""
    rank, world = dist_info()
    n = panel.n
    lo, hi = _my_slice(len(rows), payload_sharded)
    my_chain = None
    if chain_off is not None and mode == "lmm":
        if not payload_sharded:
            # deal whole chains: the cut nearest to the even share, so that no chain starts without its predecessor's state
            cuts = st.deal_whole_chains(chain_off, world)
            lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        inside = chain_off[(chain_off >= lo) & (chain_off <= hi)] - lo
        my_chain = np.unique(np.concatenate([[0], inside, [hi - lo]])).astype(np.int64)
    mine = rows[lo:hi]
    lrt, tag, ncol = None, mode, (6 if mode == "lmm2" else 3)
    miss_col = miss[rows]
    if not force_model:
        from .janusx import gwas_lmm_lm_null_lrt_decision
        sw, stat, pv, _ml = gwas_lmm_lm_null_lrt_decision(y, np.asarray(x)[:, 1:], model.null.ml0)   # same on every rank
        lrt = (sw, stat, pv)
        if sw:
            tag, ncol = "lm", 3
            miss_col = counts[rows, 0].astype(np.float32)
    if tag == "lm":
        out = scan_rows_lm(panel, mine, af[mine], x, y)[:, :3].contiguous()
    else:
        lut = st.scan_lut_from_counts(af[mine], np.zeros(len(mine), dtype=bool), counts[mine], n)
        init, nullml = None, None
        lo_b, hi_b = model.null.bounds
        if mode == "lmm" and warm_start and model.null.lbd > 0:
            init = min(max(math.log10(model.null.lbd), lo_b), hi_b)
        if mode == "lmm2":
            o2 = torch.empty(2, dtype=torch.float64, device=panel.device)
            init = min(max(math.log10(model.null.lbd), lo_b), hi_b) if model.null.lbd > 0 else None
            check(lib().jxg_lmm2_null_ml(_ptr(model.S), _ptr(model.xcov), _ptr(model.y), n, model.p, lo_b, hi_b, int(max_iter),
                                         float(tol), 1 if init is not None else 0, float(init or 0.0), o2.data_ptr(),
                                         _stream()))
            nullml = float(o2.cpu().numpy()[1])
            if not math.isfinite(nullml):
                raise RuntimeError("failed to optimize null ML for LMM2 unified scan")
        if mode in ("lmm", "lmm2"):
            out = scan_rows(panel, model, mine, lut, mode, max_iter=max_iter, tol=tol, init_log10_lbd=init, nullml=nullml,
                            chain_off=my_chain if (mode == "lmm" and len(mine)) else None)
        else:
            out = scan_rows(panel, model, mine, lut, "fvlmm")
    out = _allgather_rows(out)
    keep_all, af_k, miss_k = keep, af[rows], miss_col
    if payload_sharded:       # the QC columns are per shard too: concatenate them in rank (= BED) order
        keep_all, af_k, miss_k = _allgather_np(keep), _allgather_np(af[rows]), _allgather_np(miss_col)
    stats = out.cpu().numpy()
    if on_rows is not None and rank == 0:
        on_block = on_rows(keep_all, af_k, miss_k, ncol, tag)
        if on_block is not None:
            on_block(0, stats)
    res = GwasResult(keep_all, af_k, miss_k, stats, model.null, 0, {})
    res.model_tag, res.null_lrt = tag, lrt
    return res


def run_gwas(packed: torch.Tensor, n_samples: int, y: np.ndarray, covar: np.ndarray = None, mode="lmm",
             maf=0.02, geno=0.05, het=1.0, grm_method=1, max_iter=30, tol=1e-2, timing=True,
             payload_sharded: bool = False, warm_chain=None) -> GwasResult:
    """`jx gwas -lmm/-fvlmm` on an HBM-resident payload (all samples phenotyped): GRM -> eigh -> null REML -> per-SNP scan.
    Several ranks (torch.distributed initialised; `init_distributed`): the composition of SURVEY.md 8(e) -- SNP-sharded GRM with
    one sum-reduction (`build_grm`), eigenvectors shared out over the ranks (`eigh_from_grm`), SNP-sharded scan, result rows
    gathered in BED order on every rank; `packed` is the whole payload on every rank, or this rank's contiguous SNP shard of it
    (`payload_sharded`)."""
    tm = StageTimes()

    def tick():
        if timing:
            torch.cuda.synchronize()
        return time.perf_counter()

    t0 = tick()
    panel = Panel(packed, n_samples)
    counts = panel.counts()
    n = panel.n
    t1 = tick()
    tm.add("prep", t1 - t0)
    # GRM: `grm_stream_bed_f32` defaults maf/geno from the CLI, het filter off (workflow.py:3095)
    k32, geff, _ = build_grm(packed, n_samples, grm_method, maf, geno, panel=panel, payload_sharded=payload_sharded)
    t2 = tick()
    tm.add("grm", t2 - t1)
    s, ut64 = eigh_from_grm(k32, 1e-6, f32_consumer=True)             # only the f32 U^T of SpectralModel is kept
    t3 = tick()
    tm.add("eigh", t3 - t2)
    x = np.ones((n, 1)) if covar is None else np.concatenate([np.ones((n, 1)), np.asarray(covar, dtype=np.float64)], 1)
    model = SpectralModel(s, ut64, x, y)
    del ut64
    t4 = tick()
    tm.add("null", t4 - t3)
    keep, af, miss = st.gwas_scan_row_stats(counts, n, maf, geno, het)
    rows = np.nonzero(keep)[0]
    lo, hi = _my_slice(len(rows), payload_sharded)
    mine = rows[lo:hi]
    lut = st.scan_lut_from_counts(af[mine], np.zeros(len(mine), dtype=bool), counts[mine], n)
    if mode == "lmm":
        chain_off, init = None, None
        if warm_chain is not None:       # the reference CLI's scan: chains over chunks of the file's rows, seeded with lambda0
            chain_off = st.warm_chain_offsets(st.warm_chain_blocks_bed(mine, panel.m, int(warm_chain[0])), len(mine),
                                              int(warm_chain[1]))
            if model.null.lbd > 0:
                init = min(max(math.log10(model.null.lbd), model.null.bounds[0]), model.null.bounds[1])
        out = scan_rows(panel, model, mine, lut, "lmm", max_iter=max_iter, tol=tol, init_log10_lbd=init,
                        times=tm if timing else None, chain_off=chain_off)
    else:
        out = scan_rows(panel, model, mine, lut, "fvlmm", times=tm if timing else None)
    res = _allgather_rows(out).cpu().numpy()
    af_k, miss_k = af[rows], miss[rows]
    if payload_sharded and dist_info()[1] > 1:
        keep, af_k, miss_k = _allgather_np(keep), _allgather_np(af_k), _allgather_np(miss_k)
    t5 = tick()
    tm.add("scan_total", t5 - t4)
    tm.add("total", t5 - t0)
    return GwasResult(keep, af_k, miss_k, res, model.null, geff, tm.t)
