"""Drop-in mirror of the reference's native module ``janusx.janusx`` for the mixed-model hot path.

Every function here has the name, argument meaning, defaults and error behaviour of the PyO3 function it
replaces (signatures: /root/reference/src/lib.rs:691-1005 registrations, ``#[pyo3(signature=...)]`` at the cited
lines) and forwards to the HIP library through the C ABI (``include/jxgpu.h``, host layer).  Arguments that only
steer CPU threading in the reference (``threads``, ``block_cols``, ``rotate_block_rows``, ``mmap_window_mb``) are
accepted and ignored; ``progress_callback(done, total)`` follows the reference's cadence on the packed association
scans (C-ABI hook `jx_set_progress`: every `progress_every` rows, default one 8192-row block; an exception raised by the
callback stops the scan) and is called once at completion elsewhere.

A reference call site such as ``jxrs.grm_packed_f32(packed, n, flip, maf, idx, method=1)`` works unchanged
with ``import janusx_amd.janusx as jxrs``.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from ._lib import check, lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _is_device_tensor(a):
    """A torch CUDA tensor (a payload that is already resident in HBM)."""
    return hasattr(a, "is_cuda") and bool(getattr(a, "is_cuda"))


def _payload(packed, n_samples):
    """Packed 2-bit payload (m, ceil(n / 4)) uint8 as (object that owns the memory, pointer, m): a C-contiguous numpy
    array on the host, or a torch CUDA tensor used in place (extension of the reference's numpy-only interface: a panel
    that already lives in HBM is never staged through the host)."""
    if _is_device_tensor(packed):
        import torch
        if packed.dtype != torch.uint8 or packed.dim() != 2:
            raise RuntimeError("packed must be 2D uint8 (m, bytes_per_snp)")
        pk = packed.contiguous()
        ptr, shape = C.c_void_p(pk.data_ptr()), tuple(pk.shape)
    else:
        pk = _c(packed, np.uint8)
        if pk.ndim != 2:
            raise RuntimeError("packed must be 2D (m, bytes_per_snp)")
        _guard_host_payload(pk.nbytes, "packed")
        ptr, shape = _p(pk), pk.shape
    if shape[1] != (int(n_samples) + 3) // 4:
        raise RuntimeError(f"packed second dimension mismatch: got {shape[1]}, expected {(int(n_samples) + 3) // 4}")
    return pk, ptr, int(shape[0])


def _guard_host_payload(nbytes, what):
    """A host array handed to the device layer is copied once more on its way (contiguous copy / staging): refuse clearly
    when that cannot fit instead of driving the machine out of memory (an n = 200 000 x m = 1 000 000 payload is 50 GB)."""
    try:
        import psutil
        avail = int(psutil.virtual_memory().available)
    except Exception:   # noqa: BLE001 - no psutil: no guard
        return
    if int(nbytes) > (1 << 30) and int(nbytes) > avail // 2:
        raise RuntimeError(f"{what}: a host array of {int(nbytes) / 2**30:.1f} GiB with {avail / 2**30:.1f} GiB of host memory "
                           "available cannot be staged safely; pass the payload as a device (torch CUDA uint8) tensor or "
                           "split the SNP rows over several calls")


def _opt_idx(a):
    if a is None:
        return None, 0
    a = _c(a, np.int64).ravel()
    return a, int(a.shape[0])


def _done(cb, total):
    if cb is not None:
        cb(int(total), int(total))


class _progress_hook:
    """Installs `progress_callback(done, total)` as the C-ABI progress hook for the duration of one host-layer call
    (the reference's cadence: every `progress_every` rows, default one internal block; src/stats/lmm.rs:3214-3330).  An
    exception raised by the callback (KeyboardInterrupt included) stops the native loop and is re-raised here."""
    _CB = C.CFUNCTYPE(C.c_int, C.c_int64, C.c_int64, C.c_void_p)

    def __init__(self, cb, every):
        self.cb, self.every, self.exc, self.fn = cb, int(every or 0), None, None

    def __enter__(self):
        if self.cb is not None:
            def tramp(done, total, _user):
                try:
                    self.cb(int(done), int(total))
                    return 0
                except BaseException as e:      # noqa: BLE001 - handed back to the caller of the native function
                    self.exc = e
                    return 1
            self.fn = self._CB(tramp)
            lib().jx_set_progress(self.fn, None, self.every)
        return self

    def __exit__(self, et, ev, tb):
        if self.cb is not None:
            lib().jx_set_progress(None, None, 0)
        if self.exc is not None:
            raise self.exc
        return False


# ------------------------------------------------------------------------------------------------
# GRM  (src/stats/grm.rs)
# ------------------------------------------------------------------------------------------------

def _grm_packed(packed, n_samples, row_flip, row_maf, sample_indices, method, out_dtype, with_stats):
    packed = _c(packed, np.uint8)
    if packed.ndim != 2:
        raise RuntimeError("packed must be 2D (m, bytes_per_snp)")
    m = int(packed.shape[0])
    n_samples = int(n_samples)
    if n_samples <= 0:
        raise RuntimeError("n_samples must be > 0")
    if m == 0:
        raise RuntimeError("packed must contain at least one SNP row")
    bps = (n_samples + 3) // 4
    if packed.shape[1] != bps:
        raise RuntimeError(f"packed length mismatch: got {packed.size}, expected {m * bps}")
    flip = _c(np.asarray(row_flip).astype(np.uint8), np.uint8).ravel()
    maf = _c(row_maf, np.float32).ravel()
    if maf.shape[0] != m or flip.shape[0] != m:
        raise RuntimeError(f"row_maf length mismatch: got {maf.shape[0]}, expected {m}")
    idx, n_sel = _opt_idx(sample_indices)
    n = n_sel if idx is not None else n_samples
    out = np.empty((n, n), dtype=out_dtype)
    row_sum = np.zeros(m, dtype=np.float64) if with_stats else None
    varsum = np.zeros(1, dtype=np.float64)
    check(lib().jx_grm_packed(_p(packed), m, n_samples, _p(flip), _p(maf), _p(idx), n_sel, int(method), _p(out),
                              1 if out_dtype == np.float64 else 0, _p(row_sum), _p(varsum)))
    return out, row_sum, float(varsum[0])


def grm_packed_f32(packed, n_samples, row_flip, row_maf, sample_indices=None, method=1, block_cols=65536,
                   threads=0, progress_callback=None, progress_every=0):
    """src/stats/grm.rs:3053-3066 -> f32 (n, n)."""
    k, _, _ = _grm_packed(packed, n_samples, row_flip, row_maf, sample_indices, method, np.float32, False)
    _done(progress_callback, np.asarray(packed).shape[0])
    return k


def grm_packed_f64(packed, n_samples, row_flip, row_maf, sample_indices=None, method=1, block_cols=65536,
                   threads=0, progress_callback=None, progress_every=0):
    """src/stats/grm.rs:3596 -> f64 (n, n) (same f32-block/f64-merge arithmetic as the f32 entry)."""
    k, _, _ = _grm_packed(packed, n_samples, row_flip, row_maf, sample_indices, method, np.float64, False)
    _done(progress_callback, np.asarray(packed).shape[0])
    return k


def grm_packed_f32_with_stats(packed, n_samples, row_flip, row_maf, sample_indices=None, method=1,
                              block_cols=65536, threads=0, progress_callback=None, progress_every=0):
    """src/stats/grm.rs:5583 -> (f32 (n,n), row_sum f64 (m), varsum)."""
    k, rs, vs = _grm_packed(packed, n_samples, row_flip, row_maf, sample_indices, method, np.float32, True)
    _done(progress_callback, np.asarray(packed).shape[0])
    return k, rs, vs


def grm_packed_f64_with_stats(packed, n_samples, row_flip, row_maf, sample_indices=None, method=1,
                              block_cols=65536, threads=0, progress_callback=None, progress_every=0):
    """src/stats/grm.rs:5611-5651 -> (f64 (n,n), row_sum f64 (m), varsum)."""
    k, rs, vs = _grm_packed(packed, n_samples, row_flip, row_maf, sample_indices, method, np.float64, True)
    _done(progress_callback, np.asarray(packed).shape[0])
    return k, rs, vs


def _normalize_spgrm_path(prefix):
    """src/stats/spgrm.rs:450-469."""
    t = str(prefix).strip()
    if not t:
        return ""
    if t.lower().endswith(".spgrm") or t.lower().endswith(".jxgrm"):
        return t
    if os.path.exists(t + ".jxgrm") and not os.path.exists(t + ".spgrm"):
        return t + ".jxgrm"
    return t + ".spgrm"


def _spgrm_packed(packed, n_samples, row_flip, row_maf, out_prefix, sample_indices, method, threshold, abs_threshold,
                  stream_denominator):
    n_samples = int(n_samples)
    pk, pk_ptr, pk_rows = _payload(packed, n_samples)
    flip = np.ascontiguousarray(np.asarray(row_flip).astype(np.uint8)).ravel()
    maf = _c(row_maf, np.float32).ravel()
    m = int(flip.shape[0])
    if m and maf.shape[0] != m:
        raise RuntimeError(f"Sparse GRM row_maf length mismatch: got {maf.shape[0]}, expected {m}")
    if m and pk_rows != m:
        raise RuntimeError(f"Sparse GRM packed length mismatch: got {pk_rows * ((n_samples + 3) // 4)}, expected "
                           f"{m * ((n_samples + 3) // 4)}")
    idx, n_sel = _opt_idx(sample_indices)
    if sample_indices is not None and n_sel == 0:
        raise RuntimeError("Sparse GRM sample_indices must not be empty")
    out_path = _normalize_spgrm_path(out_prefix)
    if not out_path:
        raise RuntimeError("Sparse GRM output prefix must not be empty")
    out_n, out_nnz = C.c_int64(0), C.c_int64(0)
    from .pipeline import dist_info
    rank, world = dist_info()
    if world > 1:
        # one process per GPU (SURVEY.md 8(e), configs[5]): the row panels of the lower triangle are dealt over the ranks, every
        # rank thresholds its own and leaves them in `<out>.part<rank>`; rank 0 joins the parts into the one `.spgrm` file (the
        # output directory is shared: one node).  No data-path collective: two barriers around the merge.
        import torch.distributed as dist
        check(lib().jx_spgrm_set_part(rank, world))
    try:
        check(lib().jx_spgrm_packed_to_jxgrm(pk_ptr, m, n_samples, _p(flip), _p(maf), _p(idx) if idx is not None else None,
                                             n_sel, int(method), float(threshold), int(bool(abs_threshold)),
                                             int(bool(stream_denominator)), out_path.encode(), C.byref(out_n),
                                             C.byref(out_nnz)))
    finally:
        if world > 1:
            lib().jx_spgrm_set_part(0, 1)
    if world > 1:
        dist.barrier()
        if rank == 0:
            check(lib().jx_spgrm_merge_parts(out_path.encode(), int(out_n.value), world, C.byref(out_nnz)))
        dist.barrier()
        nnz_t = _bcast_int(int(out_nnz.value))
        return out_path, int(out_n.value), nnz_t
    return out_path, int(out_n.value), int(out_nnz.value)


def _bcast_int(v):
    """Rank 0's integer on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v)], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.to(torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(t, src=0)
    return int(t.cpu()[0])


def _scan_my_rows(scan, rows, lut):
    """SNP-sharded SparseLMM scan: `scan(rows, lut)` over this rank's contiguous share of the rows, the per-rank tables
    concatenated in rank (= row) order on every rank.  One rank: the plain call."""
    from . import pipeline as pl
    if pl.dist_info()[1] == 1:
        return scan(rows, lut)
    lo, hi = pl._my_slice(len(rows), False)
    return pl.gather_results(scan(rows[lo:hi], lut[lo:hi]).contiguous())


def spgrm_packed_to_jxgrm(packed, n_samples, row_flip, row_maf, out_prefix, sample_indices=None, method=1,
                          threshold=0.05, abs_threshold=False, block_rows=0, sample_block=0, threads=0,
                          progress_callback=None, progress_every=0):
    """src/stats/spgrm.rs:5201-5278: sparse (thresholded, lower-triangle CSC) GRM of a packed panel written as
    `<out_prefix>.spgrm` -> (path, n_samples_used, nnz).  The tile / spill planning arguments of the reference
    (block_rows, sample_block, threads) have no counterpart: the whole accumulator lives in HBM."""
    out = _spgrm_packed(packed, n_samples, row_flip, row_maf, out_prefix, sample_indices, method, threshold,
                        abs_threshold, False)
    if progress_callback is not None:
        progress_callback(1, 1)
    return out


def prepare_bed_logic_meta_selected(prefix, sample_indices=None, maf_threshold=0.0, max_missing_rate=1.0, het_threshold=1.0,
                                    snps_only=False, mmap_window_mb=None, threads=1):
    """src/io/gfreader.rs:7108-7235 -> `prepare_bed_logic_meta_owned_for_stats_samples_with_mmap_window` (:5236-5480): the QC
    pre-pass of the SparseLMM / sparse-GRM routes over the selected samples (python/janusx/assoc/workflow_model_packed.py:1106,
    1248) -> (row_source_indices i64 (kept), missing_rate f32 (kept), maf f32 (kept) = ALT allele frequency, row_flip bool
    (kept, all False), site_keep bool (all rows), n_samples_full, n_snps_total).  Per-SNP counts are popcounts on the device,
    the O(m) f32 decisions run on the host with the reference's expressions (`stats.packed_prep_row_stats`)."""
    from . import stats as st
    from .bed import snps_only_mask, stage_bed_payload
    if not (0.0 <= maf_threshold <= 0.5):
        raise ValueError("maf_threshold must be within [0, 0.5]")
    if not (0.0 <= max_missing_rate <= 1.0):
        raise ValueError("max_missing_rate must be within [0, 1.0]")
    if not (0.0 <= het_threshold <= 1.0):
        raise ValueError("het_threshold must be within [0, 1.0]")
    packed, n_fam, bim = stage_bed_payload(_bed_prefix(prefix), mmap_window_mb)
    if n_fam == 0:
        raise RuntimeError("no samples found in PLINK input")
    idx, n_sel = _opt_idx(sample_indices)
    if idx is not None and n_sel and (idx.min() < 0 or idx.max() >= n_fam):
        bad = int(idx[(idx < 0) | (idx >= n_fam)][0])
        raise ValueError(f"sample index out of range: {bad} for n_samples={n_fam}")
    n_stats = n_sel if (idx is not None and n_sel) else n_fam
    counts = bed_row_counts(packed, n_fam, idx if (idx is not None and n_sel) else None)
    keep, miss, maf, _std = st.packed_prep_row_stats(counts, n_stats, np.float32(maf_threshold), np.float32(max_missing_rate),
                                                     np.float32(het_threshold))
    if snps_only:
        keep = keep & snps_only_mask(bim)
    if not keep.any():
        raise RuntimeError("No SNPs left after packed BED filtering. Please relax thresholds.")
    rows = np.nonzero(keep)[0].astype(np.int64)
    return (rows, np.ascontiguousarray(miss[rows], dtype=np.float32), np.ascontiguousarray(maf[rows], dtype=np.float32),
            np.zeros(len(rows), dtype=bool), np.ascontiguousarray(keep, dtype=bool), int(n_fam), int(packed.shape[0]))


def load_bim_columns(path_or_prefix, row_indices=None):
    """src/io/gfreader.rs:8813-8862 -> (chrom, pos, snp, allele0, allele1) lists of a PLINK prefix (or explicit .bed/.bim/.fam
    path), optionally of the rows `row_indices` (the result-table metadata of python/janusx/assoc/workflow.py:4633)."""
    from .bed import read_bim
    p = str(path_or_prefix).strip()
    if not p:
        raise ValueError("path_or_prefix must not be empty")
    explicit = p.lower().endswith((".bed", ".bim", ".fam"))
    if not (explicit or all(os.path.exists(p + e) for e in (".bed", ".bim", ".fam"))):
        raise ValueError("load_bim_columns requires a PLINK BED/BIM/FAM prefix or explicit PLINK file path")
    bim = read_bim(_bed_prefix(p))
    if row_indices is None:
        sel = range(len(bim.chrom))
    else:
        sel = [int(v) for v in np.asarray(row_indices).ravel()]
        for v in sel:
            if v < 0:
                raise ValueError(f"row_indices must be non-negative, got {v}")
            if v >= len(bim.chrom):
                raise RuntimeError(f"BIM row index out of range: {v} >= {len(bim.chrom)}")
    return ([bim.chrom[j] for j in sel], [int(bim.pos[j]) for j in sel], [bim.snp[j] for j in sel],
            [bim.a0[j] for j in sel], [bim.a1[j] for j in sel])


def spgrm_bed_to_jxgrm(prefix, out_prefix=None, sample_indices=None, method=1, threshold=0.05, abs_threshold=False,
                       maf_threshold=0.02, max_missing_rate=0.05, het_threshold=0.0, snps_only=False, block_rows=0,
                       sample_block=0, threads=0, mmap_window_mb=None, progress_callback=None, progress_every=0):
    """src/stats/spgrm.rs:5280-5356 -> `spgrm_bed_to_jxgrm_core` :4937-5025: metadata pre-pass over the selected
    samples (`prepare_bed_logic_meta_owned_for_stats_samples_with_mmap_window`, src/io/gfreader.rs:5236-5480: the
    packed-prep QC rule, maf := alt allele frequency, no flips), then the stream core (:3910-4264) whose centred
    denominator is the f64 sum of 2p(1-p) whatever the sample selection.  -> (path, n_samples_used, nnz)."""
    from . import stats as st
    from .bed import read_bed_payload, snps_only_mask
    if not (0.0 <= maf_threshold <= 0.5):
        raise RuntimeError("maf_threshold must be within [0, 0.5]")
    if not (0.0 <= max_missing_rate <= 1.0):
        raise RuntimeError("max_missing_rate must be within [0, 1.0]")
    if not (0.0 <= het_threshold <= 1.0):
        raise RuntimeError("het_threshold must be within [0, 1.0]")
    bed_prefix = str(prefix).strip()
    if bed_prefix.lower().endswith((".bed", ".bim", ".fam")):
        bed_prefix = bed_prefix[:-4]
    if not bed_prefix:
        raise RuntimeError("Sparse GRM BED prefix must not be empty")
    import torch
    from .bed import stage_bed_payload
    packed, n_fam, bim = stage_bed_payload(bed_prefix, mmap_window_mb)      # windowed staging to HBM, used in place
    if n_fam == 0:
        raise RuntimeError("No samples found in BED input.")
    idx, n_sel = _opt_idx(sample_indices)
    if idx is not None and n_sel and (idx.min() < 0 or idx.max() >= n_fam):
        raise RuntimeError("selected sample index out of range for BED logic preparation")
    n_stats = n_sel if (idx is not None and n_sel) else n_fam
    counts = bed_row_counts(packed, n_fam, idx if (idx is not None and n_sel) else None)
    keep, _miss, maf, _std = st.packed_prep_row_stats(counts, n_stats, np.float32(maf_threshold),
                                                      np.float32(max_missing_rate), np.float32(het_threshold))
    if snps_only:
        keep &= snps_only_mask(bim)
    if not keep.any():
        raise RuntimeError("No SNPs left after packed BED filtering. Please relax thresholds.")
    rows = np.nonzero(keep)[0]
    pk = packed if len(rows) == int(packed.shape[0]) else packed[torch.from_numpy(rows).to(packed.device)]
    del packed
    out = _spgrm_packed(pk, n_fam, np.zeros(len(rows), dtype=bool), maf[rows],
                        out_prefix if out_prefix is not None else bed_prefix,
                        idx if (idx is not None and n_sel) else None, method, threshold, abs_threshold, True)
    if progress_callback is not None:
        progress_callback(1, 1)
    return out


def _write_spgrm(path, n, col_ptr, rows, vals):
    """`.spgrm` layout of `write_sparse_grm_csc` (src/stats/spgrm.rs:3745-3767)."""
    nnz = int(len(vals))
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as fh:
        fh.write(np.array([n, nnz], dtype="<u8").tobytes())
        fh.write(np.ascontiguousarray(col_ptr, dtype="<u8").tobytes())
        fh.write(np.ascontiguousarray(rows, dtype="<u4").tobytes())
        fh.write(b"\0" * ((-(4 * nnz)) % 8))
        fh.write(np.ascontiguousarray(vals, dtype="<f8").tobytes())
    os.replace(tmp, path)


def _spgrm_dense_to_jxgrm(k, out_prefix, threshold, abs_threshold, progress_callback):
    import math
    import torch
    from . import pipeline as pl
    if k.ndim != 2 or k.shape[0] != k.shape[1]:
        raise RuntimeError(f"Sparse GRM dense writer expects a square matrix, got shape {tuple(k.shape)}")
    if k.dtype not in (np.float32, np.float64):
        raise RuntimeError(f"Sparse GRM dense writer expects float32 or float64, got {k.dtype}")
    n = int(k.shape[0])
    if n == 0:
        raise RuntimeError("Sparse GRM dense writer requires n_samples > 0")
    if not math.isfinite(threshold):
        raise RuntimeError("Sparse GRM threshold must be finite")
    out_path = _normalize_spgrm_path(out_prefix)
    if not out_path:
        raise RuntimeError("Sparse GRM output prefix must not be empty")
    dev = torch.device("cuda", torch.cuda.current_device())
    ld = int(lib().jxg_num_tiles(n)) * 128
    acc = torch.zeros((ld, ld), dtype=torch.float64, device=dev)
    acc[:n, :n] = torch.from_numpy(np.array(k, copy=True)).to(dev).to(torch.float64)
    work = torch.empty(int(lib().jxg_spgrm_work_bytes(n)), dtype=torch.uint8, device=dev)
    colptr = torch.empty(n + 1, dtype=torch.int64, device=dev)
    st = pl._stream()
    check(lib().jxg_spgrm_count(acc.data_ptr(), n, 1.0, float(threshold), int(bool(abs_threshold)), work.data_ptr(),
                                colptr.data_ptr(), st))
    cp = colptr.cpu().numpy().view(np.uint64)
    nnz = int(cp[n])
    rows = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    vals = torch.empty(max(nnz, 1), dtype=torch.float64, device=dev)
    check(lib().jxg_spgrm_fill(acc.data_ptr(), n, 1.0, float(threshold), int(bool(abs_threshold)), work.data_ptr(),
                               colptr.data_ptr(), rows.data_ptr(), vals.data_ptr(), st))
    _write_spgrm(out_path, n, cp, rows.cpu().numpy()[:nnz].view(np.uint32), vals.cpu().numpy()[:nnz])
    if progress_callback is not None:
        progress_callback(n, n)
    return out_path, n, nnz


def spgrm_dense_npy_to_jxgrm(npy_path, out_prefix, threshold=0.05, abs_threshold=False, progress_callback=None,
                             progress_every=0):
    """src/stats/spgrm.rs:5972-6003 -> `spgrm_dense_npy_to_jxgrm_core` :5103-5199 / `spgrm_dense_f32_to_jxgrm_core`
    :5027-5101: an existing dense GRM (`.npy`, f32 or f64, square) thresholded into `<out_prefix>.spgrm` — lower triangle
    of the stored matrix, values widened to f64, same keep rule, (col, row) order.  The matrix goes to HBM once and the
    count / fill kernels of the sparse GRM build compact it (bit-exact: no arithmetic besides the widening).
    -> (path, n_samples, nnz)."""
    return _spgrm_dense_to_jxgrm(np.load(npy_path, mmap_mode="r"), out_prefix, threshold, abs_threshold, progress_callback)


def spgrm_dense_f32_to_jxgrm(grm, out_prefix, threshold=0.05, abs_threshold=False, progress_callback=None,
                             progress_every=0):
    """src/stats/spgrm.rs:5924-5970 (core :5027-5101): an in-memory dense f32 GRM thresholded into `<out_prefix>.spgrm`
    (the `SparseLMM` API of python/janusx/assoc/api.py:769-780) -> (path, n_samples, nnz)."""
    grm = np.asarray(grm)
    if grm.ndim != 2:
        raise RuntimeError("grm must be 2D (n, n).")
    if grm.shape[0] != grm.shape[1]:
        raise RuntimeError(f"grm must be square, got shape=({grm.shape[0]}, {grm.shape[1]})")
    return _spgrm_dense_to_jxgrm(_c(grm, np.float32), out_prefix, threshold, abs_threshold, progress_callback)


def spgrm_bed_to_jxgrm_from_meta(prefix, row_source_indices, row_flip, row_maf, n_total_sites, out_prefix=None,
                                 sample_indices=None, method=1, threshold=0.05, abs_threshold=False, block_rows=0,
                                 sample_block=0, threads=0, mmap_window_mb=None, progress_callback=None, progress_every=0):
    """src/stats/spgrm.rs:5377-5500: the stream core of `spgrm_bed_to_jxgrm` (:3910-4264) on rows, flips and allele
    frequencies the caller prepared (`jx grm -sparse` after its own QC pass, python/janusx/script/grm.py:1771)
    -> (path, n_samples_used, nnz)."""
    import torch
    from .bed import stage_bed_payload
    bed_prefix = _bed_prefix(prefix)
    if int(n_total_sites) <= 0:
        raise RuntimeError("n_total_sites must be positive for sparse GRM meta route.")
    src = _c(row_source_indices, np.int64).ravel()
    flip = np.asarray(row_flip).astype(bool).ravel()
    maf = _c(row_maf, np.float32).ravel()
    if src.size == 0:
        raise RuntimeError("row_source_indices must not be empty")
    if (src < 0).any():
        raise RuntimeError(f"row_source_indices must be non-negative, got {int(src[src < 0][0])}")
    if flip.shape[0] != src.shape[0] or maf.shape[0] != src.shape[0]:
        raise RuntimeError(f"row meta length mismatch: row_source_indices={src.shape[0]}, row_flip={flip.shape[0]}, "
                           f"row_maf={maf.shape[0]}")
    if int(src.max()) >= int(n_total_sites):
        raise RuntimeError(f"row_source index out of range: {int(src[src >= int(n_total_sites)][0])} >= "
                           f"n_total_sites={int(n_total_sites)}")
    packed, n_fam, _bim = stage_bed_payload(bed_prefix, mmap_window_mb)
    if n_fam == 0:
        raise RuntimeError("No samples found in BED input.")
    if int(src.max()) >= int(packed.shape[0]):
        raise RuntimeError(f"row_source index out of range for the BED payload: {int(src.max())} >= {int(packed.shape[0])}")
    idx, n_sel = _opt_idx(sample_indices)
    pk = packed[torch.from_numpy(src).to(packed.device)]
    del packed
    out = _spgrm_packed(pk, n_fam, flip, maf, out_prefix if out_prefix is not None else bed_prefix,
                        idx if (idx is not None and n_sel) else None, method, threshold, abs_threshold, True)
    if progress_callback is not None:
        progress_callback(1, 1)
    return out


def load_spgrm(path):
    """Reader of the `.spgrm` layout (`write_sparse_grm_csc`, src/stats/spgrm.rs:3745-3767)
    -> (n, col_ptr u64 (n+1), row_indices u32 (nnz), values f64 (nnz))."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size < 16:
        raise RuntimeError(f"sparse GRM file too small: {path}")
    n, nnz = (int(v) for v in raw[:16].view("<u8"))
    at = 16
    need = at + 8 * (n + 1) + 4 * nnz + ((-(4 * nnz)) % 8) + 8 * nnz
    if raw.size != need:
        raise RuntimeError(f"sparse GRM file length mismatch: got {raw.size}, expected {need}")
    col_ptr = raw[at:at + 8 * (n + 1)].view("<u8").copy()
    at += 8 * (n + 1)
    rows = raw[at:at + 4 * nnz].view("<u4").copy()
    at += 4 * nnz + ((-(4 * nnz)) % 8)
    vals = raw[at:at + 8 * nnz].view("<f8").copy()
    return n, col_ptr, rows, vals


# ------------------------------------------------------------------------------------------------
# Sparse REML null model over a `.spgrm` (src/stats/spreml.rs)
# ------------------------------------------------------------------------------------------------

def _brent_minimize_with_init(f, low, high, tol, max_iter, init_x):
    """Host Brent of src/math/brent.rs (`brent_minimize_with_init`; the device twin is `brent_reml` in csrc/k_scan.hip):
    golden-section / parabolic steps, `e` kept on parabolic steps, start at init_x when it lies inside [low, high]."""
    import math
    a, c = (low, high) if low < high else (high, low)
    eps = float(np.finfo(np.float64).eps)
    tol = max(abs(tol), 1e-12)
    x = init_x if (init_x is not None and math.isfinite(init_x) and a <= init_x <= c) else 0.5 * (a + c)
    w = v = x
    fx = f(x)
    fw = fv = fx
    d = e = 0.0
    for _ in range(int(max_iter)):
        m = 0.5 * (a + c)
        tol1 = tol * abs(x) + eps
        tol2 = 2.0 * tol1
        if abs(x - m) <= tol2 - 0.5 * (c - a):
            break
        parabolic = False
        if abs(e) > tol1:
            pp = (x - v) * ((x - w) * (fx - fv)) - (x - w) * ((x - v) * (fx - fw))
            qq = 2.0 * (((x - v) * (fx - fw)) - ((x - w) * (fx - fv)))
            if qq > 0.0:
                pp = -pp
            else:
                qq = -qq
            if abs(qq) > eps:
                st = pp / qq
                u = x + st
                if (u - a) >= tol2 and (c - u) >= tol2 and abs(st) < 0.5 * abs(e):
                    d = st
                    if (u - a) < tol2 or (c - u) < tol2:
                        d = tol1 if x < m else -tol1
                    parabolic = True
        if not parabolic:
            e = (c - x) if x < m else (a - x)
            d = 0.3819660 * e
        if abs(d) < tol1:
            d = tol1 if d >= 0.0 else -tol1
        u = x + d
        fu = f(u)
        if fu <= fx:
            if u >= x:
                a = x
            else:
                c = x
            v, fv = w, fw
            w, fw = x, fx
            x, fx = u, fu
        else:
            if u >= x:
                c = u
            else:
                a = u
            if fu <= fw or w == x:
                v, fv = w, fw
                w, fw = u, fu
            elif fu <= fv or v == x or v == w:
                v, fv = u, fu
    return x, fx


def _spgrm_dense_device(path, sample_indices):
    """Dense symmetric image (HBM, f64) of a `.spgrm`, optionally K[idx][:, idx] in the given order
    (`subset_sparse_grm_csc`, src/math/cholesky.rs:618-690, + `sparse_grm_to_dense`, src/stats/splmm.rs:2024-2039).
    -> (tensor (n, n), idx or None)."""
    import torch
    from . import pipeline as pl
    n_all, col_ptr, rows, vals = load_spgrm(path)
    idx, n_sel = _opt_idx(sample_indices)
    if idx is not None:
        if n_sel == 0:
            raise RuntimeError("Sparse GRM subset requires at least one sample")
        if idx.min() < 0 or idx.max() >= n_all:
            raise RuntimeError(f"Sparse GRM subset index out of range for n_samples={n_all}")
        uniq, cnt = np.unique(idx, return_counts=True)
        if (cnt > 1).any():
            first = next(int(v) for v in idx if cnt[np.searchsorted(uniq, v)] > 1)
            raise RuntimeError(f"Sparse GRM subset contains duplicated sample index: {first}")
    n = n_sel if idx is not None else n_all
    if n == 0:
        raise RuntimeError("SPREML requires n > 0")
    dev = torch.device("cuda", torch.cuda.current_device())
    d_cp = torch.from_numpy(col_ptr.view(np.int64)).to(dev)
    d_ri = torch.from_numpy(rows.view(np.int32)).to(dev)
    d_va = torch.from_numpy(vals).to(dev)
    d_map = None
    if idx is not None:
        mp = np.full(n_all, -1, dtype=np.int32)
        mp[idx] = np.arange(n, dtype=np.int32)
        d_map = torch.from_numpy(mp).to(dev)
    k = torch.empty((n, n), dtype=torch.float64, device=dev)
    check(lib().jxg_spgrm_densify(d_cp.data_ptr(), d_ri.data_ptr(), d_va.data_ptr(), int(n_all),
                                  d_map.data_ptr() if d_map is not None else None, n, k.data_ptr(), pl._stream()))
    return k, idx


def splmm_sparse_grm_diag_stats(jxgrm_path, sample_indices=None):
    """src/stats/splmm.rs:4055-4111 (`sparse_diag_stats` :1978-2022) -> (mean |diag| floored at 1e-30, min diag, max
    diag [starting from 0], n, nnz) of the (subset of the) sparse GRM; host-side metadata pass over the CSC arrays."""
    n_all, col_ptr, rows, vals = load_spgrm(jxgrm_path)
    idx, n_sel = _opt_idx(sample_indices)
    cp = col_ptr.astype(np.int64)
    cols = np.repeat(np.arange(n_all, dtype=np.int64), np.diff(cp))
    if n_all == 0:
        raise RuntimeError("SparseLMM diagonal stats require n_samples > 0")
    if idx is not None and not (n_sel == n_all and np.array_equal(idx, np.arange(n_all))):
        if n_sel == 0:
            raise RuntimeError("Sparse GRM subset requires at least one sample")
        if idx.min() < 0 or idx.max() >= n_all:
            raise RuntimeError(f"Sparse GRM subset index out of range for n_samples={n_all}")
        if len(np.unique(idx)) != n_sel:
            dup = next(int(v) for k, v in enumerate(idx) if v in idx[:k])
            raise RuntimeError(f"Sparse GRM subset contains duplicated sample index: {dup}")
        inside = np.zeros(n_all, dtype=bool)
        inside[idx] = True
        sel = inside[rows.astype(np.int64)] & inside[cols]
        rows, cols, vals = rows[sel], cols[sel], vals[sel]
        n_out = n_sel
        want = np.sort(idx)
    else:
        n_out = n_all
        want = np.arange(n_all, dtype=np.int64)
    on_diag = rows.astype(np.int64) == cols
    dcols, dvals = cols[on_diag], vals[on_diag]
    first = np.unique(dcols, return_index=True)[1]                  # the first diagonal entry of every column counts
    dcols, dvals = dcols[first], dvals[first]
    missing = np.setdiff1d(want, dcols)
    if missing.size:
        raise RuntimeError(f"SparseLMM diagonal is missing at column {int(missing[0])}")
    if not np.isfinite(dvals).all():
        bad = int(dcols[np.nonzero(~np.isfinite(dvals))[0][0]])
        raise RuntimeError(f"SparseLMM diagonal contains non-finite value at column {bad}")
    mean_abs = max(float(np.abs(dvals).sum()) / float(n_out), 1e-30)
    return mean_abs, float(dvals.min()), float(max(dvals.max(), 0.0)), int(n_out), int(len(vals))


def splmm_load_sparse_grm_subset_dense(jxgrm_path, sample_indices=None):
    """src/stats/splmm.rs:4022-4054 -> dense (n, n) f64 image of the (subset of the) sparse GRM."""
    k, _ = _spgrm_dense_device(jxgrm_path, sample_indices)
    return k.cpu().numpy()


def _sparse_block_size():
    """Samples per diagonal block of the block route (JXGPU_SPLMM_BLOCK, default 4096)."""
    v = os.environ.get("JXGPU_SPLMM_BLOCK", "")
    return max(int(v), 2) if v.strip() else 4096


def _sparse_block_route(n):
    """Block-diagonal spectral form (connected components of the sparse GRM) instead of one dense n x n eigenproblem:
    from n = 16384 (JXGPU_SPLMM_BLOCK_MIN_N), or always / never with JXGPU_SPLMM_ROUTE=block / dense."""
    mode = os.environ.get("JXGPU_SPLMM_ROUTE", "").strip().lower()
    if mode == "block":
        return True
    if mode == "dense":
        return False
    v = os.environ.get("JXGPU_SPLMM_BLOCK_MIN_N", "")
    return int(n) >= (int(v) if v.strip() else 16384)


def _pack_components_into_blocks(lab, bsz):
    """Sample order of the block-diagonal route: `lab[i]` = connected component of sample i.  Whole components are packed
    (largest first, first fit among the most recent blocks) into blocks of at most `bsz` samples; a component of `bsz` or
    more samples is a block of its own.  -> (perm, offs): perm[pos] = sample at position pos of the block order (components
    contiguous inside a block), block b = positions offs[b] .. offs[b + 1]."""
    lab = np.asarray(lab, dtype=np.int64)
    ncomp = int(lab.max()) + 1 if lab.size else 0
    sizes = np.bincount(lab, minlength=ncomp)
    order = np.argsort(-sizes, kind="stable")
    block_of = np.empty(ncomp, dtype=np.int64)
    fill = []
    for c in order:
        placed = False
        if sizes[c] < bsz:
            for b in range(len(fill) - 1, max(len(fill) - 64, -1), -1):
                if fill[b] + sizes[c] <= bsz:
                    fill[b] += int(sizes[c])
                    block_of[c] = b
                    placed = True
                    break
        if not placed:
            fill.append(int(sizes[c]))
            block_of[c] = len(fill) - 1
    sample_block = block_of[lab]
    perm = np.lexsort((lab, sample_block)).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(np.bincount(sample_block, minlength=len(fill)))]).astype(np.int64)
    return perm, offs


def _hbm_free_agreed():
    """Free HBM in bytes.  When the caller has declared the sparse routes collective (`dist.enable_collective_size_checks()`:
    every rank makes the same calls) the MINIMUM over the ranks, so that a size check decides the same way on every rank (a rank
    that raised alone would leave the others waiting in their next collective).  Otherwise the local figure: these routes may be
    called on one rank only, or be answered from the spectral cache on some ranks, and a hidden collective would hang the rest."""
    import torch
    from . import dist as _jd
    free, _total = torch.cuda.mem_get_info()
    try:
        import torch.distributed as tdist
        if (_jd.collective_size_checks() and tdist.is_available() and tdist.is_initialized()
                and tdist.get_world_size() > 1):
            t = torch.tensor([float(free)], dtype=torch.float64)
            if tdist.get_backend() == "nccl":
                t = t.to(torch.device("cuda", torch.cuda.current_device()))
            tdist.all_reduce(t, op=tdist.ReduceOp.MIN)
            free = int(t.item())
    except Exception:   # noqa: BLE001 - no usable process group: the local figure
        pass
    return int(free)


def sparse_component_limit(free_bytes=None):
    """Largest connected component (samples) of a thresholded GRM the spectral SparseLMM / sparse-REML routes take: one dense
    eigenproblem per component on ONE GPU -- the f64 image of the component, its eigenvectors and the eigensolver's workspace,
    about 5 n^2 doubles -- so n <= sqrt(free HBM / 40 B): ~ 79 000 samples on an empty MI355X (288 GB; BASELINE configs[3] runs
    n = 50 000 in 16 s).  The reference factorises K + lambda I sparsely for any structure (src/math/cholesky.rs:776-1075,
    src/stats/spreml.rs:384-760); beyond this limit the routes here refuse with the size and the limit in the message.
    JXGPU_SPLMM_COMPONENT_MAX overrides (tests)."""
    import math
    v = os.environ.get("JXGPU_SPLMM_COMPONENT_MAX", "").strip()
    if v:
        return max(int(v), 1)
    free = _hbm_free_agreed() if free_bytes is None else int(free_bytes)
    return int(math.isqrt(max(free, 0) // 40))


class _ComponentTooLarge(RuntimeError):
    """A dense eigenproblem the spectral routes would need does not fit this GPU: the caller switches to the sparse-factor
    route (`_SparseFactorReml`) or, where none exists, lets the message through."""


def _check_spectral_sparse_size(n, what="the selected samples"):
    """The single-eigenproblem form of the sparse REML / SparseLMM routes holds a dense f64 image of K, its eigenvectors
    and the eigensolver's workspace (~5 n^2 doubles) in HBM.  Refuse clearly -- on every rank alike -- instead of failing
    inside hipMalloc."""
    limit = sparse_component_limit()
    if int(n) > limit:
        raise _ComponentTooLarge(
            f"sparse-GRM spectral route: {what} form one dense eigenproblem of {int(n)} samples; the limit on this GPU is "
            f"{limit} samples (about 5 n^2 doubles of HBM: the f64 image of K, its eigenvectors and the eigensolver's workspace). "
            "Raise the GRM cut-off so that the relatedness graph falls apart into smaller components, or restrict the samples; "
            "the reference's sparse factorisation of K + lambda I for arbitrary structure (src/math/cholesky.rs) is not "
            "rebuilt here")


# One-entry cache of the spectral form of a sparse GRM (eigenvalues + eigenvectors of the dense image or of its diagonal
# blocks, in HBM): the reference's workflow calls the null fit and then the scan of a trait as separate functions, each of
# which factorises the same K; here the second call (and every further trait on the same samples) reuses the decomposition.
# Keyed by the file's identity (path, size, mtime), the sample list and the route; `spectral_cache_clear()` frees the HBM.
_SPECTRAL_CACHE = {}


def _spectral_cache_key(path, sample_indices, route):
    import hashlib
    if os.environ.get("JXGPU_SPECTRAL_CACHE", "1").strip() == "0":
        return None
    try:
        st = os.stat(path)
    except OSError:
        return None
    idx, _n = _opt_idx(sample_indices)
    h = "all" if idx is None else hashlib.sha1(np.ascontiguousarray(idx, dtype=np.int64).tobytes()).hexdigest()
    return (os.path.abspath(path), st.st_size, st.st_mtime_ns, h, route)


def _spectral_cache_put(key, value):
    if key is None:
        return
    _SPECTRAL_CACHE.clear()
    _SPECTRAL_CACHE[key] = value


def spectral_cache_clear():
    """Drop the cached spectral form of the last sparse GRM (frees its eigenvector blocks in HBM)."""
    _SPECTRAL_CACHE.clear()


def release_device_scratch():
    """Hand everything this library keeps in HBM between calls back to the driver: the cached spectral form of the last sparse
    GRM and the eigensolver's kept workspaces (`jxg_scratch_trim`; at most 6 GB each).  For a long-lived process between two
    problems of very different size; returns the bytes of workspace released.  (The reference's CPU path frees per call.)"""
    _SPECTRAL_CACHE.clear()
    return int(lib().jxg_scratch_trim())


class _SpectralSparseReml:
    """K + lambda I of a (subset of a) sparse GRM handled through ONE eigendecomposition on the GPU instead of one
    sparse LLT per lambda (src/stats/spreml.rs:384-512 factorises at every evaluation): K = U diag(s) U', so
    (K + lambda I)^-1 = U diag(1 / (s + lambda)) U', log det = sum log(s + lambda), and the matrix is factorisable
    exactly when min(s) + lambda > 0.  Densify (jxg_spgrm_densify), eigh (jxg_eigh_f64) and the rotation of [y | X]
    (rocBLAS dgemm through torch) run on the device; an evaluation is then O(n p^2) on (n, p + 2) numbers."""

    def __init__(self, path, y, x_cov, sample_indices):
        import torch
        from . import pipeline as pl
        y = _c(y, np.float64).ravel()
        self.perm, self.blocks = None, None
        if _sparse_block_route(y.shape[0]):
            self._init_blocks(path, y, x_cov, sample_indices)
            return
        key = _spectral_cache_key(path, sample_indices, "dense")
        hit = _SPECTRAL_CACHE.get(key)
        if hit is None:
            _check_spectral_sparse_size(y.shape[0])
            k, idx = _spgrm_dense_device(path, sample_indices)
            n = int(k.shape[0])
        else:
            n, idx = int(hit[0].shape[0]), hit[2]
        if n != y.shape[0]:
            raise RuntimeError(f"SPREML subset sample size mismatch: sparse n={n}, phenotype n={y.shape[0]}")
        if n == 0:
            raise RuntimeError("SPREML requires n > 0")
        if x_cov is None:
            x = np.ones((n, 1), dtype=np.float64)
        else:
            xc = _c(x_cov, np.float64)
            if xc.ndim != 2 or xc.shape[0] != n:
                raise RuntimeError(f"x_cov shape mismatch: got {list(xc.shape)}, expected ({n}, p)")
            x = np.concatenate([np.ones((n, 1)), xc], axis=1)
        self.n, self.p = n, int(x.shape[1])
        if hit is None:
            s, ut = pl.eigh_from_grm(k, ridge=0.0)               # row j of ut = eigenvector j
            del k
            _spectral_cache_put(key, (s, ut, idx))
        else:
            s, ut = hit[0], hit[1]
        dev = s.device
        rot = ut @ torch.from_numpy(np.concatenate([y[:, None], x], axis=1)).to(dev)
        rot = rot.cpu().numpy()
        self.s = s.cpu().numpy()
        self.yr, self.xr = rot[:, 0].copy(), rot[:, 1:].copy()
        self.smin = float(self.s.min())
        self.s_dev, self.ut_dev, self.x_design, self.y_raw, self.sample_idx = s, ut, x, y, idx

    def _init_blocks(self, path, y, x_cov, sample_indices):
        """Block-diagonal form: the graph of the (subset of the) sparse GRM is split into connected components, whole
        components are packed into diagonal blocks of at most `_sparse_block_size()` samples (a larger component is its own
        block), every block is densified and eigendecomposed on the device on its own.  The model then lives in the BLOCK
        ORDER of the samples (`self.perm`: position -> index into the caller's sample list); likelihood evaluations only
        see (s, U'y, U'X) and are unchanged."""
        import torch
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        from . import pipeline as pl
        n_all, col_ptr, rows, vals = load_spgrm(path)
        idx, n_sel = _opt_idx(sample_indices)
        if idx is not None:
            if n_sel == 0:
                raise RuntimeError("Sparse GRM subset requires at least one sample")
            if idx.min() < 0 or idx.max() >= n_all:
                raise RuntimeError(f"Sparse GRM subset index out of range for n_samples={n_all}")
            if len(np.unique(idx)) != n_sel:
                raise RuntimeError("Sparse GRM subset contains duplicated sample index")
        n = n_sel if idx is not None else n_all
        if n != y.shape[0]:
            raise RuntimeError(f"SPREML subset sample size mismatch: sparse n={n}, phenotype n={y.shape[0]}")
        if x_cov is None:
            x = np.ones((n, 1), dtype=np.float64)
        else:
            xc = _c(x_cov, np.float64)
            if xc.ndim != 2 or xc.shape[0] != n:
                raise RuntimeError(f"x_cov shape mismatch: got {list(xc.shape)}, expected ({n}, p)")
            x = np.concatenate([np.ones((n, 1)), xc], axis=1)
        self.n, self.p = n, int(x.shape[1])
        key = _spectral_cache_key(path, sample_indices, f"block{_sparse_block_size()}")
        hit = _SPECTRAL_CACHE.get(key)
        if hit is not None:
            # the eigendecompositions of the diagonal blocks depend on the GRM and the sample list only: a null fit followed by
            # a scan of the same trait (or the next trait on the same samples) reuses them and only rotates its own [y | X]
            perm, blocks, s_all, glob = hit
            dev = blocks[0][2].device
            yx = np.concatenate([y[:, None], x], axis=1)[perm]
            rot_all = np.empty((n, 1 + self.p), dtype=np.float64)
            for o0, nb, utb in blocks:
                rot_all[o0:o0 + nb] = (utb @ torch.from_numpy(yx[o0:o0 + nb]).to(dev)).cpu().numpy()
            self.s = s_all
            self.yr, self.xr = rot_all[:, 0].copy(), rot_all[:, 1:].copy()
            self.smin = float(self.s.min())
            self.perm, self.blocks = perm, blocks
            self.s_dev, self.ut_dev = torch.from_numpy(s_all).to(dev), None
            self.x_design, self.y_raw = x[perm], y[perm]
            self.sample_idx = glob
            return
        # connected components of the selected sub-graph (host: one pass over the nnz entries)
        cp = col_ptr.astype(np.int64)
        cols = np.repeat(np.arange(n_all, dtype=np.int64), np.diff(cp))
        r64 = rows.astype(np.int64)
        local = np.arange(n_all, dtype=np.int64)
        if idx is not None:
            local = np.full(n_all, -1, dtype=np.int64)
            local[idx] = np.arange(n, dtype=np.int64)
        lr, lc = local[r64], local[cols]
        ok = (lr >= 0) & (lc >= 0) & (lr != lc) & (vals != 0.0)
        graph = coo_matrix((np.ones(int(ok.sum()), dtype=np.int8), (lr[ok], lc[ok])), shape=(n, n))
        ncomp, lab = connected_components(graph, directed=False)
        sizes = np.bincount(lab, minlength=ncomp)
        bsz = _sparse_block_size()
        # a component beyond the block size is a block of its own, up to what one GPU's eigensolver holds (the same decision on
        # every rank once the caller has declared the call collective: dist.enable_collective_size_checks)
        if int(sizes.max()) > bsz:
            _check_spectral_sparse_size(int(sizes.max()), what="the samples of the largest connected component of the sparse GRM")
        perm, offs = _pack_components_into_blocks(lab, bsz)
        fill = np.diff(offs)
        dev = torch.device("cuda", torch.cuda.current_device())
        d_cp = torch.from_numpy(col_ptr.view(np.int64)).to(dev)
        d_ri = torch.from_numpy(rows.view(np.int32)).to(dev)
        d_va = torch.from_numpy(vals).to(dev)
        glob = perm if idx is None else idx[perm]                        # sparse-GRM index of every position of the block order
        yx = np.concatenate([y[:, None], x], axis=1)[perm]
        s_all = np.empty(n, dtype=np.float64)
        rot_all = np.empty((n, 1 + self.p), dtype=np.float64)
        blocks = []
        # several ranks: the blocks are dealt over the ranks (largest first, each to the rank with the least n^3 so far), a rank
        # decomposes its own with the eigensolver's distribution paused (jxg_eigh_set_local) and the owner's (s, U) go to
        # everybody afterwards: same bits on every rank, the decompositions of the null fit in 1 / world of the time
        rank, world = pl.dist_info()
        owner = np.zeros(len(fill), dtype=np.int64)
        if world > 1:
            load = np.zeros(world)
            for b in np.argsort(-fill, kind="stable"):
                owner[b] = int(np.argmin(load))
                load[owner[b]] += float(fill[b]) ** 3
            check(lib().jxg_eigh_set_local(1))
        try:
            for b in range(len(fill)):
                o0, o1 = int(offs[b]), int(offs[b + 1])
                nb = o1 - o0
                if owner[b] != rank:
                    blocks.append((o0, nb, torch.empty((nb, nb), dtype=torch.float64, device=dev)))
                    continue
                mp = np.full(n_all, -1, dtype=np.int32)
                mp[glob[o0:o1]] = np.arange(nb, dtype=np.int32)
                d_map = torch.from_numpy(mp).to(dev)
                kb = torch.empty((nb, nb), dtype=torch.float64, device=dev)
                check(lib().jxg_spgrm_densify(d_cp.data_ptr(), d_ri.data_ptr(), d_va.data_ptr(), int(n_all), d_map.data_ptr(), nb,
                                              kb.data_ptr(), pl._stream()))
                if nb == 1:                                  # a lone sample: its own eigenpair
                    sb, utb = kb.reshape(1).clone(), torch.ones((1, 1), dtype=torch.float64, device=dev)
                else:
                    sb, utb = pl.eigh_from_grm(kb, ridge=0.0)
                del kb
                s_all[o0:o1] = sb.cpu().numpy()
                blocks.append((o0, nb, utb))
        finally:
            if world > 1:
                lib().jxg_eigh_set_local(0)
        if world > 1:
            import torch.distributed as dist
            through_host = dist.get_backend() != "nccl"        # functional mode on shared GPUs (gloo): through host memory
            s_t = torch.from_numpy(s_all)
            for b, (o0, nb, utb) in enumerate(blocks):
                src = int(owner[b])
                if through_host:
                    h = utb.cpu()
                    dist.broadcast(h, src=src)
                    if src != rank:
                        utb.copy_(h)
                    dist.broadcast(s_t[o0:o0 + nb], src=src)
                else:
                    dist.broadcast(utb, src=src)
                    sd = s_t[o0:o0 + nb].to(dev)
                    dist.broadcast(sd, src=src)
                    s_t[o0:o0 + nb] = sd.cpu()
        for o0, nb, utb in blocks:
            rot_all[o0:o0 + nb] = (utb @ torch.from_numpy(yx[o0:o0 + nb]).to(dev)).cpu().numpy()
        self.s = s_all
        self.yr, self.xr = rot_all[:, 0].copy(), rot_all[:, 1:].copy()
        self.smin = float(self.s.min())
        self.perm, self.blocks = perm, blocks
        self.s_dev, self.ut_dev = torch.from_numpy(s_all).to(dev), None
        self.x_design, self.y_raw = x[perm], y[perm]
        self.sample_idx = glob
        _spectral_cache_put(key, (perm, blocks, s_all, glob))

    def factorizable(self, lam):
        import math
        return math.isfinite(lam) and lam > 0.0 and (self.smin + lam) > 0.0

    def evaluate(self, log10_lambda, vp_fixed=None):
        import math
        lam = 10.0 ** log10_lambda
        if not (math.isfinite(lam) and lam > 0.0):
            raise RuntimeError(f"SPREML lambda is invalid at log10(lambda)={log10_lambda}")
        n, p = self.n, self.p
        if p == 0 or n <= p:
            raise RuntimeError(f"SPREML requires n > p, got n={n}, p={p}")
        d = self.s + lam
        if not (d.min() > 0.0):
            raise RuntimeError(f"K + lambda I is not positive definite at lambda={lam} (min eigenvalue {d.min():.3e})")
        wy = self.yr / d
        y_vinv_y = float(self.yr @ wy)
        xt_vinv_y = self.xr.T @ wy
        xt_vinv_x = self.xr.T @ (self.xr / d[:, None])
        chol = _spd_cholesky_with_jitter(xt_vinv_x, "SPREML XtVinvX")
        beta = np.linalg.solve(chol.T, np.linalg.solve(chol, xt_vinv_y))
        ypy = y_vinv_y - float(xt_vinv_y @ beta)
        if not math.isfinite(ypy) or ypy <= 1e-30:
            raise RuntimeError(f"SPREML profiled residual quadratic form is invalid at lambda={lam}: yPy={ypy}")
        df = float(n - p)
        log_det_m = float(np.log(d).sum())
        log_det_x = 2.0 * float(np.log(np.diag(chol)).sum())
        if vp_fixed is None:
            sigma_g2 = ypy / df
            if not math.isfinite(sigma_g2) or sigma_g2 <= 0.0:
                raise RuntimeError(f"SPREML sigma_g2 is invalid at lambda={lam}: sigma_g2={sigma_g2}")
            sigma_e2 = lam * sigma_g2
            reml = df * (math.log(df) - 1.0 - math.log(2.0 * math.pi)) * 0.5 - 0.5 * (
                df * math.log(ypy) + log_det_m + log_det_x)
            nf = float(n)
            ml = nf * (math.log(nf) - 1.0 - math.log(2.0 * math.pi)) * 0.5 - 0.5 * (nf * math.log(ypy) + log_det_m)
        else:
            if not (math.isfinite(vp_fixed) and vp_fixed > 0.0):
                raise RuntimeError(f"SPREML fastGWA fixed-Vp objective requires finite vp_fixed > 0, got {vp_fixed}")
            sigma_g2 = vp_fixed / (1.0 + lam)
            sigma_e2 = lam * sigma_g2
            reml = -0.5 * (df * math.log(sigma_g2) + log_det_m + log_det_x + ypy / sigma_g2)
            ml = float("nan")
        if not math.isfinite(reml) or not (math.isfinite(ml) or math.isnan(ml)):
            raise RuntimeError(f"SPREML likelihood is invalid at lambda={lam}: ml={ml}, reml={reml}")
        return (log10_lambda, lam, sigma_g2, sigma_e2, ml, reml)


class _SparseFactorReml:
    """K + lambda I of a (subset of a) sparse GRM whose relatedness graph holds a connected component beyond one dense
    eigenproblem on this GPU (`sparse_component_limit()`): no spectral form exists, so the matrix is handled as the reference
    handles every sparse GRM -- a sparse factorisation of K + lambda I per lambda ON THE HOST (`SparseJxgrmCholesky`,
    src/math/cholesky.rs:733, 1018-1183; one per objective evaluation, src/stats/spreml.rs:384-512): scipy's SuperLU in symmetric
    mode with a minimum-degree ordering of A + A' gives log det(K + lambda I) exactly (sum of the logs of U's diagonal; positive
    definite <=> every pivot positive) and the few solves of the null model (V^-1 [y | X]).  The per-SNP solves of the exact scan
    run on the device (`jxg_sps_solve_multi`: one multi-vector CG per block of SNP rows over the CSR image of K,
    csrc/k_spsolve.hip); the GRAMMAR-gamma route needs V^-1 for the <= 2000 sampled markers only (host factor) and scans in
    sample space as always.  Same interface as `_SpectralSparseReml` for the likelihood searches (`evaluate`, `factorizable`)."""

    factor_route = True

    def __init__(self, path, y, x_cov, sample_indices):
        import scipy.sparse as sp
        y = _c(y, np.float64).ravel()
        n_all, col_ptr, rows, vals = load_spgrm(path)
        idx, n_sel = _opt_idx(sample_indices)
        low = sp.csc_matrix((vals, rows.astype(np.int64), col_ptr.astype(np.int64)), shape=(n_all, n_all))
        full = low + sp.tril(low, -1).T                       # the file holds the lower triangle, diagonal first in every column
        if idx is not None:
            if n_sel == 0:
                raise RuntimeError("Sparse GRM subset requires at least one sample")
            if idx.min() < 0 or idx.max() >= n_all:
                raise RuntimeError(f"Sparse GRM subset index out of range for n_samples={n_all}")
            if len(np.unique(idx)) != n_sel:
                raise RuntimeError("Sparse GRM subset contains duplicated sample index")
            full = full.tocsr()[idx][:, idx]
        self.k = full.tocsr()
        self.k.sort_indices()
        n = int(self.k.shape[0])
        if n != y.shape[0]:
            raise RuntimeError(f"SPREML subset sample size mismatch: sparse n={n}, phenotype n={y.shape[0]}")
        if n == 0:
            raise RuntimeError("SPREML requires n > 0")
        if x_cov is None:
            x = np.ones((n, 1), dtype=np.float64)
        else:
            xc = _c(x_cov, np.float64)
            if xc.ndim != 2 or xc.shape[0] != n:
                raise RuntimeError(f"x_cov shape mismatch: got {list(xc.shape)}, expected ({n}, p)")
            x = np.concatenate([np.ones((n, 1)), xc], axis=1)
        self.n, self.p = n, int(x.shape[1])
        self.x_design, self.y_raw, self.sample_idx = x, y, idx
        self.perm, self.blocks = None, None
        self.diag = np.asarray(self.k.diagonal(), dtype=np.float64)
        self._fac = (None, None)         # (lambda, SuperLU) of the last factorisation
        self._dev = None                 # CSR image in HBM (exact scan)

    def _factor(self, lam):
        """SuperLU of K + lambda I without row pivoting, or None when a pivot is not positive (not positive definite)."""
        import scipy.sparse as sp
        from scipy.sparse.linalg import splu
        if self._fac[0] == lam:
            return self._fac[1]
        a = (self.k + lam * sp.identity(self.n, format="csr")).tocsc()
        try:
            lu = splu(a, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
        except RuntimeError:             # exactly singular
            lu = None
        if lu is not None:
            d = lu.U.diagonal()
            if not (np.all(np.isfinite(d)) and np.all(d > 0.0) and np.array_equal(lu.perm_r, lu.perm_c)):
                lu = None                # an indefinite matrix shows as a non-positive pivot or as a row interchange
        self._fac = (lam, lu)
        return lu

    def factorizable(self, lam):
        import math
        return bool(math.isfinite(lam) and lam > 0.0 and self._factor(float(lam)) is not None)

    @property
    def smin(self):
        """Sign of the smallest eigenvalue of K (the spectral form's `smin` is only ever compared with 0 by the callers)."""
        return 1.0 if self._factor(0.0) is not None else -1.0

    def solve(self, lam, b):
        """(K + lambda I)^-1 b for a vector or an (n, k) block (host sparse factor)."""
        lu = self._factor(float(lam))
        if lu is None:
            raise RuntimeError(f"K + lambda I is not positive definite at lambda={lam}")
        return lu.solve(np.ascontiguousarray(b, dtype=np.float64))

    def evaluate(self, log10_lambda, vp_fixed=None):
        """`sparse_reml_evaluate` (src/stats/spreml.rs:384-512): the same formulas as the spectral form, V^-1 and log det V from
        the sparse factor."""
        import math
        lam = 10.0 ** log10_lambda
        if not (math.isfinite(lam) and lam > 0.0):
            raise RuntimeError(f"SPREML lambda is invalid at log10(lambda)={log10_lambda}")
        n, p = self.n, self.p
        if p == 0 or n <= p:
            raise RuntimeError(f"SPREML requires n > p, got n={n}, p={p}")
        lu = self._factor(lam)
        if lu is None:
            raise RuntimeError(f"K + lambda I is not positive definite at lambda={lam}")
        vi = lu.solve(np.concatenate([self.y_raw[:, None], self.x_design], axis=1))
        y_vinv_y = float(self.y_raw @ vi[:, 0])
        xt_vinv_y = self.x_design.T @ vi[:, 0]
        xt_vinv_x = self.x_design.T @ vi[:, 1:]
        xt_vinv_x = 0.5 * (xt_vinv_x + xt_vinv_x.T)
        chol = _spd_cholesky_with_jitter(xt_vinv_x, "SPREML XtVinvX")
        beta = np.linalg.solve(chol.T, np.linalg.solve(chol, xt_vinv_y))
        ypy = y_vinv_y - float(xt_vinv_y @ beta)
        if not math.isfinite(ypy) or ypy <= 1e-30:
            raise RuntimeError(f"SPREML profiled residual quadratic form is invalid at lambda={lam}: yPy={ypy}")
        df = float(n - p)
        log_det_m = float(np.log(lu.U.diagonal()).sum())          # L has a unit diagonal, no interchanges: det = prod U_ii
        log_det_x = 2.0 * float(np.log(np.diag(chol)).sum())
        if vp_fixed is None:
            sigma_g2 = ypy / df
            if not math.isfinite(sigma_g2) or sigma_g2 <= 0.0:
                raise RuntimeError(f"SPREML sigma_g2 is invalid at lambda={lam}: sigma_g2={sigma_g2}")
            sigma_e2 = lam * sigma_g2
            reml = df * (math.log(df) - 1.0 - math.log(2.0 * math.pi)) * 0.5 - 0.5 * (
                df * math.log(ypy) + log_det_m + log_det_x)
            nf = float(n)
            ml = nf * (math.log(nf) - 1.0 - math.log(2.0 * math.pi)) * 0.5 - 0.5 * (nf * math.log(ypy) + log_det_m)
        else:
            if not (math.isfinite(vp_fixed) and vp_fixed > 0.0):
                raise RuntimeError(f"SPREML fastGWA fixed-Vp objective requires finite vp_fixed > 0, got {vp_fixed}")
            sigma_g2 = vp_fixed / (1.0 + lam)
            sigma_e2 = lam * sigma_g2
            reml = -0.5 * (df * math.log(sigma_g2) + log_det_m + log_det_x + ypy / sigma_g2)
            ml = float("nan")
        if not math.isfinite(reml) or not (math.isfinite(ml) or math.isnan(ml)):
            raise RuntimeError(f"SPREML likelihood is invalid at lambda={lam}: ml={ml}, reml={reml}")
        return (log10_lambda, lam, sigma_g2, sigma_e2, ml, reml)

    def null_state(self, lam):
        """Null state of the exact scan (`build_sparse_splmm_null_state`, src/stats/splmm.rs:3500-3660) in SAMPLE space:
        V^-1 X, A = X'V^-1 X (jittered only if it fails), Py = V^-1 (y - X b), yPy -> (V^-1 X (n, p), Py (n), chol(A), yPy)."""
        vi = self.solve(lam, np.concatenate([self.y_raw[:, None], self.x_design], axis=1))
        vinv_x = vi[:, 1:]
        a = self.x_design.T @ vinv_x
        a_chol = _spd_cholesky_with_jitter(0.5 * (a + a.T), "SparseLMM XtWX")
        b0 = np.linalg.solve(a_chol.T, np.linalg.solve(a_chol, vinv_x.T @ self.y_raw))
        py = vi[:, 0] - vinv_x @ b0
        ypy = float(self.y_raw @ py)
        if not (np.isfinite(ypy) and ypy > 0.0):
            raise RuntimeError(f"SparseLMM exact scan requires finite positive yPy on K + lambda I scale, got {ypy}")
        return vinv_x, py, a_chol, ypy

    def device_csr(self):
        """(rowptr int64, col int32, val f64) of K in HBM, uploaded once."""
        import torch
        if self._dev is None:
            dev = torch.device("cuda", torch.cuda.current_device())
            self._dev = (torch.from_numpy(self.k.indptr.astype(np.int64)).to(dev),
                         torch.from_numpy(self.k.indices.astype(np.int32)).to(dev),
                         torch.from_numpy(np.ascontiguousarray(self.k.data, dtype=np.float64)).to(dev))
        return self._dev


def _sparse_reml_model(path, y, x_cov, sample_indices):
    """The model of K + lambda I for the sparse-GRM routes: the spectral form (one dense or block-diagonal eigendecomposition on
    the GPU) wherever every connected component fits one dense eigenproblem, the sparse-factor form beyond that
    (`sparse_component_limit()`; JXGPU_SPLMM_ROUTE=factor forces it)."""
    if os.environ.get("JXGPU_SPLMM_ROUTE", "").strip().lower() == "factor":
        return _SparseFactorReml(path, y, x_cov, sample_indices)
    try:
        return _SpectralSparseReml(path, y, x_cov, sample_indices)
    except _ComponentTooLarge:
        return _SparseFactorReml(path, y, x_cov, sample_indices)


def _spd_cholesky_with_jitter(mat, label):
    """src/stats/spreml.rs:324-351 (pivot floor 1e-18 of `cholesky_inplace`, src/math/linalg.rs:341-363)."""
    def try_chol(a):
        try:
            l = np.linalg.cholesky(a)
        except np.linalg.LinAlgError:
            return None
        return l if (np.diag(l) ** 2 > 1e-18).all() else None
    dim = mat.shape[0]
    l = try_chol(mat)
    if l is not None:
        return l
    base = max(float(np.abs(np.diag(mat)).sum()) / max(dim, 1), 1.0) * 1e-10
    for k in range(8):
        l = try_chol(mat + np.eye(dim) * (base * 10.0 ** k))
        if l is not None:
            return l
    raise RuntimeError(f"{label} is not SPD even after diagonal jitter")


def _spreml_grid(model, low, high, grid_size, vp_fixed):
    import math
    if not (math.isfinite(low) and math.isfinite(high)) or low >= high:
        raise RuntimeError(f"SPREML grid search requires finite low < high, got low={low}, high={high}")
    grid_n = max(int(grid_size), 2)
    evals, best, first_err = [], None, None
    for i in range(grid_n):
        x = low + (high - low) * (i / (grid_n - 1))
        try:
            ev = model.evaluate(x, vp_fixed)
        except RuntimeError as e:
            if first_err is None:
                first_err = str(e)
            continue
        if best is None or ev[5] > best[5]:
            best = ev
        evals.append(ev)
    if best is None:
        tail = f"; first failure: {first_err}" if first_err else ""
        raise RuntimeError(f"SPREML sparse grid search found no valid lambda in [{low}, {high}]{tail}")
    return best, evals


def _spreml_tuple(best, grid):
    return (best[1], best[2], best[3], best[4], best[5], best[0], [g[0] for g in grid], [g[5] for g in grid],
            [g[2] for g in grid], [g[3] for g in grid])


def _spreml_brent(model, low, high, grid_size, tol, max_iter, vp_fixed, progress_callback):
    """`sparse_reml_brent_search_with_progress` (src/stats/spreml.rs:591-757)."""
    import math
    if not (math.isfinite(tol) and tol > 0.0):
        raise RuntimeError(f"SPREML Brent tol must be finite and > 0, got {tol}")
    if int(max_iter) == 0:
        raise RuntimeError("SPREML Brent max_iter must be > 0")
    grid_n = max(int(grid_size), 2)
    total = 1 + grid_n + max(int(max_iter), 1)
    best, grid = _spreml_grid(model, low, high, grid_size, vp_fixed)
    bi = next((i for i, g in enumerate(grid) if g[0] == best[0]), 0)
    b_low = grid[bi - 1][0] if bi > 0 else low
    b_high = grid[bi + 1][0] if bi + 1 < len(grid) else high
    if bi == 0 and grid and grid_n > 1:
        raw_step = (high - low) / (grid_n - 1)
        first_valid = grid[0][0]
        prev_raw = max(first_valid - raw_step, low)
        if prev_raw < first_valid:     # `refine_monotone_valid_lower_bound` (:152-186), 24 bisections
            valid = lambda v: model.factorizable(10.0 ** v)   # noqa: E731
            if valid(first_valid):
                if valid(prev_raw):
                    b_low = prev_raw
                else:
                    lo, hi, tol_use = prev_raw, first_valid, max(abs(min(tol, 1e-2)), 1e-6)
                    for _ in range(24):
                        if abs(hi - lo) <= tol_use:
                            break
                        mid = 0.5 * (lo + hi)
                        if valid(mid):
                            hi = mid
                        else:
                            lo = mid
                    b_low = hi
            else:
                b_low = first_valid
    if not (math.isfinite(b_low) and math.isfinite(b_high) and b_low < b_high):
        b_low, b_high = low, high

    def cost(v):
        try:
            return -model.evaluate(v, vp_fixed)[5]
        except RuntimeError:
            return 1e300

    x_best, _ = _brent_minimize_with_init(cost, b_low, b_high, tol, max_iter, best[0])
    out = model.evaluate(x_best, vp_fixed)
    if progress_callback is not None:
        progress_callback(total, total)
    return _spreml_tuple(out, grid)


def spreml_sparse_reml_grid_from_jxgrm(jxgrm_path, y, x_cov=None, sample_indices=None, low=-5.0, high=5.0,
                                       grid_size=33, threads=1):
    """src/stats/spreml.rs:826-918 -> (lambda, sigma_g2, sigma_e2, ml, reml, log10_lambda, grid_log10, grid_reml,
    grid_sigma_g2, grid_sigma_e2): REML profile of y ~ [1, x_cov] + g, Var g = sigma_g2 K (sparse), on a lambda grid."""
    model = _sparse_reml_model(jxgrm_path, y, x_cov, sample_indices)
    return _spreml_tuple(*_spreml_grid(model, low, high, grid_size, None))


def spreml_sparse_reml_brent_from_jxgrm(jxgrm_path, y, x_cov=None, sample_indices=None, low=-5.0, high=5.0,
                                        grid_size=9, tol=1e-3, max_iter=20, threads=1, progress_callback=None):
    """src/stats/spreml.rs:920-1042: grid, then Brent between the best point's neighbours (same 10-tuple)."""
    model = _sparse_reml_model(jxgrm_path, y, x_cov, sample_indices)
    return _spreml_brent(model, low, high, grid_size, tol, max_iter, None, progress_callback)


def spreml_sparse_fastgwa_fixed_vp_brent_from_jxgrm(jxgrm_path, y_resid, vp_fixed, sample_indices=None, low=-5.0,
                                                    high=5.0, grid_size=9, tol=1e-3, max_iter=20, threads=1,
                                                    progress_callback=None):
    """src/stats/spreml.rs:1044-1160: the fastGWA objective (Vp fixed, intercept-only design on residuals)."""
    model = _sparse_reml_model(jxgrm_path, y_resid, None, sample_indices)
    return _spreml_brent(model, low, high, grid_size, tol, max_iter, float(vp_fixed), progress_callback)


def _splmm_exact_null_state(model, lam):
    """Null state of the exact scan on the K + lambda I scale from the f64 spectrum (`build_sparse_splmm_null_state`,
    src/stats/splmm.rs:3500-3660): W = 1 / (s + lambda), A = X~'WX~ (jittered only if it fails, :1947-1976),
    Py~ = W (y~ - X~ b), yPy -> (w, Py~, WX~ as f32 device tensors, chol(A), yPy)."""
    import torch
    dev = model.s_dev.device
    d = model.s + lam
    wxh = model.xr / d[:, None]
    a_chol = _spd_cholesky_with_jitter(model.xr.T @ wxh, "SparseLMM XtWX")
    b0 = np.linalg.solve(a_chol.T, np.linalg.solve(a_chol, wxh.T @ model.yr))
    pyh = (model.yr - model.xr @ b0) / d
    ypy = float(model.yr @ pyh)
    if not (np.isfinite(ypy) and ypy > 0.0):
        raise RuntimeError(f"SparseLMM exact scan requires finite positive yPy on K + lambda I scale, got {ypy}")
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)   # noqa: E731
    return f32(1.0 / d), f32(pyh), f32(wxh), a_chol, ypy


def splmm_assoc_pcg_dense_f32(g, y, lbd, sparse_jxgrm_path, x_cov=None, sparse_sample_indices=None, threads=0, block_rows=0):
    """src/stats/splmm.rs:5464-5650: SparseLMM exact scan of an already decoded SNP-major f32 matrix `g` (m, n) at a GIVEN
    lambda against the sparse GRM file (the `SparseLMM.gwas` API of python/janusx/assoc/api.py:898-940): null state of
    `build_sparse_splmm_null_state`, then `exact_scan_blocks_core` with the rows taken as they are (no flip, no imputation)
    -> f64 (m, 3) = beta, se, p.  V = K + lambda I through the eigenbasis of the sparse K (one dense or block-diagonal
    decomposition, cached per GRM): rows rotated on the device, g'V^-1 g / X'V^-1 g / g.Py as weighted sums."""
    import torch
    from . import pipeline as pl
    lbd = float(lbd)
    if not (np.isfinite(lbd) and lbd >= 0.0):
        raise RuntimeError("lbd must be finite and >= 0")
    g = np.asarray(g)
    if g.ndim != 2:
        raise RuntimeError("g must be 2D SNP-major (m, n)")
    m, n = int(g.shape[0]), int(g.shape[1])
    if m == 0 or n == 0:
        raise RuntimeError("g must have positive shape (m > 0, n > 0)")
    g = _c(g, np.float32)
    y = _c(y, np.float64).ravel()
    if y.shape[0] != n:
        raise RuntimeError(f"Dense SparseLMM requires len(y) to equal g.shape[1], got len(y)={y.shape[0]} and n={n}")
    p = 1
    if x_cov is not None:
        xc = _c(x_cov, np.float64)
        if xc.ndim != 2:
            raise RuntimeError("x_cov must be 2D (n, p_cov)")
        if xc.shape[0] != n:
            raise RuntimeError(f"x_cov row count mismatch: got {xc.shape[0]}, expected {n}")
        p += int(xc.shape[1])
    if n <= p:
        raise RuntimeError(f"Dense SparseLMM requires n > p, got n={n}, p={p}")
    if sparse_sample_indices is not None and len(np.asarray(sparse_sample_indices).ravel()) != n:
        raise RuntimeError("Dense SparseLMM scan requires factor subset n to match g/y n; "
                           f"factor_n={len(np.asarray(sparse_sample_indices).ravel())}, y_n={n}")
    model = _sparse_reml_model(sparse_jxgrm_path, y, x_cov, sparse_sample_indices)
    if not model.factorizable(lbd):
        raise RuntimeError(f"K + lambda I is not positive definite at lambda={lbd}")
    if getattr(model, "factor_route", False):
        # a connected component beyond one dense eigenproblem: multi-vector CG over the CSR image of K per block of rows
        dev = torch.device("cuda", torch.cuda.current_device())
        vinv_x, py, a_chol, ypy = model.null_state(lbd)
        rows_f32 = lambda r0, nr: torch.from_numpy(g[r0:r0 + nr]).to(dev)      # noqa: E731
        out = pl.scan_rows_splmm_factor(rows_f32, m, n, model.p, model.device_csr(), model.diag, lbd, vinv_x, py, a_chol, ypy, dev)
        return out.cpu().numpy()
    fv_state = _splmm_exact_null_state(model, lbd)
    dev = model.s_dev.device
    if model.blocks is not None:
        parts = [(off, nb, ut64.to(torch.float32), torch.from_numpy(np.ascontiguousarray(model.perm[off:off + nb])).to(dev))
                 for off, nb, ut64 in model.blocks]
    else:
        parts = [(0, n, model.ut_dev.to(torch.float32), None)]
    out = pl.scan_rows_splmm_dense(g, parts, n, model.p, fv_state, dev, int(block_rows) if int(block_rows) > 0 else 8192)
    return out.cpu().numpy()


def splmm_exact_scan_from_jxgrm(jxgrm_path, y, packed, packed_n_samples, maf, row_flip, x_cov=None,
                                sample_indices=None, row_indices=None, log10_lambda=None, low=-5.0, high=5.0,
                                grid_size=9, tol=1e-3, max_iter=20, grm_sample_indices=None):
    """SparseLMM exact association scan: the null model of `spreml_sparse_reml_brent_from_jxgrm` (or a given
    log10_lambda) followed by `exact_scan_blocks_core` (src/stats/splmm.rs:2567-2880) over the packed rows — the two
    stages the reference's `splmm_assoc_pcg_bed` workflow chains for its exact mode.  V = K + lambda I is never
    factorised: the eigenvectors of the sparse K rotate every SNP (the MFMA rotation kernel of the dense LMM, LUT
    [0, 2 maf, 1, 2] or flipped, missing = mean, not centred), and g'V^-1 g, X'V^-1 g, g.Py are weighted sums over the
    rotated row (`jxg_splmm_exact_scan_dev`).  -> (stats (m, 3) f64 [beta, se, p], log10_lambda, null 10-tuple or None).
    `grm_sample_indices` (extension): positions of the same samples inside the sparse GRM when its sample order is not
    the genotype file's (`sample_indices` then indexes the packed payload only); default: the same indices for both."""
    import torch
    from . import pipeline as pl
    model = _sparse_reml_model(jxgrm_path, y, x_cov, sample_indices if grm_sample_indices is None else grm_sample_indices)
    panel_idx = model.sample_idx if grm_sample_indices is None else (
        None if sample_indices is None else _c(sample_indices, np.int64).ravel())
    if grm_sample_indices is not None and panel_idx is not None and panel_idx.shape[0] != model.n:
        raise RuntimeError(f"sample_indices ({panel_idx.shape[0]}) and grm_sample_indices ({model.n}) differ in length")
    if model.blocks is not None and grm_sample_indices is not None:
        # block route: the model lives in its block order of the samples; the payload samples follow it
        panel_idx = model.perm if panel_idx is None else panel_idx[model.perm]
    null = None
    if log10_lambda is None:
        null = _spreml_brent(model, low, high, grid_size, tol, max_iter, None, None)
        log10_lambda = null[5]
    lam = 10.0 ** float(log10_lambda)
    if not model.factorizable(lam):
        raise RuntimeError(f"K + lambda I is not positive definite at lambda={lam}")
    n_full = int(packed_n_samples)
    pk, _pk_ptr, _pk_rows = _payload(packed, n_full)
    maf32 = _c(maf, np.float32).ravel()
    flip = np.asarray(row_flip).astype(bool).ravel()
    if maf32.shape[0] != pk.shape[0] or flip.shape[0] != pk.shape[0]:
        raise RuntimeError("maf / row_flip length must match packed rows")
    rows = np.arange(pk.shape[0], dtype=np.int64) if row_indices is None else _c(row_indices, np.int64).ravel()
    if rows.size and (rows.min() < 0 or rows.max() >= pk.shape[0]):
        raise RuntimeError("row_indices out of range")
    mean_g = np.clip(np.float32(2.0) * maf32[rows], np.float32(0.0), np.float32(2.0)).astype(np.float32)
    lut = np.empty((len(rows), 4), dtype=np.float32)
    lut[:, 0] = np.where(flip[rows], 2.0, 0.0)
    lut[:, 1] = mean_g
    lut[:, 2] = 1.0
    lut[:, 3] = np.where(flip[rows], 0.0, 2.0)
    if getattr(model, "factor_route", False):
        dev = torch.device("cuda", torch.cuda.current_device())
        packed_t = pk.to(dev) if _is_device_tensor(pk) else torch.from_numpy(pk if pk.flags.writeable else pk.copy()).to(dev)
        panel = pl.Panel(packed_t, n_full, panel_idx)
        rows_t = torch.from_numpy(rows.astype(np.int32)).to(dev)
        lut_t = torch.from_numpy(lut).to(dev)
        vinv_x, py, a_chol, ypy = model.null_state(lam)

        def rows_f32(r0, nr):
            buf = torch.empty((nr, model.n), dtype=torch.float32, device=dev)
            check(lib().jxg_decode_rows_p32(panel.p32.data_ptr(), panel.m, model.n, rows_t[r0:].data_ptr(), nr,
                                            lut_t[r0:].data_ptr(), buf.data_ptr(), model.n, pl._stream()))
            return buf
        out = pl.scan_rows_splmm_factor(rows_f32, len(rows), model.n, model.p, model.device_csr(), model.diag, lam, vinv_x, py,
                                        a_chol, ypy, dev)
        return out.cpu().numpy(), float(log10_lambda), null
    dev = model.s_dev.device
    fv_state = _splmm_exact_null_state(model, lam)
    if _is_device_tensor(pk):
        packed_t = pk.to(dev)
    else:   # torch.from_numpy wants a writable array (memmapped payloads are not)
        packed_t = torch.from_numpy(pk if pk.flags.writeable else pk.copy()).to(dev)
    if model.blocks is not None:
        if panel_idx is None:
            panel_idx = np.arange(n_full, dtype=np.int64)
        rot = pl.BlockRotation(packed_t, n_full, panel_idx, model.blocks)
        out = _scan_my_rows(lambda r, l: pl.scan_rows_splmm_blocks(rot, model.p, r, l, fv_state), rows.astype(np.int32), lut)
    else:
        panel = pl.Panel(packed_t, n_full, panel_idx)
        sm = pl.SpectralModel(model.s_dev, model.ut_dev, model.x_design, model.y_raw, fit_null=False)
        out = _scan_my_rows(lambda r, l: pl.scan_rows(panel, sm, r, l, mode="splmm", fv_state=fv_state),
                            rows.astype(np.int32), lut)
    return out.cpu().numpy(), float(log10_lambda), null


# ------------------------------------------------------------------------------------------------
# `jx gwas -splmm` / `-splmm-exact`: splmm_assoc_pcg_bed[_to_tsv] (src/stats/splmm.rs:4641-5026)
# ------------------------------------------------------------------------------------------------

SPLMM_DEFAULT_RHAT_MARKERS = 30          # src/stats/splmm.rs:65-66
SPLMM_DEFAULT_RHAT_SEED = 20260527


class _StdRngU32:
    """`StdRng::seed_from_u64` of rand 0.9.2 (Cargo.toml:42; the crate is not part of /root/reference): ChaCha12 keyed from a
    PCG32 expansion of the seed, words in block order; `random_range(0..m)` with Canon's single-sample method (u32 draws
    when the range fits 32 bits).  Restated from the published algorithm; the ChaCha core is pinned to RFC 7539, the seeding
    and range-sampling conventions are not pinned by any vector of the reference."""

    def __init__(self, seed):
        state = int(seed) & 0xffffffffffffffff
        key = []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & 0xffffffffffffffff
            xs = (((state >> 18) ^ state) >> 27) & 0xffffffff
            rot = state >> 59
            key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xffffffff)
        self.key, self.counter, self.buf = key, 0, []

    def _block(self):
        def rotl(v, c):
            return ((v << c) & 0xffffffff) | (v >> (32 - c))
        st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + self.key + [self.counter & 0xffffffff,
                                                                            (self.counter >> 32) & 0xffffffff, 0, 0]
        x = list(st)

        def qr(a, b, c, d):
            x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 16)
            x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 12)
            x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 8)
            x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 7)
        for _ in range(6):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        self.counter += 1
        return [(x[i] + st[i]) & 0xffffffff for i in range(16)]

    def next_u32(self):
        if not self.buf:
            self.buf = self._block()
        return self.buf.pop(0)

    def random_range(self, m):
        if m - 1 > 0xffffffff:
            bits, draw = 64, lambda: self.next_u32() | (self.next_u32() << 32)
        else:
            bits, draw = 32, self.next_u32
        mask = (1 << bits) - 1
        prod = draw() * m
        result, lo = prod >> bits, prod & mask
        if lo > ((-m) & mask):
            if lo + ((draw() * m) >> bits) > mask:
                result += 1
        return result


def splmm_choose_rhat_rows(m, count, seed):
    """`choose_rhat_rows` (src/stats/splmm.rs:1493-1507)."""
    soft_cap = min(int(count), int(m))
    if soft_cap == m:
        return np.arange(m, dtype=np.int64)
    rng = _StdRngU32(seed)
    return np.unique(np.array([rng.random_range(int(m)) for _ in range(max(2 * soft_cap, soft_cap))], dtype=np.int64))


def _splmm_prepare_inputs(prefix, y, x_cov, sample_indices, operator_sample_indices, site_keep, packed, packed_n_samples,
                          maf, row_flip, row_missing, row_indices, model, mmap_window_mb):
    """`prepare_splmm_assoc_inputs` (src/stats/splmm.rs:1302-1490): the three input forms -- external packed payload,
    BED prefix with the caller's row metadata, BED prefix alone (row statistics on the scan samples, no filter)."""
    from . import bed as _bed
    from . import stats as st
    gm_code = st.genetic_model_code(model)     # add / dom / rec / het, case-insensitive (`PackedGeneticModel::parse`, decode.rs:107-119)
    bed_prefix = _bed_prefix(prefix)
    if not bed_prefix:
        raise RuntimeError("BED-prefix mode requires a non-empty prefix")
    yv = _c(y, np.float64).ravel()
    n = int(yv.shape[0])
    if n == 0:
        raise RuntimeError("y must not be empty")
    xc = None
    if x_cov is not None:
        xc = _c(x_cov, np.float64)
        if xc.ndim != 2:
            raise RuntimeError("x_cov must be 2D (n, p_cov)")
        if xc.shape[0] != n:
            raise RuntimeError(f"x_cov row count mismatch: got {xc.shape[0]}, expected {n}")
    p = 1 + (0 if xc is None else int(xc.shape[1]))
    if n <= p:
        raise RuntimeError(f"n must be > p for SparseLMM: n={n}, p={p}")
    use_packed = packed is not None or int(packed_n_samples) > 0
    use_meta = (not use_packed) and any(a is not None for a in (maf, row_flip, row_missing, row_indices))
    if use_packed:
        for name, arr in (("packed", packed), ("maf", maf), ("row_flip", row_flip)):
            if arr is None:
                raise RuntimeError(f"splmm_assoc_pcg_bed: packed payload path requires `{name}` argument.")
        if int(packed_n_samples) <= 0:
            raise RuntimeError("splmm_assoc_pcg_bed: packed payload path requires packed_n_samples > 0")
        n_full = int(packed_n_samples)
        pk, _ptr_, m_packed = _payload(packed, n_full)
    else:
        if use_meta:
            for name, arr in (("maf", maf), ("row_flip", row_flip), ("row_indices", row_indices)):
                if arr is None:
                    raise RuntimeError(f"splmm_assoc_pcg_bed: mmap metadata path requires `{name}` argument.")
        pk, n_full, _bim = _bed.stage_bed_payload(bed_prefix, mmap_window_mb)
        m_packed = int(pk.shape[0])
    if n_full == 0:
        raise RuntimeError("No samples found in BED input.")

    def idx(a, label):
        if a is None:
            return None
        v = _c(a, np.int64).ravel()
        if v.size and (v.min() < 0 or v.max() >= n_full):
            raise RuntimeError(f"{label} out of range for n_samples={n_full}")
        return v
    scan_idx = idx(sample_indices, "sample_indices")
    op_idx = idx(operator_sample_indices, "operator_sample_indices")
    if op_idx is None:
        op_idx = scan_idx
    for v, label in ((scan_idx, "sample_indices"), (op_idx, "operator_sample_indices")):
        if v is not None and v.shape[0] != n:
            raise RuntimeError(f"{label} length mismatch: got {v.shape[0]}, expected {n}")
    if scan_idx is None and n != n_full:
        raise RuntimeError(f"len(y)={n} must equal scan n_samples={n_full} when sample_indices is not provided")
    if use_packed or use_meta:
        rows = None if row_indices is None else _c(row_indices, np.int64).ravel()
        if rows is not None and rows.size and (rows.min() < 0 or rows.max() >= m_packed):
            raise RuntimeError("row_indices out of range")
        m = m_packed if rows is None else int(rows.shape[0])
        maf32 = _c(maf, np.float32).ravel()
        flip = np.asarray(row_flip).astype(bool).ravel()
        miss = np.full(m, np.nan, dtype=np.float32) if row_missing is None else _c(row_missing, np.float32).ravel()
        if maf32.shape[0] != m or flip.shape[0] != m or miss.shape[0] != m:
            raise RuntimeError(f"{'packed' if use_packed else 'mmap'} metadata length mismatch: rows={m}, "
                               f"row_maf={maf32.shape[0]}, row_flip={flip.shape[0]}, row_missing={miss.shape[0]}")
        if rows is None:
            rows = np.arange(m, dtype=np.int64)
    else:
        # `prepare_prefix_input` (:1248-1290): packed-prep row statistics on the scan samples, thresholds (0, 1, 0); the
        # logic meta keeps maf = ALT frequency and never flips (src/io/gfreader.rs:5139, 5378-5460)
        counts = bed_row_counts(pk, n_full, scan_idx)
        keep, miss_all, maf_all, _std = st.packed_prep_row_stats(counts, n if scan_idx is not None else n_full, 0.0, 1.0, 0.0)
        flip_all = np.zeros(m_packed, dtype=bool)
        if site_keep is not None:
            sk = np.asarray(site_keep).astype(bool).ravel()
            if sk.shape[0] != m_packed:
                raise RuntimeError(f"site_keep length mismatch: got {sk.shape[0]}, expected {m_packed}")
            keep = keep & sk
        rows = np.nonzero(keep)[0].astype(np.int64)
        maf32, flip, miss = maf_all[rows].astype(np.float32), flip_all[rows], miss_all[rows].astype(np.float32)
    return dict(pk=pk, n_full=n_full, rows=rows, maf=maf32, flip=flip, miss=miss, y=yv, x_cov=xc, scan_idx=scan_idx,
                op_idx=op_idx, bed_prefix=None if use_packed else bed_prefix, n=n, p=p, gm_code=gm_code)


def _splmm_approx_scan(model, lam, pk, n_full, maf32, flip, rows, scan_idx, rhat_markers, rhat_seed, rhat_rows=None,
                       on_block=None, gm_code=0):
    """`estimate_residualized_approx_scan_sparse` (src/stats/splmm_approx.rs:701-795) on the spectral form of K + lambda I
    (`_SpectralSparseReml`): residualised response, a = V^-1 y_r / sigma2 through the eigenbasis, gamma from the sampled
    markers (their rotation by the MFMA kernel + `jxg_splmm_gamma_sums`), scan model a_r = M_X a, GRAMMAR scan in sample
    space (`jxg_splmm_grammar_scan_p32`).  -> (gamma, stats (m, 3) on the device, markers requested, markers used)."""
    import math
    import torch
    from . import pipeline as pl
    n, p = model.n, model.p
    x, yv = model.x_design, model.y_raw                   # the model's sample order (block order on the block route)
    xtx = x.T @ x
    cx = _spd_cholesky_with_jitter(xtx, "SparseLMM approx XtX")
    solve = lambda b: np.linalg.solve(cx.T, np.linalg.solve(cx, b))        # noqa: E731
    c_y = solve(x.T @ yv)
    y_resid = yv - x @ c_y
    rss = float(y_resid @ y_resid)
    if not (math.isfinite(rss) and rss > 1e-30):
        raise RuntimeError(f"SparseLMM residualized approx produced invalid residualized RSS: {rss}")
    sigma2 = rss / (float(n - p) * (1.0 + lam))
    if not (math.isfinite(sigma2) and sigma2 > 0.0):
        raise RuntimeError(f"SparseLMM residualized approx produced invalid scan sigma2 at lambda={lam}: {sigma2}")
    factor = getattr(model, "factor_route", False)
    if factor:
        # no spectral form (a connected component beyond one dense eigenproblem): V^-1 from the host sparse factor, as the
        # reference does (`SparseJxgrmCholesky`): one solve for a, one multi-vector solve for the sampled markers
        dev = torch.device("cuda", torch.cuda.current_device())
        a_vec = model.solve(lam, y_resid) / sigma2
    else:
        w = 1.0 / (model.s + lam)
        a_rot = w * (model.yr - model.xr @ c_y) / sigma2      # U'a,  a = (K + lambda I)^-1 y_r / sigma2
        dev = model.s_dev.device
        a_rot_t = torch.from_numpy(a_rot).to(dev)
        if model.blocks is not None:
            a_t = torch.empty(n, dtype=torch.float64, device=dev)
            for off, nb, utb in model.blocks:
                a_t[off:off + nb] = utb.T @ a_rot_t[off:off + nb]
        else:
            a_t = model.ut_dev.T @ a_rot_t
        a_vec = a_t.cpu().numpy()
    # ---- gamma from sampled markers
    m = len(rows)
    rr = splmm_choose_rhat_rows(m, rhat_markers, rhat_seed) if rhat_rows is None else _c(rhat_rows, np.int64).ravel()
    if rr.size and (rr.min() < 0 or rr.max() >= m):
        raise RuntimeError("rhat rows out of range")
    mean_g = np.clip(np.float32(2.0) * maf32, np.float32(0.0), np.float32(2.0)).astype(np.float32)
    lut = np.empty((m, 4), dtype=np.float32)
    lut[:, 0] = np.where(flip, 2.0, 0.0)
    lut[:, 1] = mean_g
    lut[:, 2] = 1.0
    lut[:, 3] = np.where(flip, 0.0, 2.0)
    if int(gm_code) != 0:
        # dom / rec / het (`decode_packed_row_model_into_f64`, src/decode/decode.rs:305-364, as the GRAMMAR scan's non-additive
        # branch and the sampled-marker decode call it: src/stats/splmm.rs:3211-3262, 1514-1560): the genetic model applied to
        # the raw values [0 | 2, max(2 maf, 0), 1, 2 | 0] INCLUDING the imputed entry, no centring (the scan residualises on X)
        from .stats import _apply_genetic_model
        lut[:, 1] = np.maximum(np.float32(2.0) * maf32, np.float32(0.0))
        for c in range(4):
            lut[:, c] = _apply_genetic_model(int(gm_code), lut[:, c])
    if _is_device_tensor(pk):
        packed_t = pk.to(dev)
    else:
        packed_t = torch.from_numpy(pk if pk.flags.writeable else pk.copy()).to(dev)
    panel_idx = scan_idx
    if model.perm is not None:
        panel_idx = model.perm if scan_idx is None else scan_idx[model.perm]
    sub = packed_t[torch.from_numpy(rows[rr]).to(dev)]    # payload of the sampled markers only
    sub_rows = np.arange(len(rr), dtype=np.int32)
    if factor:
        # the sampled markers decoded on the device, their sums in sample space on the host: gg, g'V^-1 g (one multi-vector
        # solve per 256 markers), g'a, X'g, X'V^-1 g = (V^-1 X)'g
        sp_panel = pl.Panel(sub, n_full, panel_idx)
        gdec = torch.empty((len(rr), n), dtype=torch.float32, device=dev)
        lut_s = torch.from_numpy(np.ascontiguousarray(lut[rr])).to(dev)
        check(lib().jxg_decode_rows_p32(sp_panel.p32.data_ptr(), sp_panel.m, n, None, len(rr), lut_s.data_ptr(), gdec.data_ptr(), n,
                                        pl._stream()))
        gh = gdec.cpu().numpy().astype(np.float64)
        del gdec, sp_panel
        vinv_x = model.solve(lam, x)
        sums = np.empty((len(rr), 3 + 2 * p), dtype=np.float64)
        for k0 in range(0, len(rr), 256):
            gb = gh[k0:k0 + 256]
            zb = model.solve(lam, gb.T)                     # (n, <= 256)
            sums[k0:k0 + 256, 0] = np.einsum("kn,kn->k", gb, gb)
            sums[k0:k0 + 256, 1] = np.einsum("kn,nk->k", gb, zb)
            sums[k0:k0 + 256, 2] = gb @ a_vec
            sums[k0:k0 + 256, 3:3 + p] = gb @ x
            sums[k0:k0 + 256, 3 + p:] = gb @ vinv_x
        del gh, sub
        xtwx = x.T @ vinv_x
        xta = x.T @ a_vec
    else:
        if model.blocks is not None:
            full_idx = np.arange(n_full, dtype=np.int64) if panel_idx is None else panel_idx
            rot = pl.BlockRotation(sub, n_full, full_idx, model.blocks)
            grot = pl.rotate_rows_blocks(rot, sub_rows, lut[rr])
            del rot
        else:
            sm = pl.SpectralModel(model.s_dev, model.ut_dev, model.x_design, model.y_raw, fit_null=False)
            grot = pl.rotate_rows(pl.Panel(sub, n_full, panel_idx), sm, sub_rows, lut[rr])
        sums = torch.empty((len(rr), 3 + 2 * p), dtype=torch.float64, device=dev)
        xr_t = torch.from_numpy(np.ascontiguousarray(model.xr)).to(dev)
        w_t = torch.from_numpy(w).to(dev)
        check(lib().jxg_splmm_gamma_sums(grot.data_ptr(), len(rr), n, n, p, w_t.data_ptr(), a_rot_t.data_ptr(), xr_t.data_ptr(),
                                         sums.data_ptr(), pl._stream()))
        sums = sums.cpu().numpy()
        del grot, sub
        xtwx = model.xr.T @ (model.xr * w[:, None])
        xta = model.xr.T @ a_rot
    fast_sum = res_sum = 0.0
    n_used = res_used = 0
    for k in range(len(rr)):
        gg, wgg, ag = sums[k, 0], sums[k, 1], sums[k, 2]
        xg, wxg = sums[k, 3:3 + p], sums[k, 3 + p:]
        c = solve(xg)
        s_ms = gg - float(xg @ c)
        if not (math.isfinite(s_ms) and math.isfinite(gg)) or s_ms <= max(1e-10, 1e-12 * max(abs(gg), 1.0)):
            continue
        svs = wgg - 2.0 * float(c @ wxg) + float(c @ xtwx @ c)
        if not (math.isfinite(svs) and svs > 1e-30):
            continue
        ratio = svs / s_ms
        if not (math.isfinite(ratio) and ratio > 0.0):
            continue
        res_sum += ratio
        res_used += 1
        score = ag - float(c @ xta)
        chisq = score * score / svs
        if math.isfinite(chisq) and chisq < 5.0:
            fast_sum += ratio
            n_used += 1
        if k + 1 >= rhat_markers and n_used >= 100:
            break
    if n_used == 0 and res_used == 0:
        raise RuntimeError("SparseLMM residualized approx gamma estimation found no valid sampled markers")
    gamma, used = (fast_sum / n_used, n_used) if n_used >= 100 else (res_sum / res_used, res_used)
    gamma *= 1.0 / sigma2
    if not (math.isfinite(gamma) and gamma > 0.0):
        raise RuntimeError(f"SparseLMM residualized approx gamma must be finite and > 0, got {gamma}")
    # ---- scan model + scan, in the caller's sample order
    a_resid = a_vec - x @ solve(x.T @ a_vec)
    if model.perm is not None:
        inv = np.empty(n, dtype=np.int64)
        inv[model.perm] = np.arange(n)
        a_resid, x_scan = a_resid[inv], x[inv]
    else:
        x_scan = x
    panel = pl.Panel(packed_t, n_full, scan_idx)
    if pl.dist_info()[1] > 1:
        on_block = None        # block offsets are those of a rank's share: the gathered table is handed on by the caller
    out = _scan_my_rows(lambda r, l: pl.scan_rows_grammar(panel, r, l, x_scan, a_resid, gamma, on_block=on_block),
                        rows.astype(np.int32), lut)
    return float(gamma), out, int(len(rr)), int(used)


def _splmm_assoc(prefix, y, lbd, x_cov, sample_indices, operator_sample_indices, site_keep, tol, max_iter, block_rows,
                 std_eps, threads, model, rhat_markers, rhat_seed, packed, packed_n_samples, maf, row_flip, row_missing,
                 row_indices, sparse_sample_indices, sparse_jxgrm_path, progress_callback, progress_every, scan_mode,
                 mmap_window_mb, rhat_rows=None):
    import math
    if int(max_iter) == 0:
        raise RuntimeError("max_iter must be > 0")
    if not (math.isfinite(tol) and tol > 0.0):
        raise RuntimeError("tol must be finite and > 0")
    if not (math.isfinite(std_eps) and std_eps > 0.0):
        raise RuntimeError("std_eps must be finite and > 0")
    if not (math.isfinite(lbd) and lbd >= 0.0):
        raise RuntimeError("lbd must be finite and >= 0")
    mode = str(scan_mode).strip().lower()
    if mode not in ("approx", "exact"):
        raise RuntimeError(f"unsupported SparseLMM scan mode: {scan_mode}")
    if mode == "approx" and int(rhat_markers) == 0:
        raise RuntimeError("rhat_markers must be > 0")
    inp = _splmm_prepare_inputs(prefix, y, x_cov, sample_indices, operator_sample_indices, site_keep, packed,
                                packed_n_samples, maf, row_flip, row_missing, row_indices, model, mmap_window_mb)
    path = sparse_jxgrm_path if sparse_jxgrm_path else _normalize_spgrm_path(_bed_prefix(prefix))
    sp_idx = inp["op_idx"] if sparse_sample_indices is None else _c(sparse_sample_indices, np.int64).ravel()
    if sp_idx is not None and sp_idx.shape[0] != inp["n"]:
        raise RuntimeError(f"sparse_sample_indices length mismatch: got {sp_idx.shape[0]}, expected {inp['n']}")
    lam = float(lbd)
    with _progress_hook(progress_callback, progress_every):
        if mode == "exact":
            if inp["gm_code"] != 0:      # `exact_scan_blocks_core`, src/stats/splmm.rs:2662-2664
                raise RuntimeError("SparseLMM exact denominator mode requires additive model")
            if not lam > 0.0:
                raise RuntimeError("K + lambda I is not positive definite at lambda=0")
            out, _l10, _null = splmm_exact_scan_from_jxgrm(path, inp["y"], inp["pk"], inp["n_full"],
                                                           _expand_rows(inp["maf"], inp["rows"], inp["pk"].shape[0]),
                                                           _expand_rows(inp["flip"], inp["rows"], inp["pk"].shape[0]),
                                                           inp["x_cov"], inp["scan_idx"], inp["rows"], math.log10(lam),
                                                           grm_sample_indices=sp_idx)
            r_hat, req, used = float("nan"), 0, 0
        else:
            spm = _sparse_reml_model(path, inp["y"], inp["x_cov"], sp_idx)
            if not spm.factorizable(lam) and not (lam == 0.0 and spm.smin > 0.0):
                raise RuntimeError(f"K + lambda I is not positive definite at lambda={lam}")
            r_hat, out_t, req, used = _splmm_approx_scan(spm, lam, inp["pk"], inp["n_full"], inp["maf"], inp["flip"],
                                                         inp["rows"], inp["scan_idx"], int(rhat_markers), int(rhat_seed),
                                                         rhat_rows, gm_code=inp["gm_code"])
            out = out_t.cpu().numpy()
            req = int(rhat_markers)
    _done(progress_callback, len(inp["rows"]))
    n_all, col_ptr, _r, _v = load_spgrm(path)
    return inp, r_hat, out, req, used, int(col_ptr[-1])


def _expand_rows(values, rows, m_total):
    """Row metadata given for the selected rows -> an array indexed by payload row (what the packed entry points take)."""
    full = np.zeros(int(m_total), dtype=np.asarray(values).dtype)
    full[rows] = values
    return full


def splmm_assoc_pcg_bed(prefix, y, lbd, x_cov=None, sample_indices=None, operator_sample_indices=None, site_keep=None,
                        tol=1e-5, max_iter=200, block_rows=0, std_eps=1e-12, use_train_maf=True, threads=0, model="add",
                        rhat_markers=SPLMM_DEFAULT_RHAT_MARKERS, rhat_seed=SPLMM_DEFAULT_RHAT_SEED, packed=None,
                        packed_n_samples=0, maf=None, row_flip=None, row_missing=None, row_indices=None,
                        sparse_sample_indices=None, sparse_jxgrm_path=None, stage1_progress_callback=None,
                        scan_progress_callback=None, progress_every=0, rhat_tol=1e-3, scan_mode="exact",
                        mmap_window_mb=None, rhat_rows=None):
    """src/stats/splmm.rs:4608-4773 -> (r_hat, y_converged, y_iters, y_rel_res, x_converged_all, x_max_iters, x_max_rel_res,
    rhat_markers_requested, rhat_markers_used, stats f64 (m, 3), factor_nnz).  scan_mode "exact" = null state + score-form
    exact scan at the given lambda (`estimate_rhat_and_scan_sparse`, :3711); "approx" = the residualised GRAMMAR-gamma route
    (`estimate_residualized_approx_scan_sparse`, src/stats/splmm_approx.rs:701).  V^-1 comes from the eigendecomposition of
    the sparse K, so the solve diagnostics are those of a direct method (converged, one iteration, zero residual: what the
    reference reports for its sparse Cholesky, :3620-3645); `factor_nnz` is the number of stored entries of the sparse GRM
    (no L factor exists here).  `rhat_rows` (extension) overrides the seeded choice of the sampled markers."""
    inp, r_hat, out, req, used, nnz = _splmm_assoc(prefix, y, lbd, x_cov, sample_indices, operator_sample_indices, site_keep,
                                                   tol, max_iter, block_rows, std_eps, threads, model, rhat_markers,
                                                   rhat_seed, packed, packed_n_samples, maf, row_flip, row_missing,
                                                   row_indices, sparse_sample_indices, sparse_jxgrm_path,
                                                   scan_progress_callback, progress_every, scan_mode, mmap_window_mb,
                                                   rhat_rows)
    return (r_hat, True, 1, 0.0, True, 1, 0.0, req, used, out, nnz)


def splmm_assoc_pcg_bed_to_tsv(prefix, y, lbd, chrom, pos, snp, allele0, allele1, out_tsv, x_cov=None, sample_indices=None,
                               operator_sample_indices=None, site_keep=None, tol=1e-5, max_iter=200, block_rows=0,
                               std_eps=1e-12, use_train_maf=True, threads=0, model="add",
                               rhat_markers=SPLMM_DEFAULT_RHAT_MARKERS, rhat_seed=SPLMM_DEFAULT_RHAT_SEED, packed=None,
                               packed_n_samples=0, maf=None, row_flip=None, row_missing=None, row_indices=None,
                               sparse_sample_indices=None, sparse_jxgrm_path=None, stage1_progress_callback=None,
                               scan_progress_callback=None, progress_every=0, rhat_tol=1e-3, scan_mode="exact",
                               mmap_window_mb=None, rhat_rows=None):
    """src/stats/splmm.rs:4775-5026 -> (r_hat, y_converged, y_iters, y_rel_res, x_converged_all, x_max_iters, x_max_rel_res,
    rhat_markers_requested, rhat_markers_used, rows written, (prepare, bim, null + scan, writer wait, ...) seconds); the TSV
    has the Basic3 schema with `af` = row_maf and `miss` = the missing rate (:461-475).  Empty metadata lists: the BIM file
    of `prefix` is read for the selected rows."""
    import time
    from .tsv import write_assoc_tsv
    t0 = time.perf_counter()
    inp, r_hat, out, req, used, _nnz = _splmm_assoc(prefix, y, lbd, x_cov, sample_indices, operator_sample_indices, site_keep,
                                                    tol, max_iter, block_rows, std_eps, threads, model, rhat_markers,
                                                    rhat_seed, packed, packed_n_samples, maf, row_flip, row_missing,
                                                    row_indices, sparse_sample_indices, sparse_jxgrm_path,
                                                    scan_progress_callback, progress_every, scan_mode, mmap_window_mb,
                                                    rhat_rows)
    t1 = time.perf_counter()
    m = out.shape[0]
    if not len(chrom):
        from .bed import read_bim
        bim = read_bim(_bed_prefix(prefix))
        sel = inp["rows"]
        chrom = [bim.chrom[j] for j in sel]
        pos = [bim.pos[j] for j in sel]
        snp = [bim.snp[j] for j in sel]
        allele0 = [bim.a0[j] for j in sel]
        allele1 = [bim.a1[j] for j in sel]
    elif not (len(chrom) == len(pos) == len(snp) == len(allele0) == len(allele1) == m):
        raise RuntimeError(f"SparseLMM TSV metadata length mismatch: rows={m}")
    t2 = time.perf_counter()
    from .pipeline import dist_info
    if dist_info()[0] == 0:        # several ranks: every rank holds the gathered table, rank 0 writes it
        written = write_assoc_tsv(out_tsv, chrom, pos, snp, allele0, allele1, inp["maf"], inp["miss"], out, resolve=False)
    else:
        written = m
    t3 = time.perf_counter()
    return (r_hat, True, 1, 0.0, True, 1, 0.0, req, used, int(written), (0.0, t2 - t1, t1 - t0, t3 - t2))


def _bed_prefix(prefix):
    """PLINK prefix with a trailing .bed / .bim / .fam removed (`normalize_plink_prefix`, src/io/gfcore.rs)."""
    t = str(prefix).strip()
    return t[:-4] if t.lower().endswith((".bed", ".bim", ".fam")) else t


def _read_bed_payload(prefix):
    """PLINK .bed payload as (m, bps) uint8 plus n_samples (src/stats/lmm.rs:1050-1061, gfcore.rs:307-323)."""
    from .bed import read_bed_payload
    return read_bed_payload(prefix)


def prepare_bed_2bit_packed(prefix, maf_threshold, max_missing_rate, het_threshold, snps_only=False):
    """src/io/gfreader.rs:7029-7110 (the packed workflow's loader + QC) ->
    (packed_keep u8 (k, bps), missing_rate f32 (k), maf f32 (k) [alt allele frequency], std_denom f32 (k),
    row_flip bool (k) [all False], site_keep bool (m), n_samples, n_snps_total).
    Row counts come from the device popcount kernel; thresholds out of range raise ValueError like the reference."""
    from . import stats as st
    from .bed import read_bed_payload, snps_only_mask
    if not (0.0 <= maf_threshold <= 0.5):
        raise ValueError("maf_threshold must be within [0, 0.5]")
    if not (0.0 <= max_missing_rate <= 1.0):
        raise ValueError("max_missing_rate must be within [0, 1.0]")
    if not (0.0 <= het_threshold <= 1.0):
        raise ValueError("het_threshold must be within [0, 1.0]")
    low = str(prefix).lower()
    if low.endswith((".bed", ".bim", ".fam")):
        prefix = str(prefix)[:-4]
    packed, n_fam, bim = read_bed_payload(prefix)
    counts = bed_row_counts(packed, n_fam, None)
    keep, miss, maf, std = st.packed_prep_row_stats(counts, n_fam, maf_threshold, max_missing_rate, het_threshold)
    if snps_only:
        keep &= snps_only_mask(bim)
    if not keep.any():
        raise RuntimeError("No SNPs left after packed BED filtering. Please relax thresholds.")
    rows = np.nonzero(keep)[0]
    return (np.ascontiguousarray(packed[rows]), miss[rows], maf[rows], std[rows], np.zeros(len(rows), dtype=bool),
            keep, int(n_fam), int(packed.shape[0]))


def grm_packed_bed_f32(prefix, method=1, maf_threshold=0.02, max_missing_rate=0.05, het_threshold=0.0,
                       snps_only=False, block_cols=65536, threads=0, progress_callback=None, progress_every=0):
    """src/stats/grm.rs:3757-3839: `prepare_bed_2bit_packed` then `grm_packed_f32` -> (f32 (n,n), eff_m, n_samples)."""
    maf_thr = min(max(float(maf_threshold), 0.0), 0.5)
    miss_thr = min(max(float(max_missing_rate), 0.0), 1.0)
    pk, _miss, maf, _std, flip, _keep, n, _tot = prepare_bed_2bit_packed(prefix, maf_thr, miss_thr,
                                                                        min(max(float(het_threshold), 0.0), 1.0),
                                                                        snps_only)
    k = grm_packed_f32(pk, n, flip, maf, None, method, block_cols, threads, progress_callback, progress_every)
    return k, int(pk.shape[0]), n


def grm_stream_bed_f32(prefix, method=1, maf_threshold=0.02, max_missing_rate=0.05, het_threshold=0.0,
                       snps_only=False, block_cols=65536, threads=0, progress_callback=None, progress_every=0,
                       mmap_window_mb=None):
    """src/stats/grm.rs:4676-4703 -> (f32 (n,n), eff_m, n_samples).  The payload is staged to HBM in windows of
    `mmap_window_mb` MiB (bed.stage_bed_payload: the reference's WindowedBedMatrix role); no host copy of it is made."""
    import torch
    from .bed import snps_only_mask, stage_bed_payload
    packed, n_samples, bim = stage_bed_payload(_bed_prefix(prefix), mmap_window_mb)
    if snps_only:
        packed = packed[torch.from_numpy(np.nonzero(snps_only_mask(bim))[0]).to(packed.device)]
    k, eff, _ = grm_stream_payload_f32(packed, n_samples, method, maf_threshold, max_missing_rate, het_threshold)
    _done(progress_callback, int(packed.shape[0]))
    return k, int(eff), int(n_samples)


def grm_stream_payload_f32(packed, n_samples, method=1, maf_threshold=0.02, max_missing_rate=0.05,
                           het_threshold=0.0):
    """In-memory form of `grm_stream_bed_f32` -> (K f32, eff_m, keep mask); `packed`: host array or device tensor."""
    n = int(n_samples)
    _pk, pk_ptr, m = _payload(packed, n)
    out = np.empty((n, n), dtype=np.float32)
    eff = np.zeros(1, dtype=np.int64)
    keep = np.zeros(m, dtype=np.uint8)
    check(lib().jx_grm_stream_payload_f32(pk_ptr, m, n, int(method), float(maf_threshold),
                                          float(max_missing_rate), float(het_threshold), _p(out), _p(eff), _p(keep)))
    return out, int(eff[0]), keep.astype(bool)


def grm_stream_bed_f32_to_npy(prefix, out_path, method=1, maf_threshold=0.02, max_missing_rate=0.05,
                              het_threshold=0.0, snps_only=False, block_cols=65536, threads=0,
                              progress_callback=None, progress_every=0, mmap_window_mb=None):
    """src/stats/grm.rs:5517-5545: writes NPY v1 f32 C-order, returns (eff_m, n_samples)."""
    k, eff, n = grm_stream_bed_f32(prefix, method, maf_threshold, max_missing_rate, het_threshold, snps_only,
                                   mmap_window_mb=mmap_window_mb)
    tmp = f"{out_path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as fh:
        np.lib.format.write_array(fh, np.ascontiguousarray(k, dtype=np.float32), version=(1, 0))
    os.replace(tmp, out_path)
    _done(progress_callback, eff)
    return eff, n


# ------------------------------------------------------------------------------------------------
# eigh  (src/math/eigh.rs)
# ------------------------------------------------------------------------------------------------

def grm_bed_f64_from_meta(prefix, row_indices, row_flip, row_maf, sample_indices=None, method=1, block_cols=65536,
                          threads=0, progress_callback=None, progress_every=0, mmap_window_mb=None):
    """src/stats/grm.rs:3639-3753 -> `build_grm_from_meta_stream` (src/stats/gblup.rs:406-652): the GRM of the BED rows
    `row_indices` over `sample_indices` with the caller's flip / allele-frequency metadata (`jx grm` after its own QC,
    python/janusx/script/grm.py:1365; `jx gs`, python/janusx/gs/workflow.py:4139) -> f64 (n, n).  method 1: centred
    additive, scaled by 1 / sum(2p(1-p)); method 2: standardised additive, scaled by 1 / m; 3 (dominance) is not built."""
    import torch
    from .bed import stage_bed_payload
    if int(method) not in (1, 2, 3):
        raise RuntimeError(f"unsupported method={method}; expected 1 (centered additive), 2 (standardized additive), or 3 "
                           "(centered dominance)")
    if int(method) == 3:
        raise RuntimeError("method=3 (centered dominance) is outside this build's scope (additive GRMs only)")
    packed, n_fam, _bim = stage_bed_payload(_bed_prefix(prefix), mmap_window_mb)
    if n_fam == 0:
        raise RuntimeError("no samples found in PLINK input")
    src = _c(row_indices, np.int64).ravel()
    flip = np.asarray(row_flip).astype(bool).ravel()
    maf = _c(row_maf, np.float32).ravel()
    if src.size == 0:
        raise RuntimeError("row_indices must not be empty")
    if flip.shape[0] != src.shape[0] or maf.shape[0] != src.shape[0]:
        raise RuntimeError(f"row meta length mismatch: row_indices={src.shape[0]}, row_flip={flip.shape[0]}, "
                           f"row_maf={maf.shape[0]}")
    if (src < 0).any():
        raise RuntimeError(f"row index must be non-negative, got {int(src[src < 0][0])}")
    if int(src.max()) >= int(packed.shape[0]):
        raise RuntimeError("row index out of range")
    if sample_indices is None:
        tr = np.arange(n_fam, dtype=np.int64)
    else:
        tr = _c(sample_indices, np.int64).ravel()
        if tr.size == 0:
            raise RuntimeError("sample_indices must not be empty")
        if tr.min() < 0 or tr.max() >= n_fam:
            raise RuntimeError("sample_indices out of range")
    rows_payload = packed[torch.from_numpy(src).to(packed.device)]
    del packed
    if int(method) == 1:
        k, _var_sum, _panel, _row_mean = _gblup_meta_grm(rows_payload, n_fam, tr, flip, maf)
        out = k.cpu().numpy()
    else:
        out = grm_packed_f64(rows_payload.cpu().numpy(), n_fam, flip, maf, None if sample_indices is None else tr, 2,
                             block_cols, threads)
    if progress_callback is not None:
        progress_callback(int(src.shape[0]), int(src.shape[0]))
    return np.ascontiguousarray(out, dtype=np.float64)


def _write_npy_f32(out_path, arr):
    tmp = f"{out_path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as fh:
        np.lib.format.write_array(fh, np.ascontiguousarray(arr, dtype=np.float32), version=(1, 0))
    os.replace(tmp, out_path)


def gblup_grm_from_meta_to_npy(prefix, out_npy_path, row_source_indices, row_flip, row_maf, sample_indices=None, method=1,
                               block_rows=65536, threads=0, progress_callback=None, progress_every=0, mmap_window_mb=None):
    """src/stats/gblup.rs:718-857: `build_grm_from_meta_stream` written as NPY v1 f32 (`jx grm` rust-meta route,
    python/janusx/script/grm.py:1459) -> (eff_m, n_samples)."""
    k = grm_bed_f64_from_meta(prefix, row_source_indices, row_flip, row_maf, sample_indices, method, block_rows, threads,
                              None, 0, mmap_window_mb)
    _write_npy_f32(out_npy_path, k)
    eff_m = int(np.asarray(row_source_indices).ravel().shape[0])
    if progress_callback is not None:
        progress_callback(eff_m, eff_m)
    return eff_m, int(k.shape[0])


_DENSE_META_GRM_CACHE = {}      # one entry: the dense GRM a sequence of row-band calls is cut from


def _dense_grm_from_meta_f32(prefix, row_source_indices, row_flip, row_maf, n_total_sites, sample_indices, method,
                             mmap_window_mb, what):
    """Dense GRM under the conventions of the sparse-GRM stream core (`grm_stream_bed_row_band_f32_fill_core`,
    src/stats/spgrm.rs:4266-4570): rows decoded with mean 2 maf (clamped) and, for method 2, 1 / sqrt(2p(1-p)) (0 below
    1e-12), missing = mean; denominator = the f64 sum of the positive finite 2p(1-p) (method 1) or m (method 2) whatever
    the sample selection.  -> (K f32 (n_use, n_use) on the host, eff_m, n_use).  The reference builds the matrix band by
    band to bound host memory; here the whole accumulator lives in HBM and consecutive band calls on the same inputs are
    cut from one cached matrix."""
    import hashlib
    import torch
    from . import pipeline as pl
    from . import stats as st
    from .bed import stage_bed_payload
    bed_prefix = _bed_prefix(prefix)
    if int(n_total_sites) <= 0:
        raise RuntimeError(f"n_total_sites must be positive for dense GRM {what} meta route.")
    if int(method) not in (1, 2):
        raise RuntimeError(f"Dense GRM part method must be 1 (centered) or 2 (standardized); got {method}")
    src = _c(row_source_indices, np.int64).ravel()
    flip = np.asarray(row_flip).astype(bool).ravel()
    maf = _c(row_maf, np.float32).ravel()
    if src.size == 0:
        raise RuntimeError("row_source_indices must not be empty")
    if (src < 0).any():
        raise RuntimeError(f"row_source_indices must be non-negative, got {int(src[src < 0][0])}")
    if flip.shape[0] != src.shape[0] or maf.shape[0] != src.shape[0]:
        raise RuntimeError(f"row meta length mismatch: row_source_indices={src.shape[0]}, row_flip={flip.shape[0]}, "
                           f"row_maf={maf.shape[0]}")
    if int(src.max()) >= int(n_total_sites):
        raise RuntimeError(f"row_source index out of range: {int(src[src >= int(n_total_sites)][0])} >= "
                           f"n_total_sites={int(n_total_sites)}")
    idx, n_sel = _opt_idx(sample_indices)
    st_bed = os.stat(bed_prefix + ".bed")
    h = hashlib.sha1()
    for a in (src, flip.astype(np.uint8), maf, idx if idx is not None else np.zeros(0, np.int64)):
        h.update(np.ascontiguousarray(a).tobytes())
    key = (os.path.abspath(bed_prefix), st_bed.st_size, st_bed.st_mtime_ns, int(method), h.hexdigest())
    hit = _DENSE_META_GRM_CACHE.get(key)
    if hit is not None:
        return hit
    _DENSE_META_GRM_CACHE.clear()
    packed, n_fam, _bim = stage_bed_payload(bed_prefix, mmap_window_mb)
    if n_fam == 0:
        raise RuntimeError("No samples found in BED input.")
    if int(src.max()) >= int(packed.shape[0]):
        raise RuntimeError(f"row_source index out of range for the BED payload: {int(src.max())} >= {int(packed.shape[0])}")
    if idx is not None and n_sel and (idx.min() < 0 or idx.max() >= n_fam):
        raise RuntimeError(f"Dense GRM part sample index out of range: {int(idx.max())} >= {n_fam}")
    n_use = n_sel if (idx is not None and n_sel) else n_fam
    m = int(src.shape[0])
    if int(method) == 1:
        v = 2.0 * maf.astype(np.float64) * (1.0 - maf.astype(np.float64))
        denom = float(np.sum(v[np.isfinite(v) & (v > 0.0)]))
        if not (np.isfinite(denom) and denom > 0.0):
            raise RuntimeError("Dense GRM part centered denominator is not positive")
    else:
        denom = float(m)
    rows_payload = packed[torch.from_numpy(src).to(packed.device)]
    del packed
    panel = pl.Panel(rows_payload, n_fam, idx if (idx is not None and n_sel) else None)
    acc = pl.grm_accumulate(panel, np.arange(m, dtype=np.int64), st.grm_lut_from_maf(maf, flip, int(method)))
    k = pl.grm_finalize(acc, n_use, denom, torch.float32).cpu().numpy()
    del acc
    out = (k, m, n_use)
    if k.nbytes <= (8 << 30):
        _DENSE_META_GRM_CACHE[key] = out
    return out


def _row_band(k, row_part_start, row_part_end):
    n_use = int(k.shape[0])
    a, b = int(row_part_start), int(row_part_end)
    if a >= b or b > n_use:
        raise RuntimeError(f"row band is invalid: start={a}, end={b}, n_samples={n_use}")
    # the band holds the LOWER triangle only (column <= global row; `spgrm_collect_batch_dense_rows`, spgrm.rs:2706-2753): the
    # caller mirrors when it assembles the parts
    return np.ascontiguousarray(np.tril(k[a:b], k=a))


def grm_bed_f32_row_band_from_meta(prefix, row_source_indices, row_flip, row_maf, n_total_sites, row_part_start,
                                   row_part_end, sample_indices=None, method=1, block_rows=0, sample_block=0, threads=0,
                                   mmap_window_mb=None, progress_callback=None, progress_every=0):
    """src/stats/spgrm.rs:5496-5642: rows [row_part_start, row_part_end) of the dense GRM of caller-prepared BED rows
    (`jx grm` part builds, python/janusx/script/grm.py:819) -> (f32 (part_rows, n_use), eff_m, n_use)."""
    k, eff_m, n_use = _dense_grm_from_meta_f32(prefix, row_source_indices, row_flip, row_maf, n_total_sites, sample_indices,
                                               method, mmap_window_mb, "part")
    band = _row_band(k, row_part_start, row_part_end)
    if progress_callback is not None:
        progress_callback(eff_m, eff_m)
    return band, eff_m, n_use


def grm_bed_f32_row_band_from_meta_to_npy(prefix, out_npy_path, row_source_indices, row_flip, row_maf, n_total_sites,
                                          row_part_start, row_part_end, sample_indices=None, method=1, block_rows=0,
                                          sample_block=0, threads=0, mmap_window_mb=None, progress_callback=None,
                                          progress_every=0):
    """src/stats/spgrm.rs:5644-5783: the same band written as NPY v1 f32 (part_rows, n_use) -> (eff_m, n_use)."""
    k, eff_m, n_use = _dense_grm_from_meta_f32(prefix, row_source_indices, row_flip, row_maf, n_total_sites, sample_indices,
                                               method, mmap_window_mb, "part")
    _write_npy_f32(out_npy_path, _row_band(k, row_part_start, row_part_end))
    if progress_callback is not None:
        progress_callback(eff_m, eff_m)
    return eff_m, n_use


def grm_bed_f32_tiled_from_meta_to_npy(prefix, out_npy_path, row_source_indices, row_flip, row_maf, n_total_sites,
                                       sample_indices=None, method=1, block_rows=0, sample_block=0, threads=0,
                                       mmap_window_mb=None, progress_callback=None, progress_every=0):
    """src/stats/spgrm.rs:5785-5922: the whole dense GRM of caller-prepared BED rows written as NPY v1 f32 (n_use, n_use)
    -> (eff_m, n_use) (`jx grm` tiled route, python/janusx/script/grm.py:985)."""
    k, eff_m, n_use = _dense_grm_from_meta_f32(prefix, row_source_indices, row_flip, row_maf, n_total_sites, sample_indices,
                                               method, mmap_window_mb, "tiled")
    _write_npy_f32(out_npy_path, k)
    if progress_callback is not None:
        progress_callback(eff_m, eff_m)
    return eff_m, n_use


def rust_eigh_from_array_f64(a, threads=0, driver=None, jobz="V", require_lapack=False, diag_shift=0.0):
    """src/math/eigh.rs:1621-1703 -> 10-tuple (evals asc, evecs (columns) or None, blas_backend, evd_backend,
    n, threads_before, threads_in_stage, threads_after, lapack_used, elapsed_s)."""
    a = _c(a, np.float64)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise RuntimeError("matrix must be square")
    n = int(a.shape[0])
    t0 = time.perf_counter()
    evals = np.empty(n, dtype=np.float64)
    want = str(jobz).upper() != "N"
    evecs = np.empty((n, n), dtype=np.float64) if want else None
    check(lib().jx_eigh_f64(_p(a), n, float(diag_shift), _p(evals), _p(evecs)))
    # which solver ran (csrc/eigh.cpp): rocSOLVER below n = 256 or on request, else the own one- or two-stage reduction
    # + own divide and conquer (jxg_last_kernel_ms(10) = 1 after a two-stage decomposition)
    if n < 256 or os.environ.get("JXGPU_EIGH", "") == "rocsolver":
        evd = "rocsolver_dsyevd"
    elif float(lib().jxg_last_kernel_ms(10)) > 0.5:
        evd = "jxgpu_sy2sb_sb2st_stedc"
    else:
        evd = "jxgpu_sytrd_stedc"
    return (evals, evecs, "rocblas", evd, n, 0, 0, 0, True, time.perf_counter() - t0)


def rust_eigh_from_array_f64_inplace(a, threads=0, driver=None, jobz="V", require_lapack=False):
    """src/math/eigh.rs:1883-1962.  Same 10-tuple as `rust_eigh_from_array_f64`.  Despite its name the reference's function
    does NOT write into its argument: it takes `a.readonly()` (:1914), hands a row-major copy-on-write view to
    `symmetric_eigh_f64_row_major_with_driver` (:1918, :1925) and returns freshly allocated arrays (:1943-1955) -- "in place"
    there is about not copying a contiguous input on the way IN.  What it does differently from the plain entry point, and what
    this mirror reproduces: no `diag_shift` argument, an empty or non-square input is refused with its own message (:1907-1912),
    and a Fortran-ordered / strided input is accepted (read through `_c`).  The caller's array is left untouched."""
    arr = np.asarray(a)
    nrows = int(arr.shape[0]) if arr.ndim >= 1 else 0
    ncols = int(arr.shape[1]) if arr.ndim >= 2 else 0
    if arr.ndim != 2 or nrows == 0 or ncols == 0 or nrows != ncols:
        raise RuntimeError(f"rust_eigh_from_array_f64_inplace expects a non-empty square matrix; got shape=({nrows}, {ncols})")
    return rust_eigh_from_array_f64(arr, threads, driver, jobz, require_lapack, 0.0)


def rust_eigh_from_matrix_file_f64(path, threads=0, driver=None, jobz="V", require_lapack=False, diag_shift=0.0):
    """src/math/eigh.rs:1707-1786: `.npy` (f32/f64 C-order) or whitespace text matrix."""
    a = np.load(path) if str(path).endswith(".npy") else np.loadtxt(path)
    return rust_eigh_from_array_f64(np.asarray(a, dtype=np.float64), threads, driver, jobz, require_lapack, diag_shift)


def rust_eigh_from_matrix_file_subset_f64(path, subset, threads=0, driver=None, jobz="V", require_lapack=False,
                                          diag_shift=0.0):
    """src/math/eigh.rs:1788-1880."""
    a = np.load(path, mmap_mode="r") if str(path).endswith(".npy") else np.loadtxt(path)
    ix = np.asarray(subset, dtype=np.int64)
    sub = np.asarray(a[np.ix_(ix, ix)], dtype=np.float64)
    return rust_eigh_from_array_f64(sub, threads, driver, jobz, require_lapack, diag_shift)


def rust_sgemm_backend():
    return "hip-mfma-gfx950"


def rust_eigh_lapack_backend():
    return "rocsolver"


def rust_blas_get_num_threads():
    return 0


def rust_blas_set_num_threads(n):
    return None


# ------------------------------------------------------------------------------------------------
# null model helpers  (src/stats/reml.rs)
# ------------------------------------------------------------------------------------------------

def lmm_rotate_x_y_with_ut_f64(u_t, x, y, threads=0):
    """src/stats/reml.rs:107-198 -> (f64 (n,q), f64 (n,1))."""
    y = _c(y, np.float64).ravel()
    n = int(y.shape[0])
    if n == 0:
        raise RuntimeError("y must not be empty")
    x = _c(x, np.float64)
    if x.ndim != 2:
        raise RuntimeError("x must be 2D (n, q)")
    if x.shape[0] != n:
        raise RuntimeError(f"x rows must equal len(y): rows={x.shape[0]}, len(y)={n}")
    u_t = _c(u_t, np.float32)
    if u_t.ndim != 2 or u_t.shape != (n, n):
        raise RuntimeError("u_t must be shape (n, n) and row-major U^T")
    q = int(x.shape[1])
    ox = np.empty((n, q), dtype=np.float64)
    oy = np.empty((n, 1), dtype=np.float64)
    check(lib().jx_lmm_rotate_x_y_with_ut_f64(_p(u_t), n, _p(x), q, _p(y), _p(ox), _p(oy)))
    return ox, oy


def lmm_rotate_y_with_ut_f64(u_t, y, threads=0):
    """src/stats/reml.rs:200-250 -> y_rot f64 (n): row i = <u_t[i, :], y> in f64 (what `workflow_model_packed.py:6309` calls
    per trait once X has been rotated)."""
    y = _c(y, np.float64).ravel()
    n = int(y.shape[0])
    if n == 0:
        raise RuntimeError("y must not be empty")
    u_t = _c(u_t, np.float32)
    if u_t.ndim != 2:
        raise RuntimeError("u_t must be 2D (n, n)")
    if u_t.shape != (n, n):
        raise RuntimeError("u_t must be shape (n, n) and row-major U^T")
    x = np.zeros((n, 1), dtype=np.float64)
    ox = np.empty((n, 1), dtype=np.float64)
    oy = np.empty((n, 1), dtype=np.float64)
    check(lib().jx_lmm_rotate_x_y_with_ut_f64(_p(u_t), n, _p(x), 1, _p(y), _p(ox), _p(oy)))
    return oy.ravel()


def _null_args(s, xcov, y_rot):
    s = _c(s, np.float64).ravel()
    xcov = _c(xcov, np.float64)
    y = _c(y_rot, np.float64).ravel()
    n = int(y.shape[0])
    if xcov.ndim != 2 or xcov.shape[0] != n:
        raise RuntimeError("Xcov.n_rows must equal len(y_rot)")
    if s.shape[0] != n:
        raise RuntimeError("len(S) must equal len(y_rot)")
    return s, xcov, y, n, int(xcov.shape[1])


def lmm_reml_null_f32(s, xcov, y_rot, low, high, max_iter=50, tol=1e-2):
    """src/stats/reml.rs:570-616 -> (lbd, ml, reml)."""
    s, xcov, y, n, p = _null_args(s, xcov, y_rot)
    if low >= high:
        raise RuntimeError("low must be < high")
    out = np.zeros(3, dtype=np.float64)
    check(lib().jx_lmm_reml_null(_p(s), _p(xcov), _p(y), n, p, float(low), float(high), int(max_iter), float(tol),
                                 _p(out)))
    return float(out[0]), float(out[1]), float(out[2])


def ml_loglike_null_f32(s, xcov, y_rot, log10_lbd):
    """src/stats/reml.rs:618-646 -> ML profile log-likelihood of the null model at log10 lambda (-1e8 on failure)."""
    s, xcov, y, n, p = _null_args(s, xcov, y_rot)
    out = np.zeros(1, dtype=np.float64)
    check(lib().jx_ml_loglike_null(_p(s), _p(xcov), _p(y), n, p, float(log10_lbd), _p(out)))
    return float(out[0])


def _loglike_null(s, xcov, y_rot, log10_lbd):
    """(ml, reml) of the null model at log10 lambda, one device launch (jxg_lmm_loglike_null)."""
    import torch
    s, xcov, y, n, p = _null_args(s, xcov, y_rot)
    dev = torch.device("cuda", torch.cuda.current_device())
    ds, dx, dy = (torch.from_numpy(a).to(dev) for a in (s, xcov, y))
    o = torch.empty(2, dtype=torch.float64, device=dev)
    check(lib().jxg_lmm_loglike_null(ds.data_ptr(), dx.data_ptr(), dy.data_ptr(), n, p, float(log10_lbd),
                                     o.data_ptr(), torch.cuda.current_stream().cuda_stream))
    h = o.cpu().numpy()
    return float(h[0]), float(h[1])


# ------------------------------------------------------------------------------------------------
# exact per-SNP scan  (src/stats/lmm.rs)
# ------------------------------------------------------------------------------------------------

def _chunk(s, xcov, y_rot, low, high, chunk, u_t, max_iter, tol, nullml, what):
    s, xcov, y, n, p = _null_args(s, xcov, y_rot)
    g = _c(chunk, np.float32)
    if g.ndim != 2 or g.shape[1] != n:
        raise RuntimeError(f"{what} must be (m_chunk, n)")
    if u_t is not None:
        u_t = _c(u_t, np.float32)
        if u_t.shape != (n, n):
            raise RuntimeError("u_t must be (n, n) and row-major U^T")
    if low >= high:
        raise RuntimeError("low must be < high")
    m = int(g.shape[0])
    cols = 4 if nullml is not None else 3
    out = np.zeros((m, cols), dtype=np.float64)
    check(lib().jx_lmm_reml_chunk(_p(s), _p(xcov), _p(y), n, p, float(low), float(high), _p(g), m, _p(u_t),
                                  int(max_iter), float(tol), 1 if nullml is not None else 0,
                                  float(nullml if nullml is not None else 0.0), _p(out)))
    return out


def lmm_reml_chunk_f32(s, xcov, y_rot, low, high, g_rot_chunk, max_iter=50, tol=1e-2, threads=0, nullml=None):
    """src/stats/lmm.rs:333-335 (already rotated rows) -> f64 (m, 3 or 4)."""
    return _chunk(s, xcov, y_rot, low, high, g_rot_chunk, None, max_iter, tol, nullml, "g_rot_chunk")


def lmm_reml_chunk_from_snp_f32(s, xcov, y_rot, low, high, snp_chunk, u_t, max_iter=50, tol=1e-2, threads=0,
                                nullml=None, rotate_block_rows=256):
    """src/stats/lmm.rs:1479-1630 (rotate then scan, no warm start) -> f64 (m, 3 or 4)."""
    return _chunk(s, xcov, y_rot, low, high, snp_chunk, u_t, max_iter, tol, nullml, "snp_chunk")


def _resolve_warm_start(warm_start, route_env=None):
    """`warm_start` of the exact-scan entry points -> "chain" | "none".  None = the reference's default: the chain
    (`carry_warm_start`, src/stats/lmm.rs:134-161), switched off by a truthy JX_LMM_UNIFIED_NO_WARM_START.  The reference reads
    that variable in the BED route only (:2627; its packed routes pass `true, true` unconditionally, :3244-3245, :3609-3610);
    here one setting gives every exact-scan entry point and `jx gwas -lmm` the no-chain scan."""
    from .stats import env_truthy
    if warm_start is None:
        return "none" if (route_env and env_truthy(route_env)) else "chain"
    w = str(warm_start).strip().lower()
    if w in ("chain", "carry", "on", "true", "1"):
        return "chain"
    if w in ("none", "off", "false", "0"):
        return "none"
    raise RuntimeError("warm_start must be 'chain' or 'none'")


def _assoc_packed(packed, n_samples, row_flip, row_maf, s, xcov, y_rot, u_t, sample_indices, row_indices, model,
                  low, high, max_iter, tol, warm, init, nullml=None, progress_callback=None, progress_every=0,
                  genetic_model="add", chain_off=None):
    from .stats import genetic_model_code
    gm = genetic_model_code(genetic_model)          # `PackedGeneticModel::parse` (src/decode/decode.rs:107-119)
    s, xcov, y, n, p = _null_args(s, xcov, y_rot)
    if row_indices is not None:
        ri = np.asarray(row_indices, dtype=np.int64)
        if _is_device_tensor(packed):
            import torch
            packed = packed[torch.from_numpy(ri).to(packed.device)]
        else:
            packed = np.ascontiguousarray(_c(packed, np.uint8)[ri])
    packed, pk_ptr, m = _payload(packed, n_samples)           # host array or device tensor
    flip = _c(np.asarray(row_flip).astype(np.uint8), np.uint8).ravel()
    maf = _c(row_maf, np.float32).ravel()
    u_t = _c(u_t, np.float32)
    if u_t.shape != (n, n):
        raise RuntimeError("u_t must be (n, n) and row-major U^T")
    idx, n_sel = _opt_idx(sample_indices)
    n_eff = n_sel if idx is not None else int(n_samples)
    if n_eff != n:
        raise RuntimeError(f"selected sample count {n_eff} != len(y_rot) {n}")
    out = np.zeros((m, 6 if int(model) == 2 else (4 if nullml is not None else 3)), dtype=np.float64)
    if chain_off is not None and int(model) == 0 and m > 0:
        co = np.ascontiguousarray(chain_off, dtype=np.int64)
        with _progress_hook(progress_callback, progress_every):
            check(lib().jx_assoc_packed_chain(pk_ptr, m, int(n_samples), _p(flip), _p(maf), _p(s), _p(xcov), _p(y), _p(u_t), p,
                                              _p(idx), n_sel, float(low), float(high), int(max_iter), float(tol), int(warm),
                                              float(init), 1 if nullml is not None else 0,
                                              float(nullml if nullml is not None else 0.0), _p(out), gm, _p(co), len(co) - 1))
        return out
    with _progress_hook(progress_callback, progress_every):
        check(lib().jx_assoc_packed_gm(pk_ptr, m, int(n_samples), _p(flip), _p(maf), _p(s), _p(xcov), _p(y), _p(u_t), p,
                                       _p(idx), n_sel, int(model), float(low), float(high), int(max_iter), float(tol),
                                       int(warm), float(init), 1 if nullml is not None else 0,
                                       float(nullml if nullml is not None else 0.0), _p(out), gm))
    return out


def lmm_reml_assoc_packed_f32(packed, n_samples, row_flip, row_maf, s, xcov, y_rot, u_t, sample_indices=None,
                              row_indices=None, low=-5.0, high=5.0, max_iter=50, tol=1e-2, threads=0, model="add",
                              progress_callback=None, progress_every=0, nullml=None, init_log10_lbd=None,
                              rotate_block_rows=256, warm_start=None, warm_chain_pieces=1):
    """src/stats/lmm.rs:3040-3362 -> f64 (m, 3), or (m, 4) with `nullml` (plrt column, lmm.rs:202-330).

    Warm start (`warm_start`, None = the reference's default for this entry point = "chain"): the reference runs this scan
    with `seed_with_init_guess = carry_warm_start = true` (lmm.rs:3244-3245): inside a block of `rotate_block_rows` rows
    (blocks restart at every `progress_every` rows, :3253-3256) each SNP's Brent starts from the optimum of the valid SNP before it,
    the first one from `init_log10_lbd` or the interval midpoint (:134-161).  Here the chains run in parallel, one wave per
    chain, the rows of a chain in order -- bit-for-bit the sequential semantics.  rayon additionally cuts a block into pieces
    with a fresh state each (2 T pieces on T idle threads, more under work stealing: the reference's own output depends on
    scheduling); `warm_chain_pieces` (power of two) reproduces that halving deterministically, 1 = one chain per block.
    warm_start="none": every SNP starts from `init_log10_lbd` when given, else from the midpoint (the core-API contract,
    lmm.rs:1577-1579; what JX_LMM_UNIFIED_NO_WARM_START selects on the BED route)."""
    from .stats import warm_chain_blocks_packed, warm_chain_offsets
    if low >= high:
        raise RuntimeError("low must be < high")
    if not (np.isfinite(tol) and tol > 0):
        raise RuntimeError("tol must be positive and finite")
    warm, init = 0, 0.0
    if init_log10_lbd is not None and np.isfinite(init_log10_lbd):
        warm, init = 1, float(min(max(init_log10_lbd, low), high))
    chain_off = None
    if _resolve_warm_start(warm_start, "JX_LMM_UNIFIED_NO_WARM_START") == "chain":
        m_rows = int(len(row_indices)) if row_indices is not None else int(packed.shape[0])
        chain_off = warm_chain_offsets(warm_chain_blocks_packed(m_rows, rotate_block_rows, progress_every), m_rows,
                                       warm_chain_pieces)
    return _assoc_packed(packed, n_samples, row_flip, row_maf, s, xcov, y_rot, u_t, sample_indices, row_indices, 0,
                         low, high, max_iter, tol, warm, init, nullml, progress_callback, progress_every,
                         genetic_model=model, chain_off=chain_off)


# ------------------------------------------------------------------------------------------------
# fixed-lambda scan  (src/stats/fvlmm.rs)
# ------------------------------------------------------------------------------------------------

def _fv_chunk(s, xcov, y_rot, log10_lbd, chunk, u_t, nullml, what):
    s, xcov, y, n, p = _null_args(s, xcov, y_rot)
    g = _c(chunk, np.float32)
    if g.ndim != 2 or g.shape[1] != n:
        raise RuntimeError(f"{what} must be (m_chunk, n)")
    if u_t is not None:
        u_t = _c(u_t, np.float32)
        if u_t.shape != (n, n):
            raise RuntimeError("u_t must be (n, n) and row-major U^T")
    m = int(g.shape[0])
    out = np.zeros((m, 4 if nullml is not None else 3), dtype=np.float64)
    check(lib().jx_fvlmm_assoc_chunk(_p(s), _p(xcov), _p(y), n, p, float(log10_lbd), _p(g), m, _p(u_t),
                                     1 if nullml is not None else 0, float(nullml if nullml is not None else 0.0),
                                     _p(out)))
    return out


def fvlmm_assoc_chunk_f32(s, xcov, y_rot, log10_lbd, g_rot_chunk, threads=0, nullml=None):
    """src/stats/fvlmm.rs:1941-1994 -> f64 (m, 3 or 4)."""
    return _fv_chunk(s, xcov, y_rot, log10_lbd, g_rot_chunk, None, nullml, "g_rot_chunk")


def fvlmm_assoc_chunk_from_snp_f32(s, xcov, y_rot, log10_lbd, snp_chunk, u_t, threads=0, nullml=None,
                                   rotate_block_rows=512):
    """src/stats/fvlmm.rs:2114-2262 -> f64 (m, 3)."""
    return _fv_chunk(s, xcov, y_rot, log10_lbd, snp_chunk, u_t, nullml, "snp_chunk")


def _fixed_lambda_checks(s, xcov, y_rot, log10_lbd):
    """Argument checks shared by the fixed-lambda entry points (src/stats/lmm.rs:2025-2066, src/stats/fvlmm.rs:1816-1836)."""
    s_, xcov_, y_, n, p = _null_args(s, xcov, y_rot)
    if n <= p + 1:
        raise RuntimeError("n must be > p_cov+1")
    with np.errstate(over="ignore", under="ignore"):
        lbd = float(np.float64(10.0) ** np.float64(log10_lbd))     # powf semantics: inf / 0 instead of an exception
    if not (np.isfinite(lbd) and lbd > 0.0):
        raise RuntimeError("invalid log10_lbd")
    if np.any(s_ + lbd <= 0.0):
        raise RuntimeError("non-positive s[i]+lbd")
    return n, p


def lmm_assoc_chunk_f32(s, xcov, y_rot, log10_lbd, g_rot_chunk, threads=0, nullml=None):
    """src/stats/lmm.rs:2010-2224: Wald statistics of already rotated rows at ONE given lambda (the `lmm_assoc` wrapper of
    python/janusx/pyBLUP/assoc.py:1305-1345) -> f64 (m, 3 or 4).  The same sums as the fixed-lambda scan of `-fvlmm`
    (c = X'Wg, d = g'Wg, e = g'Wy with W = 1 / (s + lambda) held in f32); a row whose Schur complement is <= 1e-12 is
    (NaN, NaN, NaN) and its plrt entry keeps the 0.0 the output was allocated with (:2155-2160)."""
    _fixed_lambda_checks(s, xcov, y_rot, log10_lbd)
    out = _fv_chunk(s, xcov, y_rot, log10_lbd, g_rot_chunk, None, nullml, "g_rot_chunk")
    if nullml is not None:
        out[np.isnan(out[:, 2]), 3] = 0.0
    return out


def lmm_assoc_chunk_from_snp_f32(s, xcov, y_rot, log10_lbd, snp_chunk, u_t, threads=0, nullml=None,
                                 rotate_block_rows=512):
    """src/stats/lmm.rs:2226-2486: `lmm_assoc_chunk_f32` behind the rotation of raw SNP rows -> f64 (m, 3 or 4)."""
    _fixed_lambda_checks(s, xcov, y_rot, log10_lbd)
    out = _fv_chunk(s, xcov, y_rot, log10_lbd, snp_chunk, u_t, nullml, "snp_chunk")
    if nullml is not None:
        out[np.isnan(out[:, 2]), 3] = 0.0
    return out


class FvLmmAssocCache:
    """`FvLmmAssocCache` (src/stats/fvlmm.rs:1808-1849): the lambda-only state of the fixed-lambda scan (W, P y, W X, the
    Cholesky factor of X'WX, y'Py, df, log det V) prepared once per trait.  Here the handle keeps the host inputs; the
    device state is rebuilt per call by one small kernel (`jxg_fvlmm_prepare`)."""

    def __init__(self, s, xcov, y_rot, log10_lbd):
        self.n, self.p = _fixed_lambda_checks(s, xcov, y_rot, log10_lbd)
        self.s, self.xcov, self.y_rot = (np.array(_c(a, np.float64), copy=True) for a in (s, xcov, y_rot))
        self.log10_lbd = float(log10_lbd)
        self.lbd = float(np.float64(10.0) ** np.float64(self.log10_lbd))


def fvlmm_assoc_prepare_cache_f32(s, xcov, y_rot, log10_lbd):
    """src/stats/fvlmm.rs:1808-1849 -> cache handle for `fvlmm_assoc_chunk_with_cache_f32`."""
    return FvLmmAssocCache(s, xcov, y_rot, log10_lbd)


def fvlmm_assoc_chunk_with_cache_f32(cache, g_rot_chunk, threads=0, nullml=None):
    """src/stats/fvlmm.rs:1920-1939 -> f64 (m, 3 or 4) for already rotated rows."""
    if not isinstance(cache, FvLmmAssocCache):
        raise TypeError("cache must come from fvlmm_assoc_prepare_cache_f32")
    return _fv_chunk(cache.s, cache.xcov, cache.y_rot, cache.log10_lbd, g_rot_chunk, None, nullml, "g_rot_chunk")


def fvlmm_assoc_chunk_from_snp_with_cache_f32(cache, snp_chunk, u_t, threads=0, nullml=None, rotate_block_rows=512):
    """src/stats/fvlmm.rs:1997-2112 -> f64 (m, 3 or 4) for raw SNP rows (rotation + scan)."""
    if not isinstance(cache, FvLmmAssocCache):
        raise TypeError("cache must come from fvlmm_assoc_prepare_cache_f32")
    return _fv_chunk(cache.s, cache.xcov, cache.y_rot, cache.log10_lbd, snp_chunk, u_t, nullml, "snp_chunk")


def fvlmm_assoc_chunk_from_snp_to_tsv_f32(s, xcov, y_rot, log10_lbd, snp_chunk, u_t, chrom, pos, snp, allele0, allele1, maf,
                                          miss, threads=0, nullml=None, rotate_block_rows=512, progress_callback=None,
                                          progress_every=0):
    """src/stats/fvlmm.rs:2266-2480: rotation + fixed-lambda scan of a dense SNP chunk, the result rows already formatted
    -> (list of byte blocks of `rotate_block_rows` TSV rows each, no header; rows).  The streaming `-fvlmm` workflow writes
    the blocks to its result file (python/janusx/assoc/workflow_model_stream.py:1678)."""
    import tempfile
    from .tsv import _native_rows
    n, _p_cov = _fixed_lambda_checks(s, xcov, y_rot, log10_lbd)
    g = _c(snp_chunk, np.float32)
    if g.ndim != 2 or g.shape[1] != n:
        raise RuntimeError("snp_chunk must be (m, n)")
    m = int(g.shape[0])
    if m == 0:
        return [], 0
    if not all(len(a) == m for a in (chrom, pos, snp, allele0, allele1, maf, miss)):
        raise RuntimeError("TSV metadata length mismatch with snp_chunk rows")
    stats = _fv_chunk(s, xcov, y_rot, log10_lbd, g, u_t, nullml, "snp_chunk")
    br = max(int(rotate_block_rows), 1)
    blocks = []
    fd, tmp = tempfile.mkstemp(prefix="jx_fvlmm_chunk_", suffix=".tsv")
    os.close(fd)
    try:
        for r0 in range(0, m, br):
            r1 = min(r0 + br, m)
            open(tmp, "wb").close()
            _native_rows(tmp, chrom[r0:r1], pos[r0:r1], snp[r0:r1], allele0[r0:r1], allele1[r0:r1], maf[r0:r1],
                         miss[r0:r1], stats[r0:r1], True, False, False)
            with open(tmp, "rb") as fh:
                blocks.append(fh.read())
            if progress_callback is not None and progress_every:
                progress_callback(r1, m)
    finally:
        try:
            os.remove(tmp)
        except OSError:
            pass
    if progress_callback is not None:
        progress_callback(m, m)
    return blocks, m


def fvlmm_assoc_packed_f32(packed, n_samples, row_flip, row_maf, s, xcov, y_rot, u_t, log10_lbd,
                           sample_indices=None, row_indices=None, threads=0, progress_callback=None,
                           progress_every=0, rotate_block_rows=512, nullml=None):
    """Array-returning core of `fvlmm_assoc_packed_f32_to_tsv` (src/stats/fvlmm.rs:4958-5190) with a
    caller-rotated null model -> f64 (m, 3 or 4)."""
    return _assoc_packed(packed, n_samples, row_flip, row_maf, s, xcov, y_rot, u_t, sample_indices, row_indices, 1,
                         float(log10_lbd), float(log10_lbd) + 1.0, 0, 1e-2, 0, 0.0, nullml, progress_callback,
                         progress_every)


# ------------------------------------------------------------------------------------------------
# BED -> QC -> scan -> TSV entry points (src/stats/lmm.rs:2488-2751, src/stats/fvlmm.rs:2482-2526,
# orchestration of src/stats/lmm.rs:975-1477 `run_unified_bed_scan_to_tsv_common`)
# ------------------------------------------------------------------------------------------------

def bed_row_counts(packed, n_samples, sample_indices=None):
    """(m,3) int32 (missing, het, hom_alt) over the selected samples (src/io/gfreader.rs:1378-1395)."""
    _pk, pk_ptr, m = _payload(packed, n_samples)          # host array or device tensor
    idx, n_sel = _opt_idx(sample_indices)
    out = np.zeros((m, 3), dtype=np.int32)
    check(lib().jx_row_counts(pk_ptr, m, int(n_samples), _p(idx), n_sel, _p(out)))
    return out


def _bed_scan_to_tsv(bed_prefix, out_tsv, s, xcov, y_rot, u_t, maf_thr, miss_thr, het_thr, genetic_model, snps_only,
                     sample_ids, row_indices, row_flip, row_missing, row_maf, mode, low, high, max_iter, tol, nullml,
                     init_log10_lbd, progress_callback, progress_every=0, mmap_window_mb=None, warm_chain=None):
    # warm_chain: None, or (rotate_block_rows, pieces) -- the exact scan along the reference's warm-start chains
    import torch
    from . import stats as st
    from .bed import read_fam_ids, snps_only_mask, stage_bed_payload
    from .tsv import write_assoc_tsv
    st.genetic_model_code(genetic_model)             # add / dom / rec / het (src/decode/decode.rs:100-147); anything else raises
    if nullml is not None and not np.isfinite(nullml):
        raise RuntimeError("nullml must be finite when provided")
    s_, xcov_, y_, n, p = _null_args(s, xcov, y_rot)
    if n <= p + 1:
        raise RuntimeError("n must be > p+1")
    # payload staged to HBM in windows of `mmap_window_mb` MiB (the reference's WindowedBedMatrix role); the counts, the row
    # gather and the scan below take the device tensor in place
    packed, n_fam, bim = stage_bed_payload(_bed_prefix(bed_prefix), mmap_window_mb)
    if sample_ids is not None:
        fam = read_fam_ids(_bed_prefix(bed_prefix))
        pos = {sid: i for i, sid in enumerate(fam)}
        try:
            sidx = np.array([pos[str(x)] for x in sample_ids], dtype=np.int64)
        except KeyError as e:
            raise RuntimeError(f"sample id not found in FAM: {e}") from None
    else:
        sidx = None
    n_sel = n_fam if sidx is None else int(sidx.shape[0])
    if n_sel != n:
        raise RuntimeError(f"selected sample count {n_sel} != len(y_rot) {n}")
    prepared = [row_indices, row_flip, row_missing, row_maf]
    if any(v is not None for v in prepared) and not all(v is not None for v in prepared):
        raise RuntimeError("prepared row metadata must provide all or none of: row_indices, row_flip, row_missing, row_maf")
    m = int(packed.shape[0])
    if row_indices is not None:
        rows = np.asarray(row_indices, dtype=np.int64)
        flip = np.asarray(row_flip).astype(bool)
        af = np.asarray(row_maf, dtype=np.float32)
        miss = np.asarray(row_missing, dtype=np.float32)
    else:
        counts = bed_row_counts(packed, n_fam, sidx)
        keep, af_all, miss_all = st.gwas_scan_row_stats(counts, n, maf_thr, miss_thr, het_thr)
        if snps_only:
            keep &= snps_only_mask(bim)
        rows = np.nonzero(keep)[0]
        flip = np.zeros(len(rows), dtype=bool)
        af = af_all[rows]
        miss = miss_all[rows]
    pk = packed if len(rows) == m and np.array_equal(rows, np.arange(m)) else \
        packed[torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)).to(packed.device)]
    del packed
    if mode == "lmm":
        warm, init = 0, 0.0
        if init_log10_lbd is not None and np.isfinite(init_log10_lbd):
            warm, init = 1, float(min(max(init_log10_lbd, low), high))
        chain_off = None
        if warm_chain is not None:
            from .stats import warm_chain_blocks_bed, warm_chain_offsets
            # scan units: the prepared row list, or every SNP row of the file (src/stats/lmm.rs:1103-1106, 1121-1145)
            units = np.arange(len(rows), dtype=np.int64) if row_indices is not None else rows
            n_units = len(rows) if row_indices is not None else m
            chain_off = warm_chain_offsets(warm_chain_blocks_bed(units, n_units, warm_chain[0]), len(rows), warm_chain[1])
        res = _assoc_packed(pk, n_fam, flip, af, s_, xcov_, y_, u_t, sidx, None, 0, low, high, max_iter, tol, warm, init,
                            nullml, progress_callback, progress_every, genetic_model=genetic_model, chain_off=chain_off)
    elif mode == "lmm2":
        warm, init = 0, 0.0
        if init_log10_lbd is not None and np.isfinite(init_log10_lbd):
            warm, init = 1, float(min(max(init_log10_lbd, low), high))
        res = _assoc_packed(pk, n_fam, flip, af, s_, xcov_, y_, u_t, sidx, None, 2, low, high, max_iter, tol, warm, init,
                            nullml, progress_callback, progress_every, genetic_model=genetic_model)
    else:
        res = _assoc_packed(pk, n_fam, flip, af, s_, xcov_, y_, u_t, sidx, None, 1, float(low), float(low) + 1.0, 0,
                            1e-2, 0, 0.0, nullml, progress_callback, progress_every, genetic_model=genetic_model)
    chrom = [bim.chrom[j] for j in rows]
    posv = [bim.pos[j] for j in rows]
    snp = [bim.snp[j] for j in rows]
    a0 = [bim.a0[j] for j in rows]
    a1 = [bim.a1[j] for j in rows]
    written = write_assoc_tsv(out_tsv, chrom, posv, snp, a0, a1, af, miss, res)
    return written


def lmm_reml_assoc_bed_to_tsv_f32(bed_prefix, out_tsv, s, xcov, y_rot, u_t, maf_thr, miss_thr, het_thr,
                                  genetic_model="add", snps_only=False, sample_ids=None, row_indices=None,
                                  row_flip=None, row_missing=None, row_maf=None, low=-5.0, high=5.0, max_iter=30,
                                  tol=1e-2, threads=0, nullml=None, init_log10_lbd=None, rotate_block_rows=512,
                                  progress_callback=None, progress_every=0, mmap_window_mb=None, warm_start=None,
                                  warm_chain_pieces=1):
    """src/stats/lmm.rs:2488-2751 (the default `jx gwas -lmm` kernel call) -> rows written.
    Warm start: as the reference, the chain is ON unless JX_LMM_UNIFIED_NO_WARM_START is truthy (:2627) or
    warm_start="none"; a chain is the kept rows of one chunk of `rotate_block_rows` scan units (:1121-1145), see
    `lmm_reml_assoc_packed_f32` for the semantics and `warm_chain_pieces`."""
    if low >= high:
        raise RuntimeError("low must be < high")
    if not (np.isfinite(tol) and tol > 0):
        raise RuntimeError("tol must be positive and finite")
    chain = _resolve_warm_start(warm_start, "JX_LMM_UNIFIED_NO_WARM_START") == "chain"
    return _bed_scan_to_tsv(bed_prefix, out_tsv, s, xcov, y_rot, u_t, maf_thr, miss_thr, het_thr, genetic_model,
                            snps_only, sample_ids, row_indices, row_flip, row_missing, row_maf, "lmm", low, high,
                            max_iter, tol, nullml, init_log10_lbd, progress_callback, progress_every, mmap_window_mb,
                            warm_chain=(int(rotate_block_rows), int(warm_chain_pieces)) if chain else None)


def lmm_reml_lmm2_chunk_from_snp_f32(s, xcov, y_rot, low, high, snp_chunk, u_t, nullml, max_iter=50, tol=1e-2,
                                     threads=0, rotate_block_rows=256):
    """src/stats/lmm.rs:1632-1760 -> f64 (m, 6) = [beta, se, pwald, lambda_reml, ml_alt, plrt] (LMM2: REML Wald test
    plus an ML likelihood-ratio test from a second Brent on `ml_loglike`, lmm.rs:202-330); no warm start."""
    s, xcov, y, n, p = _null_args(s, xcov, y_rot)
    g = _c(snp_chunk, np.float32)
    if g.ndim != 2 or g.shape[1] != n:
        raise RuntimeError("snp_chunk must be (m_chunk, n)")
    u_t = _c(u_t, np.float32)
    if u_t.shape != (n, n):
        raise RuntimeError("u_t must be (n, n) and row-major U^T")
    if low >= high:
        raise RuntimeError("low must be < high")
    if not np.isfinite(nullml):
        raise RuntimeError("nullml must be finite")
    m = int(g.shape[0])
    out = np.zeros((m, 6), dtype=np.float64)
    check(lib().jx_lmm2_chunk(_p(s), _p(xcov), _p(y), n, p, float(low), float(high), _p(g), m, _p(u_t), float(nullml),
                              int(max_iter), float(tol), _p(out)))
    return out


def lmm_reml_lmm2_assoc_bed_to_tsv_f32(bed_prefix, out_tsv, s, xcov, y_rot, u_t, maf_thr, miss_thr, het_thr,
                                       genetic_model="add", snps_only=False, sample_ids=None, row_indices=None,
                                       row_flip=None, row_missing=None, row_maf=None, low=-5.0, high=5.0,
                                       max_iter=30, tol=1e-2, threads=0, nullml=None, init_log10_lbd_reml=None,
                                       init_log10_lbd_ml=None, rotate_block_rows=512, progress_callback=None,
                                       progress_every=0, mmap_window_mb=None):
    """src/stats/lmm.rs:2779-3037 (`jx gwas -lmm2`) -> rows written; TSV schema Lmm2_6
    (`... pwald lambda ml plrt`, src/io/assoc2tsv.rs:54-56).  Without `nullml` the null ML is fitted first by Brent
    on -ml_loglike seeded with init_log10_lbd_ml or init_log10_lbd_reml (lmm.rs:2902-2921).  Deterministic start per
    SNP: init_log10_lbd_reml, else init_log10_lbd_ml, else the interval midpoint (see `lmm_reml_assoc_packed_f32`)."""
    if low >= high:
        raise RuntimeError("low must be < high")
    if not (np.isfinite(tol) and tol > 0):
        raise RuntimeError("tol must be positive and finite")
    if nullml is not None:
        if not np.isfinite(nullml):
            raise RuntimeError("nullml must be finite when provided")
        nullml_val = float(nullml)
    else:
        s_, xcov_, y_, n, p = _null_args(s, xcov, y_rot)
        init = init_log10_lbd_ml if init_log10_lbd_ml is not None else init_log10_lbd_reml
        o2 = np.zeros(2, dtype=np.float64)
        check(lib().jx_lmm2_null_ml(_p(s_), _p(xcov_), _p(y_), n, p, float(low), float(high), int(max_iter), float(tol),
                                    1 if init is not None else 0, float(init if init is not None else 0.0), _p(o2)))
        nullml_val = float(o2[1])
        if not np.isfinite(nullml_val):
            raise RuntimeError("failed to optimize null ML for LMM2 unified scan")
    init_scan = init_log10_lbd_reml if init_log10_lbd_reml is not None else init_log10_lbd_ml
    return _bed_scan_to_tsv(bed_prefix, out_tsv, s, xcov, y_rot, u_t, maf_thr, miss_thr, het_thr, genetic_model,
                            snps_only, sample_ids, row_indices, row_flip, row_missing, row_maf, "lmm2", low, high,
                            max_iter, tol, nullml_val, init_scan, progress_callback, progress_every, mmap_window_mb)


def fvlmm_assoc_bed_to_tsv_f32(bed_prefix, out_tsv, s, xcov, y_rot, log10_lbd, u_t, maf_thr, miss_thr, het_thr,
                               genetic_model="add", snps_only=False, sample_ids=None, row_indices=None,
                               row_flip=None, row_missing=None, row_maf=None, threads=0, nullml=None,
                               rotate_block_rows=512, progress_callback=None, progress_every=0, mmap_window_mb=None):
    """src/stats/fvlmm.rs:2482-2526 (the default `jx gwas -fvlmm` kernel call) -> (rows, pve, log_det_v)."""
    rows = _bed_scan_to_tsv(bed_prefix, out_tsv, s, xcov, y_rot, u_t, maf_thr, miss_thr, het_thr, genetic_model,
                            snps_only, sample_ids, row_indices, row_flip, row_missing, row_maf, "fvlmm",
                            float(log10_lbd), None, 0, 1e-2, nullml, None, progress_callback, progress_every, mmap_window_mb)
    s_ = np.asarray(s, dtype=np.float64).ravel()
    lbd = 10.0 ** float(log10_lbd)
    vg = float(np.mean(np.clip(s_, 0.0, None)))
    pve = vg / (vg + lbd) if (vg + lbd) > 0 else float("nan")
    return rows, pve, float(np.sum(np.log(s_ + lbd)))


def lmm_reml_assoc_packed_f32_to_tsv(packed, n_samples, row_flip, row_maf, row_missing, s, xcov, y_rot, u_t,
                                     chrom, pos, snp, allele0, allele1, out_tsv, sample_indices=None,
                                     row_indices=None, low=-5.0, high=5.0, max_iter=50, tol=1e-2, threads=0,
                                     model="add", progress_callback=None, progress_every=0, nullml=None,
                                     init_log10_lbd=None, rotate_block_rows=256, bed_prefix=None, warm_start=None,
                                     warm_chain_pieces=1):
    """src/stats/lmm.rs:3364-3790 -> rows written (metadata lists empty => read the BIM via `bed_prefix`); the scan with the
    warm-start chain of `lmm_reml_assoc_packed_f32` (the reference passes `true, true` here as well, lmm.rs:3609-3610)."""
    from .tsv import write_assoc_tsv
    out = lmm_reml_assoc_packed_f32(packed, n_samples, row_flip, row_maf, s, xcov, y_rot, u_t, sample_indices,
                                    row_indices, low, high, max_iter, tol, threads, model, progress_callback,
                                    progress_every, nullml, init_log10_lbd, rotate_block_rows, warm_start, warm_chain_pieces)
    m = out.shape[0]
    if not len(chrom):
        if not bed_prefix:
            raise RuntimeError("metadata lists are empty and bed_prefix is not set")
        from .bed import read_bim
        bim = read_bim(bed_prefix)
        sel = np.arange(m) if row_indices is None else np.asarray(row_indices, dtype=np.int64)
        chrom = [bim.chrom[j] for j in sel]
        pos = [bim.pos[j] for j in sel]
        snp = [bim.snp[j] for j in sel]
        allele0 = [bim.a0[j] for j in sel]
        allele1 = [bim.a1[j] for j in sel]
    written = write_assoc_tsv(out_tsv, chrom, pos, snp, allele0, allele1, np.asarray(row_maf, dtype=np.float32),
                              np.asarray(row_missing, dtype=np.float32), out, resolve=False)
    return written


def fvlmm_assoc_packed_f32_to_tsv(packed, n_samples, row_flip, row_maf, row_missing, u, s, y, x, sample_indices,
                                  low, high, max_iter, tol, tau, threads, model, chrom, pos, snp, allele0, allele1,
                                  out_tsv, progress_callback=None, progress_every=0, fixed_lbd=None, fixed_ml0=None,
                                  row_indices=None, rotate_block_rows=0, bed_prefix=None):
    """src/stats/fvlmm.rs:4958-5190 -> (lbd, ml0, reml0); writes the TSV.

    `u` (n_samples, k) f32 holds eigenvectors as columns, `s` (k) f32 their eigenvalues (caller order, the packed
    workflow passes them descending); the function adds the intercept to `x`, rotates X and y itself with the
    sample-subset of U^T (f32 widened, f64 accumulation), fits lambda by Brent on REML unless `fixed_lbd` is given,
    and scans every row at that lambda.  `fixed_ml0` switches the plrt column on."""
    import math
    from .tsv import write_assoc_tsv
    if fixed_lbd is None and low >= high:
        raise RuntimeError("low must be < high")
    if not (np.isfinite(tol) and tol > 0):
        raise RuntimeError("tol must be positive and finite")
    if not (np.isfinite(tau) and tau >= 0):
        raise RuntimeError("tau must be finite and >= 0")
    if int(n_samples) <= 0:
        raise RuntimeError("n_samples must be > 0")
    from .stats import genetic_model_code
    genetic_model_code(model)                        # add / dom / rec / het (src/decode/decode.rs:100-147); anything else raises
    packed = _c(packed, np.uint8)
    if packed.ndim != 2:
        raise RuntimeError("packed must be 2D (m, bytes_per_snp)")
    if packed.shape[1] != (int(n_samples) + 3) // 4:
        raise RuntimeError(f"packed second dimension mismatch: got {packed.shape[1]}, expected {(int(n_samples) + 3) // 4}")
    m = int(packed.shape[0]) if row_indices is None else int(len(row_indices))
    for name, arr in (("row_flip", row_flip), ("row_maf", row_maf), ("row_missing", row_missing)):
        if len(arr) != m:
            raise RuntimeError(f"{name} length mismatch: got {len(arr)}, expected {m}")
    yv = _c(y, np.float64).ravel()
    n = int(yv.shape[0])
    if n == 0:
        raise RuntimeError("y must not be empty")
    if sample_indices is not None:
        sidx = np.asarray(sample_indices, dtype=np.int64).ravel()
        if sidx.shape[0] != n:
            raise RuntimeError(f"sample_indices length mismatch: got {sidx.shape[0]}, expected {n}")
        if sidx.size and (sidx.min() < 0 or sidx.max() >= int(n_samples)):
            raise RuntimeError("sample_indices out of range")
    else:
        if n != int(n_samples):
            raise RuntimeError(f"len(y)={n} must equal n_samples={int(n_samples)} when sample_indices is not provided")
        sidx = np.arange(n, dtype=np.int64)
    u = _c(u, np.float32)
    if u.ndim != 2:
        raise RuntimeError("u must be 2D (n_samples, k)")
    if u.shape[0] != int(n_samples):
        raise RuntimeError(f"u row count mismatch: got {u.shape[0]}, expected {int(n_samples)}")
    k_full = int(u.shape[1])
    s32 = _c(s, np.float32).ravel()
    if s32.shape[0] != k_full:
        raise RuntimeError(f"s length mismatch: got {s32.shape[0]}, expected {k_full}")
    if k_full < n:
        raise RuntimeError("u must provide full-rank eigenvectors for packed fixed-lambda scan: "
                           f"got k={k_full}, expected >= n={n}")
    s_vec = s32[:n].astype(np.float64) + (float(tau) if tau != 0.0 else 0.0)
    if not np.all(np.isfinite(s_vec)):
        raise RuntimeError("invalid s/tau produced non-finite values")
    if x is not None:
        xa = _c(x, np.float64)
        if xa.ndim != 2:
            raise RuntimeError("x must be 2D (n, p0)")
        if xa.shape[0] != n:
            raise RuntimeError(f"x row count mismatch: got {xa.shape[0]}, expected {n}")
        x_full = np.concatenate([np.ones((n, 1)), xa], axis=1)
    else:
        x_full = np.ones((n, 1), dtype=np.float64)
    p = int(x_full.shape[1])
    if n <= p:
        raise RuntimeError(f"n must be > p for null model: n={n}, p={p}")
    if n <= p + 1:
        raise RuntimeError(f"n must be > p+1 for SNP tests: n={n}, p={p}")
    # u_t_sub[c, j] = u[sample_idx[j], c], c < n (fvlmm.rs:1458-1482)
    u_t = np.ascontiguousarray(u[sidx, :n].T)
    x_rot, y_rot = lmm_rotate_x_y_with_ut_f64(u_t, x_full, yv)
    y_rot = y_rot.ravel()
    with_plrt = fixed_ml0 is not None
    if with_plrt and not np.isfinite(fixed_ml0):
        raise RuntimeError("fixed_ml0 must be finite when provided")
    if fixed_lbd is not None:
        if not (np.isfinite(fixed_lbd) and fixed_lbd > 0):
            raise RuntimeError("fixed_lbd must be finite and > 0 when provided")
        lbd = float(fixed_lbd)
        ml0 = float(fixed_ml0) if with_plrt else math.nan
        reml0 = _loglike_null(s_vec, x_rot, y_rot, math.log10(lbd))[1] if with_plrt else math.nan
    else:
        lbd, _, reml0 = lmm_reml_null_f32(s_vec, x_rot, y_rot, low, high, max_iter, tol)
        ml0 = float(fixed_ml0) if with_plrt else math.nan
    if not (np.isfinite(lbd) and lbd > 0):
        raise RuntimeError("invalid fixed lambda in packed fvlmm")
    out = _assoc_packed(packed, n_samples, row_flip, row_maf, s_vec, x_rot, y_rot, u_t,
                        None if sample_indices is None else sidx, row_indices, 1, math.log10(lbd),
                        math.log10(lbd) + 1.0, 0, 1e-2, 0, 0.0, float(fixed_ml0) if with_plrt else None,
                        progress_callback, progress_every, genetic_model=model)
    if not len(chrom):
        if not bed_prefix:
            raise RuntimeError("metadata lists are empty and bed_prefix is not set")
        from .bed import read_bim
        bim = read_bim(bed_prefix)
        sel = np.arange(m) if row_indices is None else np.asarray(row_indices, dtype=np.int64)
        chrom = [bim.chrom[j] for j in sel]
        pos = [bim.pos[j] for j in sel]
        snp = [bim.snp[j] for j in sel]
        allele0 = [bim.a0[j] for j in sel]
        allele1 = [bim.a1[j] for j in sel]
    write_assoc_tsv(out_tsv, chrom, pos, snp, allele0, allele1, np.asarray(row_maf, dtype=np.float32),
                    np.asarray(row_missing, dtype=np.float32), out, resolve=False)
    return float(lbd), float(ml0), float(reml0)


# ------------------------------------------------------------------------------------------------
# LMM -> LM fallback decision (src/stats/gwas_unified.rs:54-175); O(n p^2) host arithmetic, as in the reference
# ------------------------------------------------------------------------------------------------

def gwas_lmm_lm_null_lrt_decision(y, x_cov, lmm_ml0, alpha=0.05, boundary_mixture=True):
    """-> (switch_to_lm, lrt_stat, pval, lm_ml0). `x_cov` excludes the intercept (it is prepended here)."""
    import math
    if not math.isfinite(lmm_ml0):
        raise RuntimeError("lmm_ml0 must be finite")
    if not (math.isfinite(alpha) and 0.0 < alpha < 1.0):
        raise RuntimeError("alpha must be in (0,1)")
    y = _c(y, np.float64).ravel()
    x = _c(x_cov, np.float64)
    if x.ndim != 2:
        raise RuntimeError("x_cov must be 2D")
    n, p_cov = y.shape[0], x.shape[1]
    if x.shape[0] != n:
        raise RuntimeError("x_cov rows must equal len(y)")
    if n <= p_cov + 1:
        raise RuntimeError("insufficient samples: require n > p_cov + 1")
    xd = np.concatenate([np.ones((n, 1)), x], axis=1)
    xtx = xd.T @ xd
    xty = xd.T @ y
    try:
        chol = np.linalg.cholesky(xtx)
    except np.linalg.LinAlgError:
        try:
            chol = np.linalg.cholesky(xtx + 1e-8 * np.eye(p_cov + 1))
        except np.linalg.LinAlgError:
            raise RuntimeError("failed to compute LM null log-likelihood") from None
    beta = np.linalg.solve(chol.T, np.linalg.solve(chol, xty))
    r = y - xd @ beta
    rss = float(np.dot(r, r))
    if not (math.isfinite(rss) and rss > 0.0):
        raise RuntimeError("failed to compute LM null log-likelihood")
    n_f = float(n)
    lm_ml0 = n_f * (math.log(n_f) - 1.0 - math.log(2.0 * math.pi)) / 2.0 - 0.5 * n_f * math.log(rss)
    stat = 2.0 * (float(lmm_ml0) - lm_ml0)
    if not math.isfinite(stat) or stat < 0.0:
        stat = 0.0
    pval = 1.0 if stat <= 0.0 else math.erfc(math.sqrt(0.5 * stat))
    pval = min(max(pval, 2.2250738585072014e-308), 1.0) if math.isfinite(pval) else 1.0
    if boundary_mixture:
        pval *= 0.5
    if not math.isfinite(pval):
        pval = 1.0
    pval = min(max(pval, 2.2250738585072014e-308), 1.0)
    return bool(pval >= alpha), stat, pval, lm_ml0


def lm_precompute_ixx_qr(x):
    """`_lm_precompute_ixx_qr` (python/janusx/pyBLUP/assoc.py:453-480): (X'X)^-1 of the LM design through the reduced QR,
    the Hermitian pseudo-inverse when X is rank deficient."""
    x = _c(x, np.float64)
    if x.ndim != 2:
        raise ValueError("X must be 2D for LM QR precomputation.")
    n, q = x.shape
    if n == 0 or q == 0:
        raise ValueError("X must be non-empty for LM QR precomputation.")
    _q, r = np.linalg.qr(x, mode="reduced")
    diag = np.abs(np.diag(r))
    tol = np.finfo(np.float64).eps * float(max(n, q)) * (float(diag.max()) if diag.size else 0.0)
    if int(np.sum(diag > tol)) == q:
        rinv = np.linalg.inv(r)
        return np.ascontiguousarray(rinv @ rinv.T)
    return np.ascontiguousarray(np.linalg.pinv(x.T @ x, hermitian=True))


def lm_block_assoc_packed(y, x, ixx, packed, n_samples, row_flip, row_maf, sample_indices=None, chunk_size=10000,
                          threads=0, progress_callback=None, progress_every=0):
    """src/stats/glm.rs:3550-3860 -> f64 (m, 4) = beta, se, pwald (two-sided Student t, df = n - q0 - 1), plrt.  `x` (n, q0)
    carries the intercept column (the `LM` wrapper prepends it, python/janusx/pyBLUP/assoc.py:2192-2199); `chunk_size` and
    `threads` are accepted for signature parity (the whole payload is scanned from HBM)."""
    if int(n_samples) <= 0:
        raise RuntimeError("n_samples must be > 0")
    if int(chunk_size) <= 0:
        raise RuntimeError("chunk_size must be > 0")
    y = _c(y, np.float64).ravel()
    x = _c(x, np.float64)
    ixx = _c(ixx, np.float64)
    packed = _c(packed, np.uint8)
    if packed.ndim != 2:
        raise RuntimeError("packed must be 2D (m, bytes_per_snp)")
    m, bps = int(packed.shape[0]), int(packed.shape[1])
    if bps != (int(n_samples) + 3) // 4:
        raise RuntimeError(f"packed second dimension mismatch: got {bps}, expected {(int(n_samples) + 3) // 4}")
    flip = _c(np.asarray(row_flip).astype(np.uint8), np.uint8).ravel()
    maf = _c(row_maf, np.float32).ravel()
    if flip.shape[0] != m:
        raise RuntimeError(f"row_flip length mismatch: got {flip.shape[0]}, expected {m}")
    if maf.shape[0] != m:
        raise RuntimeError(f"row_maf length mismatch: got {maf.shape[0]}, expected {m}")
    idx, n_sel = _opt_idx(sample_indices)
    n = y.shape[0]
    if (n_sel if idx is not None else int(n_samples)) != n:
        raise RuntimeError(f"sample_indices length mismatch: got {n_sel if idx is not None else int(n_samples)}, "
                           f"expected len(y)={n}")
    if x.ndim != 2 or x.shape[0] != n:
        raise RuntimeError("X.n_rows must equal len(y)")
    q0 = int(x.shape[1])
    if ixx.shape != (q0, q0):
        raise RuntimeError("ixx must be (q0,q0)")
    if n <= q0 + 1:
        raise RuntimeError(f"n too small: require n > q0+1, got n={n}, q0={q0}")
    out = np.zeros((m, 4), dtype=np.float64)
    with _progress_hook(progress_callback, progress_every):
        check(lib().jx_lm_assoc_packed(_p(y), _p(x), _p(ixx), q0, _p(packed), m, int(n_samples), _p(flip), _p(maf),
                                       _p(idx), n_sel, _p(out)))
    return out


def _resolve_assoc_tsv_metadata(bed_prefix, chrom, pos, snp, allele0, allele1, row_indices, expected_len):
    """`resolve_assoc_tsv_metadata` (src/io/assoc2tsv.rs:139-195): all five lists empty -> the BIM columns of `bed_prefix`
    (rows `row_indices` when given); otherwise every list must have `expected_len` entries."""
    if not any(len(a) for a in (chrom, pos, snp, allele0, allele1)):
        prefix = (bed_prefix or "").strip()
        if not prefix:
            raise RuntimeError("empty TSV metadata requires non-empty bed_prefix")
        from .bed import read_bim
        bim = read_bim(_bed_prefix(prefix))
        sel = range(len(bim.chrom)) if row_indices is None else [int(j) for j in row_indices]
        chrom, pos, snp = [bim.chrom[j] for j in sel], [int(bim.pos[j]) for j in sel], [bim.snp[j] for j in sel]
        allele0, allele1 = [bim.a0[j] for j in sel], [bim.a1[j] for j in sel]
        if len(chrom) != expected_len:
            raise RuntimeError(f"BIM metadata length mismatch: expected={expected_len}, chrom={len(chrom)}, pos={len(pos)}, "
                               f"snp={len(snp)}, allele0={len(allele0)}, allele1={len(allele1)}")
        return chrom, pos, snp, allele0, allele1
    if not all(len(a) == expected_len for a in (chrom, pos, snp, allele0, allele1)):
        raise RuntimeError(f"TSV metadata length mismatch: rows={expected_len}, chrom={len(chrom)}, pos={len(pos)}, "
                           f"snp={len(snp)}, allele0={len(allele0)}, allele1={len(allele1)}")
    return chrom, pos, snp, allele0, allele1


def lm_block_assoc_f32(y, x, ixx, g, chunk_size=10000, threads=0):
    """src/stats/glm.rs:4313-4497: the LM formulas of `lm_block_assoc_packed` on an already decoded SNP-major f32 block
    `g` (m, n) (the `LM.gwas` wrapper of python/janusx/pyBLUP/assoc.py:613) -> f64 (m, 4) = beta, se, pwald, plrt."""
    if int(chunk_size) <= 0:
        raise RuntimeError("chunk_size must be > 0")
    y = _c(y, np.float64).ravel()
    x = _c(x, np.float64)
    ixx = _c(ixx, np.float64)
    g = _c(g, np.float32)
    n = int(y.shape[0])
    if x.ndim != 2 or x.shape[0] != n:
        raise RuntimeError("X.n_rows must equal len(y)")
    q0 = int(x.shape[1])
    if ixx.shape != (q0, q0):
        raise RuntimeError("ixx must be (q0,q0)")
    if g.ndim != 2 or g.shape[1] != n:
        raise RuntimeError("g must be shape (m, n)")
    if n <= q0 + 1:
        raise RuntimeError(f"n too small: require n > q0+1, got n={n}, q0={q0}")
    m = int(g.shape[0])
    out = np.zeros((m, 4), dtype=np.float64)
    check(lib().jx_lm_assoc_dense(_p(y), _p(x), _p(ixx), q0, _p(g), m, n, _p(out)))
    return out


def lm_block_assoc_packed_to_tsv(y, x, ixx, packed, n_samples, row_flip, row_maf, row_missing, chrom, pos, snp, allele0,
                                 allele1, out_tsv, maf_threshold=0.0, max_missing_rate=1.0, het_threshold=0.0,
                                 sample_indices=None, row_indices=None, chunk_size=10000, threads=0,
                                 progress_callback=None, progress_every=0, bed_prefix=None):
    """src/stats/glm.rs:3862-4305: `lm_block_assoc_packed` with the result table written (11 columns, `miss` as the count
    `(row_missing * n_samples) as i64` in f32 arithmetic, pwald sanitised; the thresholds are only range-checked, the rows
    arrive filtered) -> (rows written, rows scanned).  Empty metadata lists are read from `<bed_prefix>.bim`."""
    from .tsv import write_assoc_tsv_counts
    if int(n_samples) <= 0:
        raise RuntimeError("n_samples must be > 0")
    if int(chunk_size) <= 0:
        raise RuntimeError("chunk_size must be > 0")
    if not (0.0 <= maf_threshold <= 0.5):
        raise ValueError("maf_threshold must be within [0, 0.5]")
    if not (0.0 <= max_missing_rate <= 1.0):
        raise ValueError("max_missing_rate must be within [0, 1.0]")
    if not (0.0 <= het_threshold <= 1.0):
        raise ValueError("het_threshold must be within [0, 1.0]")
    packed = _c(packed, np.uint8)
    if packed.ndim != 2:
        raise RuntimeError("packed must be 2D (m, bytes_per_snp)")
    if row_indices is not None:
        ri = _c(row_indices, np.int64).ravel()
        if ri.size and (ri.min() < 0 or ri.max() >= packed.shape[0]):
            raise RuntimeError("row_indices out of range")
        packed = np.ascontiguousarray(packed[ri])
    else:
        ri = None
    m = int(packed.shape[0])
    miss = _c(row_missing, np.float32).ravel()
    if np.asarray(row_flip).ravel().shape[0] != m:
        raise RuntimeError("row_flip length mismatch")
    if _c(row_maf, np.float32).ravel().shape[0] != m:
        raise RuntimeError("row_maf length mismatch")
    if miss.shape[0] != m:
        raise RuntimeError("row_missing length mismatch")
    chrom, pos, snp, allele0, allele1 = _resolve_assoc_tsv_metadata(bed_prefix, chrom, pos, snp, allele0, allele1, ri, m)
    stats = lm_block_assoc_packed(y, x, ixx, packed, n_samples, row_flip, row_maf, sample_indices, chunk_size, threads,
                                  progress_callback, progress_every)
    counts = (miss * np.float32(int(n_samples))).astype(np.int64).astype(np.float32)
    rows = write_assoc_tsv_counts(out_tsv, chrom, pos, snp, allele0, allele1, _c(row_maf, np.float32).ravel(), counts,
                                  stats[:, :3], resolve=False)
    if progress_callback is not None:
        progress_callback(m, m)
    return rows, m


# ------------------------------------------------------------------------------------------------
# Genotype rows as numbers and the two matrix-free products the reference's Python layer asks for beside the path
# (src/stats/packed.rs:577-760, 2060-2560): decode of selected rows, M'alpha, cross-GRM times alpha
# ------------------------------------------------------------------------------------------------

def _raw_additive_lut(maf_rows, flip_rows, clamp_low_only):
    """[0, mean, 1, 2] or flipped per row; mean = max(2 maf, 0) (`bed_packed_decode_rows_f32`, packed.rs:652) or 2 maf as it is
    (`packed_malpha_f64` / `cross_grm_times_alpha_packed_f64`, packed.rs:2470, 2240)."""
    maf = np.asarray(maf_rows, dtype=np.float32)
    mean_g = (np.float32(2.0) * maf).astype(np.float32)
    if clamp_low_only:
        mean_g = np.maximum(mean_g, np.float32(0.0))
    flip = np.asarray(flip_rows).astype(bool)
    lut = np.empty((len(maf), 4), dtype=np.float32)
    lut[:, 0] = np.where(flip, 2.0, 0.0)
    lut[:, 1] = mean_g
    lut[:, 2] = 1.0
    lut[:, 3] = np.where(flip, 0.0, 2.0)
    return lut


def _packed_host_checks(packed, n_samples):
    packed = _c(packed, np.uint8)
    if packed.ndim != 2:
        raise RuntimeError("packed must be 2D (m, bytes_per_snp)")
    if int(n_samples) <= 0:
        raise RuntimeError("n_samples must be > 0")
    bps = (int(n_samples) + 3) // 4
    if packed.shape[1] != bps:
        raise RuntimeError(f"packed second dimension mismatch: got {packed.shape[1]}, expected {bps} for "
                           f"n_samples={int(n_samples)}")
    return packed


def bed_packed_decode_rows_f32(packed, n_samples, row_indices, row_flip, row_maf, sample_indices=None):
    """src/stats/packed.rs:577-672: rows `row_indices` of the 2-bit payload as f32 values over `sample_indices` (0 / 1 / 2,
    missing = max(2 maf, 0), flipped rows 2 - g) -> f32 (len(row_indices), n_out).  `row_flip` / `row_maf` are indexed by
    payload row when they have one entry per payload row, else by position in `row_indices`."""
    import torch
    from . import pipeline as pl
    packed = _packed_host_checks(packed, n_samples)
    m = int(packed.shape[0])
    rows = _c(row_indices, np.int64).ravel()
    if rows.size and (rows.min() < 0 or rows.max() >= m):
        raise RuntimeError(f"row_indices out of range for {m} rows")
    flip = np.asarray(row_flip).astype(bool).ravel()
    maf = _c(row_maf, np.float32).ravel()
    packed_space = flip.shape[0] == m and maf.shape[0] == m
    if not (packed_space or (flip.shape[0] == rows.shape[0] and maf.shape[0] == rows.shape[0])):
        raise RuntimeError(f"row_flip/row_maf length mismatch: got row_flip={flip.shape[0]}, row_maf={maf.shape[0]}, expected "
                           f"either packed_rows={m} or selected_rows={rows.shape[0]}")
    idx, n_sel = _opt_idx(sample_indices)
    if idx is not None and n_sel and (idx.min() < 0 or idx.max() >= int(n_samples)):
        raise RuntimeError("sample_indices out of range")
    n_out = n_sel if idx is not None else int(n_samples)
    if n_out == 0 or rows.size == 0:
        return np.zeros((rows.shape[0], n_out), dtype=np.float32)
    lut = _raw_additive_lut(maf[rows] if packed_space else maf, flip[rows] if packed_space else flip, True)
    dev = torch.device("cuda", torch.cuda.current_device())
    payload = torch.from_numpy(np.ascontiguousarray(packed[rows])).to(dev)
    panel = pl.Panel(payload, int(n_samples), idx)
    out = torch.empty((rows.shape[0], n_out), dtype=torch.float32, device=dev)
    lut_t = torch.from_numpy(lut).to(dev)
    check(lib().jxg_decode_rows_p32(panel.p32.data_ptr(), panel.m, n_out, None, int(rows.shape[0]), lut_t.data_ptr(),
                                    out.data_ptr(), n_out, pl._stream()))
    return out.cpu().numpy()


def bed_decode_rows_f32_from_meta(prefix, row_indices, row_flip, row_maf, sample_indices=None, mmap_window_mb=None):
    """src/stats/packed.rs:674-760: `bed_packed_decode_rows_f32` on the rows of a BED prefix (metadata per selected row)."""
    from .bed import read_bed_payload
    packed, n_samples, _bim = read_bed_payload(_bed_prefix(prefix))
    rows = _c(row_indices, np.int64).ravel()
    flip = np.asarray(row_flip).astype(bool).ravel()
    maf = _c(row_maf, np.float32).ravel()
    if flip.shape[0] != rows.shape[0] or maf.shape[0] != rows.shape[0]:
        raise RuntimeError(f"row meta length mismatch: row_indices={rows.shape[0]}, row_flip={flip.shape[0]}, "
                           f"row_maf={maf.shape[0]}")
    if rows.size and (rows.min() < 0 or rows.max() >= packed.shape[0]):
        raise RuntimeError("row_indices out of range")
    sub = np.ascontiguousarray(packed[rows])
    return bed_packed_decode_rows_f32(sub, n_samples, np.arange(rows.shape[0], dtype=np.int64), flip, maf, sample_indices)


def _malpha_inputs(packed, n_samples, row_flip, row_maf, sample_indices):
    import torch
    from . import pipeline as pl
    packed = _packed_host_checks(packed, n_samples)
    m = int(packed.shape[0])
    if m == 0:
        raise RuntimeError("packed must contain at least one SNP row")
    flip = np.asarray(row_flip).astype(bool).ravel()
    maf = _c(row_maf, np.float32).ravel()
    if flip.shape[0] != m:
        raise RuntimeError(f"row_flip length mismatch: got {flip.shape[0]}, expected {m}")
    if maf.shape[0] != m:
        raise RuntimeError(f"row_maf length mismatch: got {maf.shape[0]}, expected {m}")
    idx = _c(sample_indices, np.int64).ravel()
    if idx.size == 0:
        raise RuntimeError("sample_indices must not be empty")
    if idx.min() < 0 or idx.max() >= int(n_samples):
        raise RuntimeError("sample_indices out of range")
    dev = torch.device("cuda", torch.cuda.current_device())
    identity = idx.shape[0] == int(n_samples) and np.array_equal(idx, np.arange(int(n_samples)))
    panel = pl.Panel(torch.from_numpy(packed).to(dev), int(n_samples), None if identity else idx)
    lut_t = torch.from_numpy(_raw_additive_lut(maf, flip, False)).to(dev)
    return panel, lut_t, m, int(idx.shape[0]), dev


def packed_malpha_f64(packed, n_samples, row_flip, row_maf, sample_indices, alpha, block_rows=4096, threads=0):
    """src/stats/packed.rs:2352-2575: m_alpha[r] = sum_j g[r, s_j] alpha[j] over the mean-imputed raw genotypes
    ([0, 2 maf, 1, 2] or flipped) of the selected samples -> f64 (m).  (`jxg_packed_tdot`: f64 sums on the device; the
    reference rounds alpha to f32 and sums in an f32 sgemm.)"""
    import torch
    from . import pipeline as pl
    panel, lut_t, m, n_out, dev = _malpha_inputs(packed, n_samples, row_flip, row_maf, sample_indices)
    a = _c(alpha, np.float64).ravel()
    if a.shape[0] != n_out:
        raise RuntimeError(f"alpha length mismatch: got {a.shape[0]}, expected {n_out} (len(sample_indices))")
    out = torch.empty(m, dtype=torch.float64, device=dev)
    a_t = torch.from_numpy(a).to(dev)
    check(lib().jxg_packed_tdot(panel.p32.data_ptr(), panel.m, n_out, None, m, lut_t.data_ptr(), a_t.data_ptr(),
                                out.data_ptr(), pl._stream()))
    return out.cpu().numpy()


def cross_grm_times_alpha_packed_f64(packed, n_samples, row_flip, row_maf, sample_indices, m_alpha, m_mean, alpha_sum,
                                     mean_sq, mean_malpha, m_var_sum, block_rows=4096, threads=0):
    """src/stats/packed.rs:2060-2350: K_cross alpha for the samples `sample_indices` without the cross GRM:
    ((M_s' m_alpha) - (M_s' m_mean) alpha_sum + mean_sq alpha_sum - mean_malpha) / m_var_sum -> f64 (n_out, 1)
    (`jxg_packed_dot`, two passes over the payload)."""
    import math
    import torch
    from . import pipeline as pl
    if int(n_samples) <= 0:
        raise RuntimeError("n_samples must be > 0")
    if not (math.isfinite(m_var_sum) and m_var_sum > 0.0):
        raise RuntimeError("m_var_sum must be finite and > 0 for compact cross-GRM prediction")
    if not (math.isfinite(alpha_sum) and math.isfinite(mean_sq) and math.isfinite(mean_malpha)):
        raise RuntimeError("alpha_sum/mean_sq/mean_malpha must be finite")
    panel, lut_t, m, n_out, dev = _malpha_inputs(packed, n_samples, row_flip, row_maf, sample_indices)
    ma = _c(m_alpha, np.float64).ravel()
    mm = _c(m_mean, np.float64).ravel()
    if ma.shape[0] != m:
        raise RuntimeError(f"m_alpha length mismatch: got {ma.shape[0]}, expected {m}")
    if mm.shape[0] != m:
        raise RuntimeError(f"m_mean length mismatch: got {mm.shape[0]}, expected {m}")
    t1 = torch.empty(n_out, dtype=torch.float64, device=dev)
    t2 = torch.empty(n_out, dtype=torch.float64, device=dev)
    for vec, dst in ((ma, t1), (mm, t2)):
        v_t = torch.from_numpy(vec).to(dev)
        check(lib().jxg_packed_dot(panel.p32.data_ptr(), panel.m, n_out, None, m, lut_t.data_ptr(), v_t.data_ptr(),
                                   dst.data_ptr(), pl._stream()))
    const_term = float(mean_sq) * float(alpha_sum) - float(mean_malpha)
    out = (t1.cpu().numpy() - t2.cpu().numpy() * float(alpha_sum) + const_term) * (1.0 / float(m_var_sum))
    return out.reshape(-1, 1)


# ------------------------------------------------------------------------------------------------
# GBLUP (`jx gs -BLUP`, n <= 15 000 branch): src/stats/gblup.rs:1242-1516 `gblup_reml_npy_grm`
# ------------------------------------------------------------------------------------------------

def gblup_reml_grm(grm, train_sample_indices, y_train, test_sample_indices=None, train_pred_local_indices=None,
                   g_eps=1e-8, low=-6.0, high=6.0, max_iter=50, tol=1e-4, threads=0,
                   return_variance_components=False, estimate_only=False):
    """In-memory form of `gblup_reml_npy_grm`: `grm` (n_full, n_full) f32/f64.  Returns the reference's 12-tuple
    (pred_train (k,1), pred_test (t,1), pve, lambda, ml, reml, evd_backend, evd_elapsed, 0, sigma_g2, sigma_e2,
    effect (empty))."""
    import math
    if not (math.isfinite(g_eps) and g_eps >= 0.0):
        raise RuntimeError("g_eps must be finite and >= 0")
    if not (math.isfinite(low) and math.isfinite(high) and low < high):
        raise RuntimeError("low/high must be finite and low < high")
    if int(max_iter) == 0:
        raise RuntimeError("max_iter must be > 0")
    if not (math.isfinite(tol) and tol > 0.0):
        raise RuntimeError("tol must be finite and > 0")
    k = np.asarray(grm)
    if k.ndim != 2 or k.shape[0] != k.shape[1]:
        raise RuntimeError("GRM must be a square matrix")
    is64 = k.dtype == np.float64
    k = _c(k, np.float64 if is64 else np.float32)
    tr = _c(train_sample_indices, np.int64).ravel()
    y = _c(y_train, np.float64).ravel()
    if y.shape[0] != tr.shape[0]:
        raise RuntimeError("y_train length must equal train_sample_indices length")
    te = np.zeros(0, dtype=np.int64) if test_sample_indices is None else _c(test_sample_indices, np.int64).ravel()
    pt = np.zeros(tr.shape[0], dtype=np.float64)
    pe = np.zeros(max(te.shape[0], 1), dtype=np.float64)
    sc = np.zeros(8, dtype=np.float64)
    t0 = time.perf_counter()
    check(lib().jx_gblup_reml_grm(_p(k), 1 if is64 else 0, int(k.shape[0]), _p(tr), int(tr.shape[0]), _p(y), _p(te),
                                  int(te.shape[0]), float(g_eps), float(low), float(high), int(max_iter), float(tol),
                                  1 if estimate_only else 0, _p(pt), _p(pe), _p(sc)))
    el = time.perf_counter() - t0
    if estimate_only:
        ptr = np.zeros((0, 1))
        pte = np.zeros((0, 1))
    else:
        if train_pred_local_indices is not None:
            pt = pt[np.asarray(train_pred_local_indices, dtype=np.int64)]
        ptr = pt.reshape(-1, 1)
        pte = pe[: te.shape[0]].reshape(-1, 1)
    sg2 = float(sc[4]) if return_variance_components else float("nan")
    se2 = float(sc[5]) if return_variance_components else float("nan")
    return (ptr, pte, float(sc[0]), float(sc[1]), float(sc[2]), float(sc[3]), "rocsolver", el, 0, sg2, se2,
            np.zeros(0, dtype=np.float64))


def _gblup_meta_lut(maf, flip, identity):
    """Centred additive decode of `decode_meta_block_f32` (src/stats/gblup.rs:239-404; bedmath.rs:1359-1441 for a sample
    subset) -> (LUT (m, 4) f32 of centred values, missing = 0; sum of the row variances; row means f64)."""
    from . import stats as st
    m = int(maf.shape[0])
    mafc = np.clip(maf, np.float32(0.0), np.float32(1.0))
    if identity:
        p64 = mafc.astype(np.float64)
        mean64 = 2.0 * p64
        var = 2.0 * p64 * (1.0 - p64)
        mean32 = mean64.astype(np.float32)
        row_mean = mean64
    else:
        mean32 = (np.float32(2.0) * mafc).astype(np.float32)
        pg = np.clip(np.float32(0.5) * mean32, np.float32(0.0), np.float32(1.0))
        var = np.maximum(np.float32(2.0) * pg * (np.float32(1.0) - pg), np.float32(0.0)).astype(np.float64)
        row_mean = mean32.astype(np.float64)
    return st.grm_lut_from_mean_scale(mean32, np.ones(m, dtype=np.float32), flip), float(np.sum(var)), row_mean


def gblup_effect_from_meta_stream(prefix, sample_indices, alpha, row_source_indices, row_flip, row_maf, mode="a",
                                  block_rows=4096, threads=0, mmap_window_mb=None):
    """src/stats/gblup.rs:2788-2895 -> `compute_effect_beta_from_meta_stream` (:930-1033): marker effects
    beta[r] = sum_j z[r, s_j] alpha[j] / sum(var) over the centred decode of the caller-prepared BED rows (the marker-effect
    output of `jx gs -BLUP`, python/janusx/gs/workflow.py:5103) -> f64 (m).  `jxg_packed_tdot` on the centred LUT."""
    import torch
    from . import pipeline as pl
    from .bed import stage_bed_payload
    mode_n = str(mode).strip().lower()
    if mode_n not in ("a", "add", "additive", "d", "dom", "dominance"):
        raise RuntimeError("mode must be one of {'a','additive','d','dominance'}")
    if mode_n in ("d", "dom", "dominance"):
        raise RuntimeError("mode 'dominance' is outside this build's scope (additive model only)")
    packed, n_fam, _bim = stage_bed_payload(_bed_prefix(prefix), mmap_window_mb)
    if n_fam == 0:
        raise RuntimeError("No samples found in BED input.")
    tr = _c(sample_indices, np.int64).ravel()
    if tr.size and (tr.min() < 0 or tr.max() >= n_fam):
        raise RuntimeError("sample_indices out of range")
    a = _c(alpha, np.float64).ravel()
    if a.shape[0] != tr.shape[0]:
        raise RuntimeError(f"alpha length mismatch: got {a.shape[0]}, expected {tr.shape[0]}")
    src = _c(row_source_indices, np.int64).ravel()
    if src.size == 0:
        raise RuntimeError("row_source_indices must not be empty.")
    if (src < 0).any():
        raise RuntimeError("row_source_indices must be non-negative.")
    flip = np.asarray(row_flip).astype(bool).ravel()
    maf = _c(row_maf, np.float32).ravel()
    if flip.shape[0] != src.shape[0] or maf.shape[0] != src.shape[0]:
        raise RuntimeError(f"metadata length mismatch: row_source_indices={src.shape[0]}, row_flip={flip.shape[0]}, "
                           f"row_maf={maf.shape[0]}")
    if tr.size == 0:
        raise RuntimeError("compute_effect_beta_from_meta_stream: sample_idx must not be empty")
    if int(src.max()) >= int(packed.shape[0]):
        raise RuntimeError("row_source_indices out of range")
    identity = tr.shape[0] == n_fam and np.array_equal(tr, np.arange(n_fam))
    glut, var_sum, _row_mean = _gblup_meta_lut(maf, flip, identity)
    if not (np.isfinite(var_sum) and var_sum > 0.0):
        raise RuntimeError("compute_effect_beta_from_meta_stream: invalid denominator")
    dev = packed.device
    rows_payload = packed[torch.from_numpy(src).to(dev)]
    del packed
    panel = pl.Panel(rows_payload, n_fam, None if identity else tr)
    m = int(src.shape[0])
    out = torch.empty(m, dtype=torch.float64, device=dev)
    lut_t = torch.from_numpy(np.ascontiguousarray(glut, dtype=np.float32)).to(dev)     # named: alive until the launch is queued
    a_t = torch.from_numpy(a).to(dev)
    check(lib().jxg_packed_tdot(panel.p32.data_ptr(), panel.m, int(tr.shape[0]), None, m, lut_t.data_ptr(), a_t.data_ptr(),
                                out.data_ptr(), pl._stream()))
    return out.cpu().numpy() * (1.0 / var_sum)


def _gblup_meta_grm(rows_payload, n_samples, tr, flip, maf):
    """Centred-additive GRM of the samples `tr` over the payload rows with the caller's flip / maf metadata, the
    formulation of `build_grm_from_meta_stream` (src/stats/gblup.rs:406-652; per-row centring and variance of
    `decode_meta_block_f32` :239-404, bedmath.rs:1359-1441 for a sample subset) -> (K f64 (n_tr, n_tr) on the device,
    scaled by 1 / sum(var); sum(var); the panel; the row means in f64)."""
    import torch
    from . import pipeline as pl
    from . import stats as st
    n_tr = int(tr.shape[0])
    m = int(rows_payload.shape[0])
    identity = n_tr == n_samples and np.array_equal(tr, np.arange(n_samples))
    glut, var_sum, row_mean = _gblup_meta_lut(maf, flip, identity)
    panel = pl.Panel(rows_payload, n_samples, None if identity else tr)
    acc = pl.grm_accumulate(panel, np.arange(m, dtype=np.int64), glut)
    k = pl.grm_finalize(acc, n_tr, var_sum, torch.float64)
    del acc
    return k, var_sum, panel, row_mean


def gblup_reml_packed_bed(prefix, train_sample_indices, y_train, test_sample_indices=None,
                          train_pred_local_indices=None, site_keep=None, g_eps=1e-8, low=-6.0, high=6.0, max_iter=50,
                          tol=1e-4, block_rows=4096, threads=0, return_variance_components=False, estimate_only=False,
                          return_effect=False, row_source_indices=None, row_flip=None, row_maf=None,
                          mmap_window_mb=None):
    """src/stats/gblup.rs:1517-1958, metadata-streaming path (the one `python/janusx/gs/workflow.py:6300-6360` takes; without
    the metadata arguments the `site_keep` route :1986-2140 with the loader's row statistics, same formulation):
    GRM of the training samples straight from the BED payload, spectral REML, then marker effects
    effect_beta = (M' alpha - mean * sum(alpha)) / sum(var) and predictions alpha0 + M beta, M never materialised
    (jxg_packed_tdot / jxg_packed_dot).  Returns the reference's 12-tuple (pred_train (k,1), pred_test (t,1), pve,
    lambda, ml, reml, evd_backend, evd_elapsed, eff_m, sigma_g2, sigma_e2, effect (m) or empty)."""
    import math
    import torch
    from . import pipeline as pl
    from . import stats as st
    from .bed import read_bed_payload
    if not (math.isfinite(g_eps) and g_eps >= 0.0):
        raise RuntimeError("g_eps must be finite and >= 0")
    if not (math.isfinite(low) and math.isfinite(high) and low < high):
        raise RuntimeError("low/high must be finite and low < high")
    if int(max_iter) == 0:
        raise RuntimeError("max_iter must be > 0")
    if not (math.isfinite(tol) and tol > 0.0):
        raise RuntimeError("tol must be finite and > 0")
    packed_loaded = None
    if row_source_indices is None or row_flip is None or row_maf is None:
        # `site_keep` route (gblup.rs:1986-2140): the whole payload (or the rows of the mask) with the loader's own row
        # statistics -- alt-allele frequency over all samples (`load_bed_2bit_packed`) and the flip mask of
        # `bed_packed_row_flip_mask` -- then the same model.  The reference evaluates it in marker space when
        # n_train > m (`gblup_marker_fast_packed`) and through `grm_packed_f64_with_stats` otherwise; here both go through
        # the formulation of the metadata route below (one GRM of the training samples, spectral REML, M' alpha).
        packed_loaded, _miss, maf_all, _std, n_loaded = load_bed_2bit_packed(prefix)
        if n_loaded == 0:
            raise RuntimeError("No samples found in BED input.")
        m_total = int(packed_loaded.shape[0])
        if m_total == 0:
            raise RuntimeError("No SNP rows found in BED input.")
        flip_all = bed_packed_row_flip_mask(packed_loaded, n_loaded)
        if site_keep is not None:
            mask = np.asarray(site_keep).astype(bool).ravel()
            if mask.shape[0] != m_total:
                raise RuntimeError(f"site_keep length mismatch: got {mask.shape[0]}, expected {m_total}")
            row_source_indices = np.nonzero(mask)[0].astype(np.int64)
            if row_source_indices.size == 0:
                raise RuntimeError("No SNPs remained after applying site_keep mask.")
        else:
            row_source_indices = np.arange(m_total, dtype=np.int64)
        row_flip = np.asarray(flip_all)[row_source_indices]
        row_maf = np.asarray(maf_all, dtype=np.float32)[row_source_indices]
    src = np.asarray(row_source_indices, dtype=np.int64).ravel()
    if src.size == 0:
        raise RuntimeError("row_source_indices must not be empty for metadata streaming path.")
    if (src < 0).any():
        raise RuntimeError("row_source_indices must be non-negative.")
    flip = np.asarray(row_flip).astype(bool).ravel()
    maf = _c(row_maf, np.float32).ravel()
    if flip.shape[0] != src.shape[0] or maf.shape[0] != src.shape[0]:
        raise RuntimeError(f"metadata length mismatch: row_source_indices={src.shape[0]}, row_flip={flip.shape[0]}, "
                           f"row_maf={maf.shape[0]}")
    if packed_loaded is not None:
        packed, n_samples = packed_loaded, int(n_loaded)
    else:
        packed, n_samples, _bim = read_bed_payload(prefix)
    if n_samples == 0:
        raise RuntimeError("No samples found in BED input.")
    if src.max() >= packed.shape[0]:
        raise RuntimeError("row_source_indices out of range")
    tr = _c(train_sample_indices, np.int64).ravel()
    if tr.size == 0:
        raise RuntimeError("train_sample_indices must not be empty.")
    if tr.min() < 0 or tr.max() >= n_samples:
        raise RuntimeError("train_sample_indices out of range")
    y = _c(y_train, np.float64).ravel()
    if y.shape[0] != tr.shape[0]:
        raise RuntimeError(f"y_train length mismatch: got {y.shape[0]}, expected {tr.shape[0]}")
    if not np.all(np.isfinite(y)):
        raise RuntimeError("y_train contains non-finite values.")
    te = np.zeros(0, dtype=np.int64) if test_sample_indices is None else _c(test_sample_indices, np.int64).ravel()
    if te.size and (te.min() < 0 or te.max() >= n_samples):
        raise RuntimeError("test_sample_indices out of range")
    n_tr = int(tr.shape[0])
    if n_tr <= 1:
        raise RuntimeError("GBLUP REML requires at least 2 training samples.")
    pick = tr if train_pred_local_indices is None else tr[np.asarray(train_pred_local_indices, dtype=np.int64)]
    m = int(src.shape[0])
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    rows_payload = torch.from_numpy(np.ascontiguousarray(packed[src])).to(dev)
    k, var_sum, panel, row_mean = _gblup_meta_grm(rows_payload, n_samples, tr, flip, maf)
    k.diagonal().add_(float(g_eps))
    y_mean = float(np.sum(y) / n_tr)
    yc = torch.from_numpy(y - y_mean).to(dev)
    alpha = torch.empty(n_tr, dtype=torch.float64, device=dev)
    fit = np.zeros(6, dtype=np.float64)
    t0 = time.perf_counter()
    check(lib().jxg_gblup_fit(k.data_ptr(), n_tr, 0.0, yc.data_ptr(), float(low), float(high), float(tol),
                              int(max_iter), alpha.data_ptr(), fit.ctypes.data, stream))
    evd_elapsed = time.perf_counter() - t0
    lbd, beta_rot, q, ml, reml, mean_s = [float(v) for v in fit]
    n_eff = float(n_tr - 1)
    sg2 = q / max(n_eff, 1.0)
    se2 = lbd * sg2
    var_g = sg2 * max(mean_s, 0.0)
    den = var_g + se2
    pve = var_g / den if (math.isfinite(den) and den > 0.0) else float("nan")
    beta0 = y_mean + beta_rot
    pred_tr = np.zeros((0, 1))
    pred_te = np.zeros((0, 1))
    effect = np.zeros(0, dtype=np.float64)
    if (not estimate_only) or return_effect:
        mg = np.clip(np.float32(2.0) * maf, np.float32(0.0), np.float32(2.0)).astype(np.float32)
        rlut = np.empty((m, 4), dtype=np.float32)      # mean-imputed raw genotype (bedmath.rs:983-988)
        rlut[:, 0] = np.where(flip, 2.0, 0.0)
        rlut[:, 1] = mg
        rlut[:, 2] = 1.0
        rlut[:, 3] = np.where(flip, 0.0, 2.0)
        rlut_t = torch.from_numpy(rlut).to(dev)
        m_alpha_t = torch.empty(m, dtype=torch.float64, device=dev)
        check(lib().jxg_packed_tdot(panel.p32.data_ptr(), panel.m, n_tr, None, m, rlut_t.data_ptr(), alpha.data_ptr(),
                                    m_alpha_t.data_ptr(), stream))
        m_alpha = m_alpha_t.cpu().numpy()
        alpha_sum = float(alpha.sum().item())
        mean_sq = float(np.sum(row_mean * row_mean))
        mean_malpha = float(np.sum(row_mean * m_alpha))
        inv_var = 1.0 / max(var_sum, 1e-12)
        effect = (m_alpha - row_mean * alpha_sum) * inv_var
        alpha0 = beta0 + (mean_sq * alpha_sum - mean_malpha) * inv_var
        if not estimate_only:
            beta_t = torch.from_numpy(effect).to(dev)

            def predict(idx):
                if len(idx) == 0:
                    return np.zeros((0, 1))
                pan = pl.Panel(rows_payload, n_samples, np.asarray(idx, dtype=np.int64))
                out = torch.empty(len(idx), dtype=torch.float64, device=dev)
                check(lib().jxg_packed_dot(pan.p32.data_ptr(), pan.m, len(idx), None, m, rlut_t.data_ptr(),
                                           beta_t.data_ptr(), out.data_ptr(), stream))
                return (out.cpu().numpy() + alpha0).reshape(-1, 1)

            pred_tr = predict(pick)
            pred_te = predict(te)
    sg2_o = sg2 if return_variance_components else float("nan")
    se2_o = se2 if return_variance_components else float("nan")
    return (pred_tr, pred_te, pve, lbd, ml, reml, "rocsolver", evd_elapsed, m, sg2_o, se2_o,
            effect if return_effect else np.zeros(0, dtype=np.float64))


def gblup_reml_npy_grm(grm_path, train_sample_indices, y_train, test_sample_indices=None,
                       train_pred_local_indices=None, g_eps=1e-8, low=-6.0, high=6.0, max_iter=50, tol=1e-4,
                       threads=0, return_variance_components=False, estimate_only=False):
    """src/stats/gblup.rs:1242-1516: `.npy` GRM (f32/f64 C-order) on disk."""
    k = np.load(grm_path)
    return gblup_reml_grm(k, train_sample_indices, y_train, test_sample_indices, train_pred_local_indices, g_eps, low,
                          high, max_iter, tol, threads, return_variance_components, estimate_only)


# ---- rrBLUP by PCG over the packed payload ---------------------------------------------------------

def load_bed_2bit_packed(prefix):
    """src/io/gfreader.rs:4401-4530: (packed (m, ceil(n/4)) u8, missing_rate f32, maf f32, std_denom f32, n_samples);
    the per-SNP counts come from the device, the f32 expressions are the reference's."""
    from .bed import read_bed_payload
    p = str(prefix)
    if p.lower().endswith((".bed", ".bim", ".fam")):
        p = p[:-4]
    packed, n_samples, _bim = read_bed_payload(p)
    if n_samples == 0:
        raise RuntimeError("no samples found in PLINK input")
    cnt = bed_row_counts(packed, n_samples)
    mi, he, ho = cnt[:, 0].astype(np.int64), cnt[:, 1].astype(np.int64), cnt[:, 2].astype(np.int64)
    f32 = np.float32
    miss = (mi.astype(f32) / f32(n_samples)).astype(f32)
    nm = n_samples - mi
    alt = he + 2 * ho
    ok = nm > 0
    pfreq = np.zeros(mi.shape[0], dtype=f32)
    pfreq[ok] = alt[ok].astype(f32) / (f32(2.0) * nm[ok].astype(f32))
    maf = np.where(ok, np.minimum(pfreq, f32(1.0) - pfreq), f32(0.0)).astype(f32)
    d = np.sqrt((f32(2.0) * pfreq * (f32(1.0) - pfreq)).astype(f32)).astype(f32)
    std = np.where(ok & np.isfinite(d), d, f32(0.0)).astype(f32)
    return packed, miss, maf, std, int(n_samples)


def bed_packed_row_flip_mask(packed, n_samples):
    """src/stats/packed.rs:45-121: True where the ALT frequency among non-missing calls exceeds 0.5 (f64)."""
    packed = _c(packed, np.uint8)
    if packed.ndim != 2:
        raise RuntimeError("packed must be 2D (m, bytes_per_snp)")
    if int(n_samples) <= 0:
        raise RuntimeError("n_samples must be > 0")
    if packed.shape[1] != (int(n_samples) + 3) // 4:
        raise RuntimeError(f"packed second dimension mismatch: got {packed.shape[1]}, expected "
                           f"{(int(n_samples) + 3) // 4} for n_samples={int(n_samples)}")
    cnt = bed_row_counts(packed, n_samples).astype(np.int64)
    nm = int(n_samples) - cnt[:, 0]
    alt = cnt[:, 1] + 2 * cnt[:, 2]
    out = np.zeros(cnt.shape[0], dtype=bool)
    ok = nm > 0
    out[ok] = (alt[ok].astype(np.float64) / (2.0 * nm[ok].astype(np.float64))) > 0.5
    return out


class pcg_image_scope:
    """`with pcg_image_scope():` -- inside, the two images of the training payload that `he_pcg_bed` / `rrblup_pcg_bed` build
    (SNP-major and sample-major, 40 GB each at BASELINE configs[4]) stay in HBM between the calls and a second call on the same
    DEVICE payload, kept rows and training samples reuses them (`jx gs -rrBLUP -rr-solver pcg`: lambda by Haseman-Elston, then the
    solve).  The caller guarantees the payload does not change inside the scope; leaving it frees the images (`jx_pcg_image_scope`)."""

    def __enter__(self):
        check(lib().jx_pcg_image_scope(1))
        return self

    def __exit__(self, et, ev, tb):
        lib().jx_pcg_image_scope(0)
        return False


def rrblup_pcg_bed(prefix, train_sample_indices, y_train, test_sample_indices=None, train_pred_local_indices=None,
                   site_keep=None, lambda_value=10000.0, tol=1e-4, max_iter=100, block_rows=4096, std_eps=1e-12,
                   threads=0, progress_callback=None, progress_every=0, compute_trainvar=False, packed=None,
                   packed_n_samples=0, maf=None, row_flip=None, row_mean=None, row_inv_sd=None, blas_threads=0):
    """src/stats/rrblup.rs:3494-4307.  Marker effects of the standardised genotypes by Jacobi-preconditioned CG on the
    device (`jx_rrblup_pcg_packed`), the payload resident in HBM for the whole solve.  Returns the reference's tuple
    (pred_train (k,1), pred_test (t,1), pve_trainvar, converged, iters, rel_res, m_effective, pve_lambda_vc,
    k_trace_mean, beta f32 (m)).  `block_rows`, `threads`, `blas_threads` are accepted and ignored; the streaming-stats
    form (maf/row_flip with a prefix and no payload) loads the payload from the prefix."""
    import math
    if int(max_iter) == 0:
        raise RuntimeError("max_iter must be > 0")
    if not (math.isfinite(tol) and tol > 0.0):
        raise RuntimeError("tol must be finite and > 0")
    if not (math.isfinite(std_eps) and std_eps > 0.0):
        raise RuntimeError("std_eps must be finite and > 0")
    if (not math.isfinite(lambda_value)) or lambda_value < 0.0:
        raise RuntimeError("lambda_value must be finite and >= 0")
    f32 = np.float32
    if packed is not None:
        if maf is None:
            raise RuntimeError("rrblup_pcg_bed: packed payload path requires `maf` argument.")
        if row_flip is None:
            raise RuntimeError("rrblup_pcg_bed: packed payload path requires `row_flip` argument.")
        if int(packed_n_samples) == 0:
            raise RuntimeError("rrblup_pcg_bed: packed payload path requires packed_n_samples > 0.")
        n_samples = int(packed_n_samples)
        if _is_device_tensor(packed):        # a payload that already lives in HBM is used in place (no host copy of 50 GB)
            if packed.dim() != 2:
                raise RuntimeError("packed BED payload must be 2D (m, bytes_per_snp).")
            pk = packed.contiguous()
        else:
            pk = _c(packed, np.uint8)
            if pk.ndim != 2:
                raise RuntimeError("packed BED payload must be 2D (m, bytes_per_snp).")
    elif maf is not None or row_flip is not None or int(packed_n_samples) > 0:
        if maf is None:
            raise RuntimeError("rrblup_pcg_bed: streaming stats path requires `maf` argument.")
        if row_flip is None:
            raise RuntimeError("rrblup_pcg_bed: streaming stats path requires `row_flip` argument.")
        if int(packed_n_samples) == 0:
            raise RuntimeError("rrblup_pcg_bed: streaming stats path requires packed_n_samples > 0.")
        if not str(prefix).strip():
            raise RuntimeError("rrblup_pcg_bed: streaming stats path requires non-empty prefix.")
        from .bed import read_bed_payload
        pk, n_file, _bim = read_bed_payload(str(prefix))
        n_samples = int(packed_n_samples)
        if n_file != n_samples:
            raise RuntimeError(f"packed_n_samples mismatch: got {n_samples}, BED has {n_file}")
    else:
        pk, _miss, maf, _std, n_samples = load_bed_2bit_packed(prefix)
        row_flip = bed_packed_row_flip_mask(pk, n_samples)
    m_total = int(pk.shape[0])
    if m_total == 0:
        raise RuntimeError("No SNP rows found in BED input.")
    if pk.shape[1] != (n_samples + 3) // 4:
        raise RuntimeError(f"packed second dimension mismatch: got {pk.shape[1]}, expected {(n_samples + 3) // 4}")
    maf_full = _c(maf, f32).ravel()
    if maf_full.shape[0] != m_total:
        raise RuntimeError(f"maf length mismatch: got {maf_full.shape[0]}, expected {m_total}")
    flip_full = np.asarray(row_flip).astype(bool).ravel()
    if flip_full.shape[0] != m_total:
        raise RuntimeError(f"row_flip length mismatch: got {flip_full.shape[0]}, expected {m_total}")
    rows = None
    maf_keep, flip_keep = maf_full, flip_full
    if site_keep is not None:
        mask = np.asarray(site_keep).astype(bool).ravel()
        if mask.shape[0] != m_total:
            raise RuntimeError(f"site_keep length mismatch: got {mask.shape[0]}, expected {m_total}")
        keep_idx = np.nonzero(mask)[0].astype(np.int64)
        if keep_idx.shape[0] == 0:
            raise RuntimeError("No SNPs remained after applying site_keep mask.")
        if keep_idx.shape[0] != m_total:
            rows = keep_idx
            maf_keep = np.clip(maf_full[keep_idx], f32(0.0), f32(0.5)).astype(f32)
            flip_keep = flip_full[keep_idx]
    eff_m = int(maf_keep.shape[0])
    tr = _c(train_sample_indices, np.int64).ravel()
    if tr.size == 0:
        raise RuntimeError("train_sample_indices must not be empty.")
    if tr.min() < 0 or tr.max() >= n_samples:
        raise RuntimeError("train_sample_indices out of range")
    y = _c(y_train, np.float64).ravel()
    if y.shape[0] != tr.shape[0]:
        raise RuntimeError(f"y_train length mismatch: got {y.shape[0]}, expected {tr.shape[0]}")
    if not np.all(np.isfinite(y)):
        raise RuntimeError("y_train contains non-finite values.")
    te = np.zeros(0, dtype=np.int64) if test_sample_indices is None else _c(test_sample_indices, np.int64).ravel()
    if te.size and (te.min() < 0 or te.max() >= n_samples):
        raise RuntimeError("test_sample_indices out of range")
    pick = None
    if train_pred_local_indices is not None:
        pick = _c(train_pred_local_indices, np.int64).ravel()
        if pick.size and (pick.min() < 0 or pick.max() >= tr.shape[0]):
            raise RuntimeError("train_pred_local_indices out of range")
    n_train = int(tr.shape[0])
    lambda_use = f32(max(float(lambda_value), 1e-8))
    std_eps32 = f32(max(float(std_eps), 1e-12))
    # row standardisation (`rrblup_subset_or_validate_stats`, rrblup.rs:568-625)
    if (row_mean is None) != (row_inv_sd is None):
        raise RuntimeError("rrBLUP standardization requires row_mean and row_inv_sd together when overriding row stats.")
    if row_mean is not None:
        rm = _c(row_mean, f32).ravel()
        ri = _c(row_inv_sd, f32).ravel()
        if rm.shape[0] == eff_m and ri.shape[0] == eff_m:
            pass
        elif rm.shape[0] == m_total and ri.shape[0] == m_total and rows is not None:
            rm, ri = rm[rows], ri[rows]
        else:
            raise RuntimeError(f"External row_mean/row_inv_sd length mismatch: mean={rm.shape[0]}, inv={ri.shape[0]}, "
                               f"expected active={eff_m} or full={m_total}.")
        m_effective = int(np.count_nonzero(np.isfinite(ri) & (ri > 0)))
        if m_effective == 0:
            raise RuntimeError("rrBLUP standardization received zero effective markers from external row_inv_sd.")
    else:
        pq = np.clip(maf_keep, f32(0.0), f32(0.5)).astype(f32)
        rm = (f32(2.0) * pq).astype(f32)
        var = np.maximum((f32(2.0) * pq * (f32(1.0) - pq)).astype(f32), f32(0.0))
        good = var > std_eps32
        ri = np.zeros_like(var)
        ri[good] = (f32(1.0) / np.sqrt(var[good])).astype(f32)
        m_effective = int(np.count_nonzero(good))
    g0 = np.where(flip_keep, f32(2.0), f32(0.0)).astype(f32)
    g2 = np.where(flip_keep, f32(0.0), f32(2.0)).astype(f32)
    lut = np.zeros((eff_m, 4), dtype=f32)
    lut[:, 0] = (g0 - rm) * ri
    lut[:, 2] = (f32(1.0) - rm) * ri
    lut[:, 3] = (g2 - rm) * ri
    need_all = pick is None
    need_train = need_all or bool(compute_trainvar) or (pick is not None and pick.size > 0)
    beta = np.zeros(eff_m, dtype=f32)
    pred_tr_full = np.zeros(n_train, dtype=np.float64) if need_train else None
    pred_te = np.zeros(te.shape[0], dtype=np.float64)
    sc = np.zeros(8, dtype=np.float64)
    from .dist import distributed_pcg, shard_range
    shard = distributed_pcg()
    if shard is not None:
        # marker-sharded solve (dist.enable_distributed_pcg): this rank's contiguous range of the kept rows; the device layer
        # all-reduces Z'p and the iteration's scalars, beta is gathered in rank (= row) order afterwards
        import torch
        import torch.distributed as tdist
        rank, world = shard
        lo, hi = shard_range(eff_m, rank, world)
        if eff_m < world:
            # the same count on every rank: all of them stop here, before any collective (a rank left alone in the first
            # all-reduce would wait for ever)
            raise RuntimeError(f"distributed rrBLUP PCG: {eff_m} kept rows cannot be dealt over {world} ranks")
        rows_all = rows if rows is not None else np.arange(m_total, dtype=np.int64)
        if _is_device_tensor(pk):
            pk_s = pk[torch.from_numpy(rows_all[lo:hi]).to(pk.device)].contiguous() if rows is not None else pk[lo:hi]
        else:
            pk_s = np.ascontiguousarray(pk[rows_all[lo:hi]])
        lut_s = np.ascontiguousarray(lut[lo:hi])
        beta_s = np.zeros(hi - lo, dtype=f32)
        check(lib().jx_rrblup_pcg_packed(_payload(pk_s, n_samples)[1], hi - lo, n_samples, None, hi - lo, _p(lut_s), _p(tr), n_train, _p(y),
                                         _p(te) if te.size else None, int(te.shape[0]), float(lambda_value), float(tol),
                                         int(max_iter), _p(beta_s), _p(pred_tr_full), _p(pred_te) if te.size else None,
                                         _p(sc)))
        sizes = [shard_range(eff_m, r, world) for r in range(world)]
        width = max(b - a for a, b in sizes)
        pad = torch.zeros(width, dtype=torch.float32)
        pad[: hi - lo] = torch.from_numpy(beta_s)
        parts = [torch.zeros(width, dtype=torch.float32) for _ in range(world)]
        if tdist.get_backend() == "nccl":
            dev = torch.device("cuda", torch.cuda.current_device())
            parts = [t.to(dev) for t in parts]
            tdist.all_gather(parts, pad.to(dev))
            parts = [t.cpu() for t in parts]
        else:
            tdist.all_gather(parts, pad)
        for (a, b), t in zip(sizes, parts):
            beta[a:b] = t[: b - a].numpy()
    else:
        check(lib().jx_rrblup_pcg_packed(_payload(pk, n_samples)[1], m_total, n_samples, _p(rows), eff_m, _p(lut), _p(tr), n_train, _p(y),
                                         _p(te) if te.size else None, int(te.shape[0]), float(lambda_value), float(tol),
                                         int(max_iter), _p(beta), _p(pred_tr_full), _p(pred_te) if te.size else None,
                                         _p(sc)))
    converged, iters, rel_res, sum_ss = bool(sc[0] != 0.0), int(sc[1]), float(sc[2]), float(sc[3])
    if progress_callback is not None:
        try:
            progress_callback(iters, int(max_iter), rel_res)
        except TypeError:
            progress_callback(iters, int(max_iter))
    pve_trainvar = float("nan")
    if need_train:
        pred_train = pred_tr_full if need_all else pred_tr_full[pick]
        if compute_trainvar:
            resid = y - pred_tr_full
            if n_train > 1:
                vg = float(np.sum((pred_tr_full - pred_tr_full.mean()) ** 2) / (n_train - 1))
                ve = float(np.sum((resid - resid.mean()) ** 2) / (n_train - 1))
            else:
                vg = ve = 0.0
            den = vg + ve
            pve_trainvar = vg / den if (math.isfinite(den) and den > 0.0) else float("nan")
    else:
        pred_train = np.zeros(0, dtype=np.float64)
    k_trace_mean = sum_ss / (float(m_effective) * float(n_train)) if (n_train > 0 and m_effective > 0) else float("nan")
    pve_lambda_vc = float("nan")
    if math.isfinite(k_trace_mean) and k_trace_mean > 0.0 and m_effective > 0:
        dv = k_trace_mean + float(lambda_use) / float(m_effective)
        if math.isfinite(dv) and dv > 0.0:
            pve_lambda_vc = k_trace_mean / dv
    return (np.asarray(pred_train, dtype=np.float64).reshape(-1, 1), pred_te.reshape(-1, 1), pve_trainvar, converged,
            iters, rel_res, m_effective, pve_lambda_vc, k_trace_mean, beta)


def rrblup_exact_snp_packed(packed, n_samples, train_sample_indices, y_train, test_sample_indices=None,
                            train_pred_local_indices=None, site_keep=None, maf=None, row_flip=None, row_mean=None,
                            row_inv_sd=None, log10_lambda_low=-6.0, log10_lambda_high=6.0, reml_tol=1e-4, reml_max_iter=50,
                            sample_block=2048, std_eps=1e-12, threads=0, blas_threads=0):
    """src/stats/rrblup.rs:3157-3490.  Exact marker-space rrBLUP (the reference's route up to 15 000 markers): SNP-space
    Gram matrix of the standardised training genotypes, eigendecomposition, REML over log10 lambda by Brent on the
    spectrum, marker effects and predictions — all on the device (`jx_rrblup_exact_snp_packed`).  Returns the reference's
    tuple (pred_train (k,1), pred_test (t,1), pve_trainvar, lambda, reml, (var_g, sigma_e2), m_effective, y_mean,
    beta f32 (m), row_mean f32 (m), row_inv_sd f32 (m), eigensolver name).  `sample_block`, `threads`, `blas_threads` are
    accepted and ignored."""
    import math
    n_samples = int(n_samples)
    if n_samples == 0:
        raise RuntimeError("n_samples must be > 0")
    if not (math.isfinite(log10_lambda_low) and math.isfinite(log10_lambda_high)):
        raise RuntimeError("rrblup_exact_snp_packed requires finite log10(lambda) bounds.")
    if not (math.isfinite(reml_tol) and reml_tol > 0.0):
        raise RuntimeError("rrblup_exact_snp_packed requires finite reml_tol > 0.")
    if int(reml_max_iter) == 0:
        raise RuntimeError("rrblup_exact_snp_packed requires reml_max_iter > 0.")
    if not (math.isfinite(std_eps) and std_eps > 0.0):
        raise RuntimeError("rrblup_exact_snp_packed requires finite std_eps > 0.")
    f32 = np.float32
    pk = _c(packed, np.uint8)
    if pk.ndim != 2:
        raise RuntimeError("packed must be 2D (m, bytes_per_snp).")
    m_total = int(pk.shape[0])
    if pk.shape[1] != (n_samples + 3) // 4:
        raise RuntimeError(f"packed second dimension mismatch: got {pk.shape[1]}, expected {(n_samples + 3) // 4}")
    if maf is None:
        raise RuntimeError("rrblup_exact_snp_packed requires `maf` argument.")
    if row_flip is None:
        raise RuntimeError("rrblup_exact_snp_packed requires `row_flip` argument.")
    maf_full = _c(maf, f32).ravel()
    if maf_full.shape[0] != m_total:
        raise RuntimeError(f"maf length mismatch: got {maf_full.shape[0]}, expected {m_total}")
    flip_full = np.asarray(row_flip).astype(bool).ravel()
    if flip_full.shape[0] != m_total:
        raise RuntimeError(f"row_flip length mismatch: got {flip_full.shape[0]}, expected {m_total}")
    rows = None
    maf_keep, flip_keep = maf_full, flip_full
    if site_keep is not None:
        mask = np.asarray(site_keep).astype(bool).ravel()
        if mask.shape[0] != m_total:
            raise RuntimeError(f"site_keep length mismatch: got {mask.shape[0]}, expected {m_total}")
        keep_idx = np.nonzero(mask)[0].astype(np.int64)
        if keep_idx.shape[0] == 0:
            raise RuntimeError("No SNPs remained after applying site_keep mask.")
        if keep_idx.shape[0] != m_total:
            rows = keep_idx
            maf_keep = np.clip(maf_full[keep_idx], f32(0.0), f32(0.5)).astype(f32)
            flip_keep = flip_full[keep_idx]
    eff_m = int(maf_keep.shape[0])
    if eff_m == 0:
        raise RuntimeError("rrblup_exact_snp_packed received zero active markers.")
    tr = _c(train_sample_indices, np.int64).ravel()
    if tr.size and (tr.min() < 0 or tr.max() >= n_samples):
        raise RuntimeError("train_sample_indices out of range")
    n_train = int(tr.shape[0])
    if n_train <= 1:
        raise RuntimeError("rrblup_exact_snp_packed requires at least two training samples.")
    y = _c(y_train, np.float64).ravel()
    if y.shape[0] != n_train:
        raise RuntimeError(f"y_train length mismatch: got {y.shape[0]}, expected {n_train}")
    if not np.all(np.isfinite(y)):
        raise RuntimeError("y_train contains non-finite values.")
    te = np.zeros(0, dtype=np.int64) if test_sample_indices is None else _c(test_sample_indices, np.int64).ravel()
    if te.size and (te.min() < 0 or te.max() >= n_samples):
        raise RuntimeError("test_sample_indices out of range")
    pick = None
    if train_pred_local_indices is not None:
        pick = _c(train_pred_local_indices, np.int64).ravel()
        if pick.size and (pick.min() < 0 or pick.max() >= n_train):
            raise RuntimeError("train_pred_local_indices out of range")
    std_eps32 = f32(max(float(std_eps), 1e-12))
    # row standardisation (`rrblup_subset_or_validate_stats`, rrblup.rs:568-625)
    if (row_mean is None) != (row_inv_sd is None):
        raise RuntimeError("rrBLUP standardization requires row_mean and row_inv_sd together when overriding row stats.")
    if row_mean is not None:
        rm = _c(row_mean, f32).ravel()
        ri = _c(row_inv_sd, f32).ravel()
        if rm.shape[0] == eff_m and ri.shape[0] == eff_m:
            pass
        elif rm.shape[0] == m_total and ri.shape[0] == m_total and rows is not None:
            rm, ri = rm[rows], ri[rows]
        else:
            raise RuntimeError(f"External row_mean/row_inv_sd length mismatch: mean={rm.shape[0]}, inv={ri.shape[0]}, "
                               f"expected active={eff_m} or full={m_total}.")
        m_effective = int(np.count_nonzero(np.isfinite(ri) & (ri > 0)))
    else:
        pq = np.clip(maf_keep, f32(0.0), f32(0.5)).astype(f32)
        rm = (f32(2.0) * pq).astype(f32)
        var = np.maximum((f32(2.0) * pq * (f32(1.0) - pq)).astype(f32), f32(0.0))
        good = var > std_eps32
        ri = np.zeros_like(var)
        ri[good] = (f32(1.0) / np.sqrt(var[good])).astype(f32)
        m_effective = int(np.count_nonzero(good))
    if m_effective == 0:
        raise RuntimeError("rrblup_exact_snp_packed found zero effective markers after standardization.")
    g0 = np.where(flip_keep, f32(2.0), f32(0.0)).astype(f32)
    g2 = np.where(flip_keep, f32(0.0), f32(2.0)).astype(f32)
    lut = np.zeros((eff_m, 4), dtype=f32)
    lut[:, 0] = (g0 - rm) * ri
    lut[:, 2] = (f32(1.0) - rm) * ri
    lut[:, 3] = (g2 - rm) * ri
    beta = np.zeros(eff_m, dtype=f32)
    need_train = pick is None or pick.size > 0
    pred_tr_full = np.zeros(n_train, dtype=np.float64) if need_train else None
    pred_te = np.zeros(te.shape[0], dtype=np.float64)
    sc = np.zeros(8, dtype=np.float64)
    check(lib().jx_rrblup_exact_snp_packed(_p(pk), m_total, n_samples, _p(rows), eff_m, _p(lut), _p(tr), n_train, _p(y),
                                           _p(te) if te.size else None, int(te.shape[0]), float(log10_lambda_low),
                                           float(log10_lambda_high), float(reml_tol), int(reml_max_iter), _p(beta),
                                           _p(pred_tr_full), _p(pred_te) if te.size else None, _p(sc)))
    if pick is None:
        pred_train = pred_tr_full
    elif pick.size:
        pred_train = pred_tr_full[pick]
    else:
        pred_train = np.zeros(0, dtype=np.float64)
    return (np.asarray(pred_train, dtype=np.float64).reshape(-1, 1), pred_te.reshape(-1, 1), float(sc[0]), float(sc[1]),
            float(sc[2]), (float(sc[3]), float(sc[4])), m_effective, float(sc[7]), beta, rm.astype(f32), ri.astype(f32),
            "jxgpu_eigh_f64")


def _he_solve_2x2(a00, a01, a11, b0, b1):
    """src/stats/he.rs:874-897."""
    import math
    if not all(math.isfinite(v) for v in (a00, a01, a11, b0, b1)):
        raise RuntimeError("HE 2x2 solve received non-finite inputs")
    det = a00 * a11 - a01 * a01
    det_scale = max(abs(a00) + abs(a11) + 2.0 * abs(a01), 1.0)
    det_floor = det_scale * det_scale * float(np.finfo(np.float64).eps)
    if (not math.isfinite(det)) or abs(det) <= det_floor:
        raise RuntimeError(f"HE 2x2 solve is singular/ill-conditioned: det={det}, floor={det_floor}")
    x0 = (b0 * a11 - b1 * a01) / det
    x1 = (a00 * b1 - a01 * b0) / det
    if not (math.isfinite(x0) and math.isfinite(x1)):
        raise RuntimeError("HE 2x2 solve produced non-finite outputs")
    return x0, x1


def _he_project_nnls_2x2(a00, a01, a11, b0, b1, x0u, x1u):
    """src/stats/he.rs:815-871 -> (sigma_g2, sigma_e2, projected, boundary status 0..3)."""
    import math

    def obj(x0, x1):
        r0 = a00 * x0 + a01 * x1 - b0
        r1 = a01 * x0 + a11 * x1 - b1
        return r0 * r0 + r1 * r1
    cands = [(x0u, x1u, 0)]
    c1 = a01 * a01 + a11 * a11
    if math.isfinite(c1) and c1 > 0.0:
        cands.append((0.0, max((a01 * b0 + a11 * b1) / c1, 0.0), 1))
    c0 = a00 * a00 + a01 * a01
    if math.isfinite(c0) and c0 > 0.0:
        cands.append((max((a00 * b0 + a01 * b1) / c0, 0.0), 0.0, 2))
    cands.append((0.0, 0.0, 3))
    best = (0.0, 0.0, float("inf"), 3)
    for x0, x1, stt in cands:
        if not (math.isfinite(x0) and math.isfinite(x1)) or x0 < 0.0 or x1 < 0.0:
            continue
        o = obj(x0, x1)
        if math.isfinite(o) and o < best[2]:
            best = (x0, x1, o, stt)
    if not math.isfinite(best[2]):
        return 0.0, 0.0, True, 3
    tolp = 1e-10 * max(abs(x0u), abs(x1u), 1.0)
    projected = best[3] != 0 or abs(best[0] - x0u) > tolp or abs(best[1] - x1u) > tolp
    return best[0], best[1], projected, best[3]


def he_pcg_bed(prefix, train_sample_indices, y_train, site_keep=None, trace_samples=32, trace_probe_batch=64, tol=1e-6,
               max_iter=32, block_rows=4096, std_eps=1e-12, use_train_maf=True, exact_trace_debug=False,
               exact_trace_max_n=256, threads=0, seed=20260512, packed=None, packed_n_samples=0, maf=None, row_flip=None,
               row_source_indices=None, x_cov=None, blas_threads=0, mmap_window_mb=None):
    """src/stats/he.rs:2073-2636: Haseman-Elston variance components of y = g + e, g ~ N(0, sigma_g2 K), K the standardised
    GRM of the training samples applied matrix-free on the device (`jx_he_traces_packed`: two streaming passes per
    probe), Hutchinson traces with the reference's splitmix64 probes.  Returns the reference's 12-tuple (sigma_g2,
    sigma_e2, h2, converged, iters, rel_res, m_effective, tr_k2, y_ky, y_y, lambda, tr_k2_solve).  The packed-payload
    form and the prefix form are built (the metadata-streaming form maps onto them: rows = row_source_indices)."""
    import math
    f32 = np.float32
    if int(trace_samples) == 0:
        raise RuntimeError("trace_samples must be > 0")
    if int(trace_probe_batch) == 0:
        raise RuntimeError("trace_probe_batch must be > 0")
    if int(max_iter) == 0:
        raise RuntimeError("max_iter must be > 0")
    if not (math.isfinite(tol) and tol > 0.0):
        raise RuntimeError("tol must be finite and > 0")
    if not (math.isfinite(std_eps) and std_eps > 0.0):
        raise RuntimeError("std_eps must be finite and > 0")
    ext = packed is not None or int(packed_n_samples) > 0
    meta = row_source_indices is not None
    if ext and meta:
        raise RuntimeError("he_pcg_bed: provide either packed payload inputs or row_source_indices metadata streaming "
                           "inputs, not both.")
    if (not ext) and (not meta) and (maf is not None or row_flip is not None):
        raise RuntimeError("he_pcg_bed: external maf/row_flip without packed payload requires row_source_indices for "
                           "metadata streaming.")
    rows = None
    if ext:
        if packed is None:
            raise RuntimeError("he_pcg_bed: packed payload path requires `packed` argument.")
        if maf is None:
            raise RuntimeError("he_pcg_bed: packed payload path requires `maf` argument.")
        if row_flip is None:
            raise RuntimeError("he_pcg_bed: packed payload path requires `row_flip` argument.")
        if int(packed_n_samples) == 0:
            raise RuntimeError("he_pcg_bed: packed payload path requires packed_n_samples > 0.")
        n_samples = int(packed_n_samples)
        if _is_device_tensor(packed):        # a payload that already lives in HBM is used in place
            if packed.dim() != 2:
                raise RuntimeError("packed BED payload must be 2D (m, bytes_per_snp).")
            pk = packed.contiguous()
        else:
            pk = _c(packed, np.uint8)
            if pk.ndim != 2:
                raise RuntimeError("packed BED payload must be 2D (m, bytes_per_snp).")
    elif meta:
        if site_keep is not None:
            raise RuntimeError("he_pcg_bed: metadata streaming path does not accept site_keep; subset rows via "
                               "row_source_indices instead.")
        if maf is None:
            raise RuntimeError("he_pcg_bed: metadata streaming path requires `maf` argument.")
        if row_flip is None:
            raise RuntimeError("he_pcg_bed: metadata streaming path requires `row_flip` argument.")
        from .bed import read_bed_payload
        pk, n_samples, _bim = read_bed_payload(str(prefix))
        rows = _c(row_source_indices, np.int64).ravel()
        if rows.size == 0:
            raise RuntimeError("row_source_indices must not be empty for metadata streaming path.")
        if rows.min() < 0:
            raise RuntimeError("row_source_indices must be non-negative.")
    else:
        pk, _miss, maf, _std, n_samples = load_bed_2bit_packed(prefix)
        row_flip = bed_packed_row_flip_mask(pk, n_samples)
    m_total = int(pk.shape[0])
    if m_total == 0:
        raise RuntimeError("No SNP rows found in BED input.")
    if pk.shape[1] != (n_samples + 3) // 4:
        raise RuntimeError(f"packed second dimension mismatch: got {pk.shape[1]}, expected {(n_samples + 3) // 4}")
    maf_in = _c(maf, f32).ravel()
    flip_in = np.asarray(row_flip).astype(bool).ravel()
    if meta:
        if maf_in.shape[0] != rows.shape[0] or flip_in.shape[0] != rows.shape[0]:
            raise RuntimeError("metadata length mismatch between row_source_indices, maf and row_flip")
        if rows.max() >= m_total:
            raise RuntimeError("row source index out of bounds")
        maf_keep, flip_keep = maf_in, flip_in
    else:
        if maf_in.shape[0] != m_total:
            raise RuntimeError(f"maf length mismatch: got {maf_in.shape[0]}, expected {m_total}")
        if flip_in.shape[0] != m_total:
            raise RuntimeError(f"row_flip length mismatch: got {flip_in.shape[0]}, expected {m_total}")
        maf_keep, flip_keep = maf_in, flip_in
        if site_keep is not None:
            mask = np.asarray(site_keep).astype(bool).ravel()
            if mask.shape[0] != m_total:
                raise RuntimeError(f"site_keep length mismatch: got {mask.shape[0]}, expected {m_total}")
            keep_idx = np.nonzero(mask)[0].astype(np.int64)
            if keep_idx.shape[0] == 0:
                raise RuntimeError("No SNPs remained after applying site_keep mask.")
            if keep_idx.shape[0] != m_total:
                rows = keep_idx
                maf_keep, flip_keep = maf_in[keep_idx], flip_in[keep_idx]
    eff_m = int(maf_keep.shape[0])
    tr = _c(train_sample_indices, np.int64).ravel()
    if tr.size == 0:
        raise RuntimeError("train_sample_indices must not be empty.")
    if tr.min() < 0 or tr.max() >= n_samples:
        raise RuntimeError("train_sample_indices out of range")
    y = _c(y_train, np.float64).ravel()
    if y.shape[0] != tr.shape[0]:
        raise RuntimeError(f"y_train length mismatch: got {y.shape[0]}, expected {tr.shape[0]}")
    if not np.all(np.isfinite(y)):
        raise RuntimeError("y_train contains non-finite values.")
    n = int(tr.shape[0])
    xc, p_cov = None, 0
    if x_cov is not None:
        xa = _c(x_cov, np.float64)
        if xa.ndim != 2:
            raise RuntimeError("x_cov must be 2D (n, p_cov)")
        if xa.shape[1] == 0:
            raise RuntimeError("x_cov must have at least one column")
        if xa.shape[0] == n_samples:
            xc = np.ascontiguousarray(xa[tr])
        elif xa.shape[0] == n:
            xc = xa
        else:
            raise RuntimeError(f"x_cov rows mismatch: got {xa.shape[0]}, expected either n_samples={n_samples} or n_train={n}")
        p_cov = int(xc.shape[1])
    # row standardisation (he.rs:454-640): oriented minor frequency, or the training samples' own frequency
    if not np.all(np.isfinite(maf_keep)):
        raise RuntimeError("row_maf contains non-finite values")
    af = np.clip(maf_keep, f32(0.0), f32(1.0))
    pfr = np.where(af <= f32(0.5), af, np.where(flip_keep, f32(1.0) - af, af)).astype(f32)
    if use_train_maf:
        if rows is None:
            pk_rows = pk
        elif _is_device_tensor(pk):
            import torch
            pk_rows = pk[torch.from_numpy(rows).to(pk.device)]
        else:
            pk_rows = np.ascontiguousarray(pk[rows])
        cnt = bed_row_counts(pk_rows, n_samples, tr).astype(np.int64)
        del pk_rows
        nm = n - cnt[:, 0]
        alt = cnt[:, 1] + 2 * cnt[:, 2]
        dos = np.where(flip_keep, 2 * nm - alt, alt)
        okc = nm > 0
        pt = np.zeros_like(pfr)
        pt[okc] = dos[okc].astype(f32) / (f32(2.0) * nm[okc].astype(f32))
        pfr = np.where(okc, pt, pfr).astype(f32)
    pfr = np.clip(pfr, f32(0.0), f32(1.0))
    rm = (f32(2.0) * pfr).astype(f32)
    var = np.maximum((f32(2.0) * pfr * (f32(1.0) - pfr)).astype(f32), f32(0.0))
    good = var > f32(max(float(std_eps), 1e-12))
    ri = np.zeros_like(var)
    ri[good] = (f32(1.0) / np.sqrt(var[good])).astype(f32)
    m_effective = int(np.count_nonzero(good))
    if m_effective == 0:
        raise RuntimeError("No effective SNPs after std_eps filtering")
    g0 = np.where(flip_keep, f32(2.0), f32(0.0)).astype(f32)
    g2 = np.where(flip_keep, f32(0.0), f32(2.0)).astype(f32)
    lut = np.zeros((eff_m, 4), dtype=f32)
    lut[:, 0] = (g0 - rm) * ri
    lut[:, 2] = (f32(1.0) - rm) * ri
    lut[:, 3] = (g2 - rm) * ri
    exact = bool(exact_trace_debug) and n <= max(int(exact_trace_max_n), 1)
    out5 = np.zeros(5, dtype=np.float64)
    check(lib().jx_he_traces_packed(_payload(pk, n_samples)[1], m_total, n_samples, _p(rows), eff_m, _p(lut), _p(tr), n, _p(y), _p(xc), p_cov,
                                    int(trace_samples), int(seed) & ((1 << 64) - 1), 1 if exact else 0,
                                    float(m_effective), _p(out5)))
    y_ky, y_y, tr_k, tr_k2, tr_p = (float(v) for v in out5)
    if not (math.isfinite(tr_k) and tr_k > 0.0):
        raise RuntimeError(f"estimated Tr(PKP) is invalid: {tr_k}. Try increasing trace_samples.")
    if not (math.isfinite(tr_k2) and tr_k2 > 0.0):
        raise RuntimeError(f"estimated Tr((PKP)^2) is invalid: {tr_k2}. Try increasing trace_samples.")
    tr_k2_solve = max(tr_k2, (tr_k * tr_k) / tr_p + tr_p * 1e-6)
    if tr_k2_solve > tr_k2 * 1.05:
        raise RuntimeError(f"Tr((PKP)^2) stochastic estimate violates PSD bound too much: raw={tr_k2}, "
                           f"adjusted={tr_k2_solve}. Increase trace_samples.")
    sg_u, se_u = _he_solve_2x2(tr_k2_solve, tr_k, tr_p, y_ky, y_y)
    sg, se, _proj, _status = _he_project_nnls_2x2(tr_k2_solve, tr_k, tr_p, y_ky, y_y, sg_u, se_u)
    r0 = tr_k2_solve * sg + tr_k * se - y_ky
    r1 = tr_k * sg + tr_p * se - y_y
    rel_res = math.sqrt(r0 * r0 + r1 * r1) / max(math.sqrt(y_ky * y_ky + y_y * y_y), 1e-20)
    converged = math.isfinite(rel_res) and rel_res <= max(float(tol), 1e-12)
    den = sg + se
    h2 = sg / den if (math.isfinite(den) and den > 0.0) else float("nan")
    lam = se / sg if (math.isfinite(sg) and sg > 0.0) else float("inf")
    return (sg, se, h2, bool(converged), 1, rel_res, min(m_effective, eff_m), tr_k2, y_ky, y_y, lam, tr_k2_solve)


# ------------------------------------------------------------------------------------------------
# Names the reference's Python layer imports unconditionally beside the hot path (python/janusx/pyBLUP/assoc.py:207-218)
# for a model this library does not replace (FastLMM, low-rank GRM): present so that the module imports, loud when called.
# ------------------------------------------------------------------------------------------------

def _out_of_scope(name, what):
    def stub(*_args, **_kwargs):
        raise RuntimeError(f"{name}: {what} is outside the mixed-model hot path this library replaces "
                           "(GRM -> eigendecomposition -> REML -> per-SNP scan); use the reference's CPU extension for it")
    stub.__name__ = name
    stub.__doc__ = f"Out of scope ({what}); raises RuntimeError."
    return stub


fastlmm_prepare_lowrank_f64 = _out_of_scope("fastlmm_prepare_lowrank_f64", "the FastLMM low-rank model")
fastlmm_assoc_from_snp_f32 = _out_of_scope("fastlmm_assoc_from_snp_f32", "the FastLMM low-rank model")
fastlmm_reml_chunk_f32 = _out_of_scope("fastlmm_reml_chunk_f32", "the FastLMM low-rank model")
fastlmm_reml_null_f32 = _out_of_scope("fastlmm_reml_null_f32", "the FastLMM low-rank model")
fastlmm_assoc_chunk_f32 = _out_of_scope("fastlmm_assoc_chunk_f32", "the FastLMM low-rank model")

