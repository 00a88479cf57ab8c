"""Association result TSV (format contract of src/io/assoc2tsv.rs:45-57, 430-548; src/math/linalg.rs:327-340).

Header  chrom pos snp allele0 allele1 af miss beta se chisq pwald [plrt | lambda ml plrt]
Row     af/miss/beta/se `{:.4}`; chisq = (beta/se)^2 as `{:.4e}` or `NaN`; p `{:.4e}`; invalid beta/se -> p = 1
        (`sanitize_assoc_pvalue`, linalg.rs:111-121); Rust float text: `NaN`, `inf`, exponent without padding.
Files are written to a temp path and renamed (python/janusx/assoc/workflow.py:833-845).
"""
from __future__ import annotations

import math
import os

MIN_POSITIVE = 2.2250738585072014e-308
HEADER3 = "chrom\tpos\tsnp\tallele0\tallele1\taf\tmiss\tbeta\tse\tchisq\tpwald\n"
HEADER4 = "chrom\tpos\tsnp\tallele0\tallele1\taf\tmiss\tbeta\tse\tchisq\tpwald\tplrt\n"
HEADER6 = "chrom\tpos\tsnp\tallele0\tallele1\taf\tmiss\tbeta\tse\tchisq\tpwald\tlambda\tml\tplrt\n"


def fmt_e6(v: float) -> str:
    if v != v:
        return "NaN"
    if v in (math.inf, -math.inf):
        return "inf" if v > 0 else "-inf"
    mant, ex = f"{v:.6e}".split("e")
    return f"{mant}e{int(ex)}"


def fmt_e4(v: float) -> str:
    if v != v:
        return "NaN"
    if v in (math.inf, -math.inf):
        return "inf" if v > 0 else "-inf"
    mant, ex = f"{v:.4e}".split("e")
    return f"{mant}e{int(ex)}"


def fmt_f4(v: float) -> str:
    if v != v:
        return "NaN"
    if v in (math.inf, -math.inf):
        return "inf" if v > 0 else "-inf"
    return f"{v:.4f}"


def resolve_snp_name(snp: str, chrom: str, pos) -> str:
    """src/stats/lmm.rs:1952-1958."""
    return snp if (snp and snp != ".") else f"{chrom}_{pos}"


def format_row(chrom, pos, snp, a0, a1, af, miss, beta, se, p, plrt=None, lmm2=None) -> str:
    if math.isfinite(beta) and math.isfinite(se) and se > 0.0:
        z = beta / se
        chisq = z * z
        pv = min(max(p, MIN_POSITIVE), 1.0) if math.isfinite(p) else 1.0
    else:
        chisq = math.nan
        pv = 1.0
    row = (f"{chrom}\t{pos}\t{resolve_snp_name(snp, chrom, pos)}\t{a0}\t{a1}\t{fmt_f4(float(af))}\t"
           f"{fmt_f4(float(miss))}\t{fmt_f4(beta)}\t{fmt_f4(se)}\t{fmt_e4(chisq)}\t{fmt_e4(pv)}")
    if lmm2 is not None:  # Lmm2_6: lambda, ml `{:.6e}`, plrt `{:.4e}` (assoc2tsv.rs:500-512)
        row += f"\t{fmt_e6(float(lmm2[0]))}\t{fmt_e6(float(lmm2[1]))}\t{fmt_e4(float(lmm2[2]))}"
    elif plrt is not None:
        row += f"\t{fmt_e4(float(plrt))}"
    return row + "\n"


def write_assoc_tsv_python(path, chrom, pos, snp, a0, a1, af, miss, stats) -> int:
    """Pure-Python statement of the row format (kept as the checker of the native writer in the CPU tests)."""
    ncol = stats.shape[1]
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "w") as fh:
        if ncol not in (3, 4, 6):
            raise RuntimeError(f"unsupported GWAS result column count: {ncol} (expected 3, 4, or 6)")
        fh.write(HEADER6 if ncol == 6 else (HEADER4 if ncol == 4 else HEADER3))
        buf = []
        for i in range(stats.shape[0]):
            buf.append(format_row(chrom[i], pos[i], snp[i], a0[i], a1[i], af[i], miss[i], float(stats[i, 0]),
                                  float(stats[i, 1]), float(stats[i, 2]), stats[i, 3] if ncol == 4 else None,
                                  stats[i, 3:6] if ncol == 6 else None))
            if len(buf) >= 8192:
                fh.write("".join(buf))
                buf = []
        fh.write("".join(buf))
    os.replace(tmp, path)
    return int(stats.shape[0])


def _native_rows(tmp, chrom, pos, snp, a0, a1, af, miss, stats, append, miss_count=False, resolve=True) -> int:
    """resolve: replace an empty / "." SNP name by chrom_pos -- what the reference's BED streaming routes do while they read
    the BIM (`resolve_snp_name`, src/stats/lmm.rs:1242, 1952-1958); its entry points that take the metadata as lists print
    the names as given (src/io/assoc2tsv.rs:430-548)."""
    import ctypes as C

    import numpy as np

    from ._lib import lib
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    if stats.ndim != 2 or stats.shape[1] not in (3, 4, 6):
        raise RuntimeError(f"unsupported GWAS result column count: {stats.shape[1] if stats.ndim == 2 else stats.shape} "
                           "(expected 3, 4, or 6)")
    rows = int(stats.shape[0])
    prefixes = [f"{chrom[i]}\t{pos[i]}\t{resolve_snp_name(snp[i], chrom[i], pos[i]) if resolve else snp[i]}\t{a0[i]}\t{a1[i]}"
                .encode() for i in range(rows)]
    off = np.zeros(rows + 1, dtype=np.int64)
    if rows:
        np.cumsum([len(p) for p in prefixes], out=off[1:])
    blob = b"".join(prefixes)
    af32 = np.ascontiguousarray(af, dtype=np.float32)
    miss32 = np.ascontiguousarray(miss, dtype=np.float32)
    if af32.shape[0] < rows or miss32.shape[0] < rows:
        raise RuntimeError("af / miss shorter than the result table")
    written = lib().jx_assoc_tsv_append(tmp.encode(), C.cast(C.c_char_p(blob), C.c_void_p), off.ctypes.data, rows,
                                        af32.ctypes.data, miss32.ctypes.data, stats.ctypes.data, int(stats.shape[1]),
                                        (1 if append else 0) | (2 if miss_count else 0))
    if written < 0:
        msg = lib().jx_last_error()
        raise RuntimeError(msg.decode("utf-8", "replace") if msg else "jx_assoc_tsv_append failed")
    return int(written)


def write_assoc_tsv(path, chrom, pos, snp, a0, a1, af, miss, stats, resolve=True) -> int:
    """stats (rows, 3|4|6) f64 [beta, se, p(, plrt | , lambda, ml, plrt)] in BED order of the kept SNPs.
    The numeric columns are formatted and written by the native writer (`jx_assoc_tsv_append`, the counterpart of the
    reference's Rust formatter, src/io/assoc2tsv.rs:430-548); only the per-row `chrom pos snp a0 a1` prefix is put
    together here.  Returns rows written."""
    tmp = f"{path}.tmp.{os.getpid()}"
    written = _native_rows(tmp, chrom, pos, snp, a0, a1, af, miss, stats, False, False, resolve)
    os.replace(tmp, path)
    return written


def write_assoc_tsv_counts(path, chrom, pos, snp, a0, a1, af, miss_counts, stats, resolve=True) -> int:
    """`write_assoc_tsv` with the `miss` column holding COUNTS of missing samples, printed as integers
    (`AssocMissValue::Count`, src/io/assoc2tsv.rs:452-458: the reference's LM routes)."""
    tmp = f"{path}.tmp.{os.getpid()}"
    written = _native_rows(tmp, chrom, pos, snp, a0, a1, af, miss_counts, stats, False, True, resolve)
    os.replace(tmp, path)
    return written


class AsyncAssocTsvWriter:
    """Block-wise writer on its own thread, the counterpart of the reference's `AsyncTsvWriter` (src/stats/common.rs:374)
    behind its streaming scan (src/stats/lmm.rs:975-1477): `put(i0, stats_block)` hands the scan results of the kept rows
    [i0, i0 + len) over and returns at once; the thread builds the row prefixes and calls the native formatter (which
    releases the GIL) while the device scans the next block.  `close()` waits, renames the temp file and returns the row
    count; an error on the thread is raised there (and by the next `put`)."""

    def __init__(self, path, ncol, chrom, pos, snp, a0, a1, af, miss, depth=4, miss_count=False):
        import queue
        import threading
        if ncol not in (3, 4, 6):
            raise RuntimeError(f"unsupported GWAS result column count: {ncol} (expected 3, 4, or 6)")
        self.path, self.ncol = path, int(ncol)
        self.miss_count = bool(miss_count)      # `miss` holds counts, printed as integers (the reference's LM routes)
        self.meta = (chrom, pos, snp, a0, a1, af, miss)
        self.tmp = f"{path}.tmp.{os.getpid()}"
        self.rows, self.err, self.first = 0, None, True
        self.q = queue.Queue(maxsize=depth)
        self.th = threading.Thread(target=self._run, name="jx-tsv-writer", daemon=True)
        self.th.start()

    def _run(self):
        chrom, pos, snp, a0, a1, af, miss = self.meta
        while True:
            item = self.q.get()
            if item is None:
                return
            if self.err is not None:
                continue                      # drain after a failure
            i0, st = item
            i1 = i0 + st.shape[0]
            try:
                self.rows += _native_rows(self.tmp, chrom[i0:i1], pos[i0:i1], snp[i0:i1], a0[i0:i1], a1[i0:i1],
                                          af[i0:i1], miss[i0:i1], st, not self.first, self.miss_count)
                self.first = False
            except BaseException as e:        # noqa: BLE001 - reported by put() / close()
                self.err = e

    def put(self, i0, stats_block):
        import numpy as np
        if self.err is not None:
            raise self.err
        st = np.array(stats_block, dtype=np.float64, copy=True)
        if st.ndim != 2 or st.shape[1] != self.ncol:
            raise RuntimeError(f"result block has {st.shape} columns, the writer was opened for {self.ncol}")
        self.q.put((int(i0), st))

    def abort(self):
        """Stop the thread and remove the temp file (the scan failed: no partial table under the final name)."""
        self.err = self.err or RuntimeError("aborted")
        self.q.put(None)
        self.th.join()
        try:
            os.remove(self.tmp)
        except OSError:
            pass

    def close(self) -> int:
        import numpy as np
        self.q.put(None)
        self.th.join()
        if self.err is not None:
            try:
                os.remove(self.tmp)
            except OSError:
                pass
            raise self.err
        if self.first:                        # no block at all: header only
            _native_rows(self.tmp, [], [], [], [], [], np.zeros(0, np.float32), np.zeros(0, np.float32),
                         np.zeros((0, self.ncol)), False)
        os.replace(self.tmp, self.path)
        return self.rows
