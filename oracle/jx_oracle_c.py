"""ctypes binding of oracle/liboracle.so (C restatement) -- TEST INFRASTRUCTURE ONLY.
See jx_oracle.c for the reference citations. Used by tests/ (checker) and bench.py cpu_baseline."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "jx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        d, i, p = C.c_double, C.c_int, C.c_void_p
        _LIB.jxo_reml_loglike.restype = d
        _LIB.jxo_reml_loglike.argtypes = [d, p, p, p, p, i, i]
        _LIB.jxo_ml_loglike.restype = d
        _LIB.jxo_ml_loglike.argtypes = [d, p, p, p, p, i, i]
        _LIB.jxo_final_beta_se.restype = None
        _LIB.jxo_final_beta_se.argtypes = [d, p, p, p, p, i, i, p]
        _LIB.jxo_lmm_reml_null.restype = None
        _LIB.jxo_lmm_reml_null.argtypes = [p, p, p, i, i, d, d, i, d, p]
        _LIB.jxo_lmm_scan_rotated_block.restype = None
        _LIB.jxo_lmm_scan_rotated_block.argtypes = [p, i, i, p, p, p, i, d, d, d, i, i, d, i, d, p, p, i]
        _LIB.jxo_lmm2_null_ml.restype = None
        _LIB.jxo_lmm2_null_ml.argtypes = [p, p, p, i, i, d, d, i, d, i, d, p]
        _LIB.jxo_lmm2_scan_rotated_block.restype = None
        _LIB.jxo_lmm2_scan_rotated_block.argtypes = [p, i, i, p, p, p, i, d, d, d, i, i, d, d, p, i]
        _LIB.jxo_fvlmm_assoc_block.restype = None
        _LIB.jxo_fvlmm_assoc_block.argtypes = [p, i, i, i, p, p, p, p, d, i, p, i]
        _LIB.jxo_decode_rows_lut_f32.restype = None
        _LIB.jxo_decode_rows_lut_f32.argtypes = [p, C.c_int64, i, p, i, p, i, p, i]
        _LIB.jxo_row_counts.restype = None
        _LIB.jxo_row_counts.argtypes = [p, C.c_int64, i, C.c_int64, p, p, p]
        _LIB.jxo_max_threads.restype = i
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def reml_loglike(x, s, xcov, y, snp=None):
    s, xcov, y = _f64(s).ravel(), _f64(xcov), _f64(y).ravel()
    snp = None if snp is None else _f64(snp).ravel()
    return lib().jxo_reml_loglike(float(x), _ptr(s), _ptr(xcov), _ptr(y), _ptr(snp), len(y), xcov.shape[1])


def ml_loglike(x, s, xcov, y, snp=None):
    s, xcov, y = _f64(s).ravel(), _f64(xcov), _f64(y).ravel()
    snp = None if snp is None else _f64(snp).ravel()
    return lib().jxo_ml_loglike(float(x), _ptr(s), _ptr(xcov), _ptr(y), _ptr(snp), len(y), xcov.shape[1])


def final_beta_se(x, s, xcov, y, snp):
    s, xcov, y, snp = _f64(s).ravel(), _f64(xcov), _f64(y).ravel(), _f64(snp).ravel()
    out = np.zeros(3)
    lib().jxo_final_beta_se(float(x), _ptr(s), _ptr(xcov), _ptr(y), _ptr(snp), len(y), xcov.shape[1], _ptr(out))
    return tuple(out)


def lmm_reml_null(s, xcov, y, low, high, max_iter=50, tol=1e-2):
    s, xcov, y = _f64(s).ravel(), _f64(xcov), _f64(y).ravel()
    out = np.zeros(3)
    lib().jxo_lmm_reml_null(_ptr(s), _ptr(xcov), _ptr(y), len(y), xcov.shape[1], float(low), float(high),
                            int(max_iter), float(tol), _ptr(out))
    return tuple(out)


def lmm_scan_rotated_block(g_rot, s, xcov, y, low, high, max_iter, tol, warm=0, init=float("nan"),
                           nullml=None, threads=0, return_evals=False):
    g = np.ascontiguousarray(g_rot, dtype=np.float32)
    s, xcov, y = _f64(s).ravel(), _f64(xcov), _f64(y).ravel()
    rows, n = g.shape
    cols = 4 if nullml is not None else 3
    out = np.zeros((rows, cols))
    ev = np.zeros(rows, dtype=np.int32)
    lib().jxo_lmm_scan_rotated_block(_ptr(g), rows, n, _ptr(s), _ptr(xcov), _ptr(y), xcov.shape[1], float(low),
                                     float(high), float(tol), int(max_iter), int(warm), float(init),
                                     1 if nullml is not None else 0, float(nullml or 0.0), _ptr(out), _ptr(ev),
                                     int(threads))
    return (out, ev) if return_evals else out


def lmm_scan_rotated_chains(g_rot, s, xcov, y, low, high, max_iter, tol, chain_off, init=None, nullml=None,
                            return_evals=False):
    """The reference's default exact scan (`carry_warm_start`, src/stats/lmm.rs:134-161): rows [chain_off[c], chain_off[c + 1])
    of the rotated block are ONE sequential chain -- the per-worker state of `run_rotated_assoc_block_f32`
    (src/stats/reml.rs:69-105) lives for one block (one rayon piece of it) -- whose first valid SNP starts from `init`
    (`state.last_log10_lbd.or(init_log10_lbd)`) or, without it, from the interval midpoint."""
    g = np.ascontiguousarray(g_rot, dtype=np.float32)
    co = np.asarray(chain_off, dtype=np.int64)
    assert co[0] == 0 and co[-1] == g.shape[0] and np.all(np.diff(co) >= 0)
    out = np.zeros((g.shape[0], 4 if nullml is not None else 3))
    ev = np.zeros(g.shape[0], dtype=np.int32)
    start = float("nan") if init is None else float(init)
    for a, b in zip(co[:-1], co[1:]):
        if b > a:
            o, e = lmm_scan_rotated_block(g[a:b], s, xcov, y, low, high, max_iter, tol, warm=2, init=start, nullml=nullml,
                                          return_evals=True)
            out[a:b], ev[a:b] = o, e
    return (out, ev) if return_evals else out


def lmm2_null_ml(s, xcov, y, low, high, max_iter=30, tol=1e-2, init=None):
    s, xcov, y = _f64(s).ravel(), _f64(xcov), _f64(y).ravel()
    out = np.zeros(2)
    lib().jxo_lmm2_null_ml(_ptr(s), _ptr(xcov), _ptr(y), len(y), xcov.shape[1], float(low), float(high),
                           int(max_iter), float(tol), 1 if init is not None else 0,
                           float(init if init is not None else 0.0), _ptr(out))
    return float(out[0]), float(out[1])


def lmm2_scan_rotated_block(g_rot, s, xcov, y, low, high, max_iter, tol, nullml, init=None, threads=0):
    g = np.ascontiguousarray(g_rot, dtype=np.float32)
    s, xcov, y = _f64(s).ravel(), _f64(xcov), _f64(y).ravel()
    rows, n = g.shape
    out = np.zeros((rows, 6))
    lib().jxo_lmm2_scan_rotated_block(_ptr(g), rows, n, _ptr(s), _ptr(xcov), _ptr(y), xcov.shape[1], float(low),
                                      float(high), float(tol), int(max_iter), 1 if init is not None else 0,
                                      float(init if init is not None else 0.0), float(nullml), _ptr(out),
                                      int(threads))
    return out


def fvlmm_assoc_block(g_rot, w, num, cbuf, a_chol, ypy, df, threads=0):
    g = np.ascontiguousarray(g_rot, dtype=np.float32)
    rows, n = g.shape
    w = np.ascontiguousarray(w, dtype=np.float32)
    num = np.ascontiguousarray(num, dtype=np.float32)
    cbuf = np.ascontiguousarray(cbuf, dtype=np.float32)
    a = _f64(a_chol)
    out = np.zeros((rows, 3))
    lib().jxo_fvlmm_assoc_block(_ptr(g), rows, n, a.shape[0], _ptr(w), _ptr(num), _ptr(cbuf), _ptr(a), float(ypy),
                                int(df), _ptr(out), int(threads))
    return out


def decode_rows_lut(packed, n, lut, row_idx=None, center=False, threads=0):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    rows = lut.shape[0]
    ri = None if row_idx is None else np.ascontiguousarray(row_idx, dtype=np.int64)
    out = np.empty((rows, n), dtype=np.float32)
    lib().jxo_decode_rows_lut_f32(_ptr(packed), packed.shape[1], n, _ptr(ri), rows, _ptr(lut), 1 if center else 0,
                                  _ptr(out), int(threads))
    return out


def row_counts(packed, n):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    m = packed.shape[0]
    a = np.zeros(m, dtype=np.int64)
    b = np.zeros(m, dtype=np.int64)
    c = np.zeros(m, dtype=np.int64)
    lib().jxo_row_counts(_ptr(packed), packed.shape[1], n, m, _ptr(a), _ptr(b), _ptr(c))
    return a, b, c


def max_threads():
    return lib().jxo_max_threads()
