#!/usr/bin/env python3
"""Pins the oracle's per-SNP path with the REFERENCE'S OWN PYTHON MODEL LAYER.

Run in the BUILD container only:  python tests/golden/gen_reference_model_fixtures.py   -> tests/golden/reference_model.npz

The reference's model layer (`/root/reference/python/janusx/pyBLUP/assoc.py`: `LMM.__init__`, `LMM.from_spectral` ->
`_initialize_from_spectral` :1702-1876, `LMM.gwas` :1962, `LMM2.gwas`, `FastLMM.gwas`, `FvLMM.gwas / gwas_rotated` :2072-2180, and
`janusx/assoc/api.py::ASSOC` :518) is pure Python around calls into the native extension `janusx.janusx`.  The extension cannot be
built here (no Rust toolchain), so a stub module stands in for it whose functions are the ORACLE's restatements behind the
native signatures; the stub records every call (name, argument values, returned value).  What is stored is therefore:

  * what the REFERENCE'S code computes around the native calls: the design matrix it builds, the ridge it adds before the
    eigendecomposition, the null search's bounds / iterations / tolerance, lambda_0, ML0, LL0, sigma_g2, sigma_e2 (its own
    numpy `_lmm_profile_exact_vc`), trace_mean, PVE (its diagonal-scaled form), the scan bounds log10(lambda_0) +- 2 or (-5, 5),
    the LMM2 null ML optimum (its scipy bounded search over the native ML likelihood), FastLMM's PVE switch, the FvLMM cache
    handling;
  * the argument values each native call received and what it returned (the oracle's output for those arguments).

tests/test_oracle_golden.py asserts the oracle's own composition (`spectral_null_model`, the scans) against these values on the
CPU; tests/test_gpu_parity.py replays the recorded native calls through `janusx_amd.janusx` (the HIP library behind the same
names) and compares `pipeline.SpectralModel` with the reference-produced model attributes.

Nothing under /root/reference is copied: the fixture holds data only (inputs, recorded arguments, outputs).
"""
import importlib
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import jx_oracle as O  # noqa: E402

REF_PY = "/root/reference/python"
CALLS = []          # (name, args dict, result) in call order


def _rec(name, args, result):
    CALLS.append((name, args, result))
    return result


class _Cache:
    """What `fvlmm_assoc_prepare_cache_f32` hands back: an opaque object with `n` and `p` (assoc.py:2093-2097 reads both)."""

    def __init__(self, fv, s, xcov, y, log10_lbd):
        self.fv, self.s, self.xcov, self.y, self.log10_lbd = fv, s, xcov, y, float(log10_lbd)
        self.n, self.p = int(xcov.shape[0]), int(xcov.shape[1])


def build_stub():
    m = types.ModuleType("janusx.janusx")

    def rust_eigh_from_array_f64_inplace(a, threads=0, driver=None, jobz="V", require_lapack=False, diag_shift=0.0):
        a = np.array(a, dtype=np.float64, copy=True)
        if diag_shift:
            a.flat[:: a.shape[0] + 1] += float(diag_shift)
        w, v = O.eigh_sym(a)
        _rec("rust_eigh_from_array_f64_inplace", dict(a=a.copy(), driver=str(driver), jobz=str(jobz)), (w.copy(), v.copy()))
        return w, v, "oracle", "lapack_dsyevd", int(a.shape[0]), 0, 0, 0, True, 0.0

    def lmm_rotate_x_y_with_ut_f64(u_t, x, y, threads=0):
        xr, yr = O.lmm_rotate_x_y_with_ut(np.asarray(u_t, dtype=np.float32), np.asarray(x, dtype=np.float64),
                                          np.asarray(y, dtype=np.float64))
        return _rec("lmm_rotate_x_y_with_ut_f64", dict(u_t=np.array(u_t), x=np.array(x), y=np.array(y)), (xr, yr))

    def lmm_reml_null_f32(s, xcov, y_rot, low, high, max_iter=50, tol=1e-2):
        out = O.lmm_reml_null(s, xcov, y_rot, low, high, max_iter, tol)
        return _rec("lmm_reml_null_f32", dict(s=np.array(s), xcov=np.array(xcov), y_rot=np.array(y_rot), low=float(low),
                                              high=float(high), max_iter=int(max_iter), tol=float(tol)),
                    tuple(float(v) for v in out))

    def ml_loglike_null_f32(s, xcov, y_rot, log10_lbd):
        v = float(O.ml_loglike(float(log10_lbd), np.asarray(s), np.asarray(xcov), np.asarray(y_rot), None))
        return _rec("ml_loglike_null_f32", dict(log10_lbd=float(log10_lbd)), v)

    def lmm_reml_chunk_from_snp_f32(s, xcov, y_rot, low, high, snp_chunk, u_t, max_iter=50, tol=1e-2, threads=0, nullml=None,
                                    rotate_block_rows=256):
        out = O.lmm_reml_chunk_from_snp(s, xcov, y_rot, low, high, snp_chunk, u_t, max_iter, tol, nullml=nullml)
        return _rec("lmm_reml_chunk_from_snp_f32",
                    dict(low=float(low), high=float(high), max_iter=int(max_iter), tol=float(tol),
                         nullml=None if nullml is None else float(nullml), rotate_block_rows=int(rotate_block_rows),
                         snp_chunk=np.array(snp_chunk), u_t=np.array(u_t)), np.array(out))

    def lmm_reml_lmm2_chunk_from_snp_f32(s, xcov, y_rot, low, high, snp_chunk, u_t, nullml, max_iter=50, tol=1e-2, threads=0,
                                         rotate_block_rows=256):
        grot = O.rotate_block_f32(np.asarray(snp_chunk, dtype=np.float32), np.asarray(u_t, dtype=np.float32))
        out = O.lmm2_scan_rotated_block(grot, np.asarray(s), np.asarray(xcov), np.asarray(y_rot), low, high, max_iter, tol, nullml)
        return _rec("lmm_reml_lmm2_chunk_from_snp_f32",
                    dict(low=float(low), high=float(high), max_iter=int(max_iter), tol=float(tol), nullml=float(nullml)),
                    np.array(out))

    def lmm_assoc_chunk_from_snp_f32(s, xcov, y_rot, log10_lbd, snp_chunk, u_t, threads=0, nullml=None, rotate_block_rows=512):
        grot = O.rotate_block_f32(np.asarray(snp_chunk, dtype=np.float32), np.asarray(u_t, dtype=np.float32))
        out = O.lmm_assoc_fixed_lambda_block(grot, np.asarray(s), np.asarray(xcov), np.asarray(y_rot), float(log10_lbd), nullml)
        return _rec("lmm_assoc_chunk_from_snp_f32", dict(log10_lbd=float(log10_lbd), nullml=None if nullml is None else float(nullml)),
                    np.array(out))

    def lmm_assoc_chunk_f32(s, xcov, y_rot, log10_lbd, g_rot_chunk, threads=0, nullml=None):
        out = O.lmm_assoc_fixed_lambda_block(np.asarray(g_rot_chunk, dtype=np.float32), np.asarray(s), np.asarray(xcov),
                                             np.asarray(y_rot), float(log10_lbd), nullml)
        return _rec("lmm_assoc_chunk_f32", dict(log10_lbd=float(log10_lbd)), np.array(out))

    def fvlmm_assoc_prepare_cache_f32(s, xcov, y_rot, log10_lbd):
        fv = O.fvlmm_prepare_cache(np.asarray(s), np.asarray(xcov), np.asarray(y_rot), 10.0 ** float(log10_lbd))
        c = _Cache(fv, np.array(s), np.array(xcov), np.array(y_rot), log10_lbd)
        _rec("fvlmm_assoc_prepare_cache_f32", dict(log10_lbd=float(log10_lbd)), None)
        return c

    def fvlmm_assoc_chunk_with_cache_f32(cache, g_rot_chunk, threads=0, nullml=None):
        out = O.fvlmm_assoc_rotated_block(np.asarray(g_rot_chunk, dtype=np.float32), cache.fv, nullml)
        return _rec("fvlmm_assoc_chunk_with_cache_f32", dict(g_rot=np.array(g_rot_chunk), log10_lbd=cache.log10_lbd), np.array(out))

    def fvlmm_assoc_chunk_from_snp_with_cache_f32(cache, snp_chunk, u_t, threads=0, nullml=None, rotate_block_rows=512):
        out = O.fvlmm_assoc_chunk_from_snp(cache.s, cache.xcov, cache.y, cache.log10_lbd, snp_chunk, u_t, nullml)
        return _rec("fvlmm_assoc_chunk_from_snp_with_cache_f32",
                    dict(log10_lbd=cache.log10_lbd, rotate_block_rows=int(rotate_block_rows)), np.array(out))

    def fvlmm_assoc_chunk_f32(s, xcov, y_rot, log10_lbd, g_rot_chunk, threads=0, nullml=None):
        return fvlmm_assoc_chunk_with_cache_f32(fvlmm_assoc_prepare_cache_f32(s, xcov, y_rot, log10_lbd), g_rot_chunk, threads, nullml)

    def fvlmm_assoc_chunk_from_snp_f32(s, xcov, y_rot, log10_lbd, snp_chunk, u_t, threads=0, nullml=None, rotate_block_rows=512):
        return fvlmm_assoc_chunk_from_snp_with_cache_f32(fvlmm_assoc_prepare_cache_f32(s, xcov, y_rot, log10_lbd), snp_chunk, u_t,
                                                         threads, nullml, rotate_block_rows)

    for fn in (rust_eigh_from_array_f64_inplace, lmm_rotate_x_y_with_ut_f64, lmm_reml_null_f32, ml_loglike_null_f32,
               lmm_reml_chunk_from_snp_f32, lmm_reml_lmm2_chunk_from_snp_f32, lmm_assoc_chunk_from_snp_f32, lmm_assoc_chunk_f32,
               fvlmm_assoc_prepare_cache_f32, fvlmm_assoc_chunk_with_cache_f32, fvlmm_assoc_chunk_from_snp_with_cache_f32,
               fvlmm_assoc_chunk_f32, fvlmm_assoc_chunk_from_snp_f32):
        setattr(m, fn.__name__, fn)
    m.rust_eigh_from_array_f64 = rust_eigh_from_array_f64_inplace

    def __getattr__(name):                      # every other native name resolves to None (the layer probes them with try / except)
        if name.startswith("__"):
            raise AttributeError(name)
        return None
    m.__getattr__ = __getattr__
    return m


def main():
    if not os.path.isdir(REF_PY):
        raise SystemExit("the reference's Python layer is only present in the build container")
    sys.path.insert(0, REF_PY)
    sys.modules["janusx.janusx"] = build_stub()
    ref = importlib.import_module("janusx.pyBLUP.assoc")

    d = np.load(os.path.join(HERE, "panel_small.npz"))
    n = int(d["n"])
    packed, y, x = d["packed"], d["y"], d["x"]
    keep, af = d["keep"], d["af"]
    k1 = d["k_stream_m1"]
    rows = np.nonzero(keep)[0]
    flip = np.zeros(len(af), dtype=bool)
    snp = O.decode_centered_block_f32(packed, n, flip, af, rows=rows)          # what the CLI hands `gwas`: centred f32 rows
    x_extra = x[:, 1:]                                                            # the layer adds the intercept itself
    out = {"n": n, "y": y, "x_extra": x_extra, "k": k1, "snp": snp}

    def model_attrs(mod, tag):
        b = mod.bounds
        out.update({f"{tag}_lbd_null": mod.lbd_null, f"{tag}_ML0": mod.ML0, f"{tag}_LL0": mod.LL0,
                    f"{tag}_sigma_g2": mod.sigma_g2_null, f"{tag}_sigma_e2": mod.sigma_e2_null, f"{tag}_pve": mod.pve,
                    f"{tag}_pve_vc_ratio_raw": mod.pve_vc_ratio_raw, f"{tag}_trace_mean": mod.trace_mean,
                    f"{tag}_bounds": np.array([float(b[0]), float(b[1])]), f"{tag}_S": np.array(mod.S), f"{tag}_Dh": np.array(mod.Dh),
                    f"{tag}_Xcov": np.array(mod.Xcov), f"{tag}_yrot": np.array(mod.y).ravel(), f"{tag}_rank": mod.rank,
                    f"{tag}_lowrank": bool(mod.lowrank)})

    # ---- 1. LMM(y, X, K): the constructor's own ridge + eigendecomposition call + spectral initialisation
    CALLS.clear()
    lmm = ref.LMM(y, x_extra, k1)
    model_attrs(lmm, "lmm")
    names = [c[0] for c in CALLS]
    assert names[:3] == ["rust_eigh_from_array_f64_inplace", "lmm_rotate_x_y_with_ut_f64", "lmm_reml_null_f32"], names
    eig_a = CALLS[0][1]["a"]
    out["eigh_input"] = eig_a                                                     # K (f64) + 1e-6 I as the layer built it
    out["eigh_w"], out["eigh_v"] = CALLS[0][2]
    out["rot_x_in"] = CALLS[1][1]["x"]                                           # the design [1, X] the layer built
    nr = CALLS[2][1]
    out["null_args"] = np.array([nr["low"], nr["high"], nr["max_iter"], nr["tol"]])
    out["null_ret"] = np.array(CALLS[2][2])
    # ---- 2. LMM.gwas: bounds / iterations / tolerance the layer passes to the exact scan
    CALLS.clear()
    t_lmm = lmm.gwas(snp, threads=1)
    c = CALLS[-1]
    assert c[0] == "lmm_reml_chunk_from_snp_f32"
    out["lmm_gwas_args"] = np.array([c[1]["low"], c[1]["high"], c[1]["max_iter"], c[1]["tol"], c[1]["rotate_block_rows"]])
    out["lmm_gwas_nullml_is_none"] = c[1]["nullml"] is None
    out["lmm_gwas"] = np.array(t_lmm)
    # ---- 3. from_spectral on the same spectrum gives the same model
    lm2 = ref.LMM.from_spectral(y, x_extra, out["eigh_w"], out["eigh_v"])
    assert lm2.lbd_null == lmm.lbd_null and lm2.bounds == lmm.bounds
    # ---- 4. FvLMM: cache handling + both scan entry points
    CALLS.clear()
    fv = ref.FvLMM.from_lmm(lmm)
    t_fv = fv.gwas(snp, threads=1)
    t_fv2 = fv.gwas(snp[:7], threads=1)                                           # second chunk: the cache must be reused
    names = [c[0] for c in CALLS]
    assert names.count("fvlmm_assoc_prepare_cache_f32") == 1, names
    grot = O.rotate_block_f32(snp, lmm.Dh)
    t_fvr = fv.gwas_rotated(grot, threads=1)
    out["fvlmm_log10_lbd"] = float([c for c in CALLS if c[0] == "fvlmm_assoc_prepare_cache_f32"][0][1]["log10_lbd"])
    out["fvlmm_gwas"], out["fvlmm_gwas_rotated"], out["grot"] = np.array(t_fv), np.array(t_fvr), grot
    assert np.allclose(np.array(t_fv2), np.array(t_fv)[:7], rtol=1e-5, atol=0, equal_nan=True)     # f32 GEMM blocking differs with the chunk height
    # ---- 5. FastLMM: fixed-lambda scan inside 0.05 <= pve <= 0.95, exact scan outside
    CALLS.clear()
    fl = ref.FastLMM.from_lmm(lmm)
    t_fl = fl.gwas(snp, threads=1)
    out["fastlmm_route"] = CALLS[-1][0]
    out["fastlmm_gwas"] = np.array(t_fl)
    # ---- 6. LMM2: null ML optimum by the layer's scipy bounded search over the native ML likelihood, then the scan
    CALLS.clear()
    l2 = ref.LMM2.from_lmm(lmm)
    t_l2 = l2.gwas(snp, threads=1)
    out["lmm2_lbd_null_ml"], out["lmm2_ml0_exact"] = float(l2._lmm2_lbd_null_ml), float(l2._lmm2_ml0_exact)
    out["lmm2_ml_evals"] = np.array([c[1]["log10_lbd"] for c in CALLS if c[0] == "ml_loglike_null_f32"])
    out["lmm2_ml_values"] = np.array([c[2] for c in CALLS if c[0] == "ml_loglike_null_f32"])
    c = CALLS[-1]
    assert c[0] == "lmm_reml_lmm2_chunk_from_snp_f32"
    out["lmm2_gwas_args"] = np.array([c[1]["low"], c[1]["high"], c[1]["max_iter"], c[1]["tol"], c[1]["nullml"]])
    out["lmm2_gwas"] = np.array(t_l2)
    # ---- 7. a trait whose PVE leaves (0.05, 0.95): the scan bounds fall back to (-5, 5)
    rng = np.random.default_rng(3)
    y_noise = rng.normal(size=n)
    lm_n = ref.LMM.from_spectral(y_noise, x_extra, out["eigh_w"], out["eigh_v"])
    out["y_noise"] = y_noise
    model_attrs(lm_n, "noise")
    y_gen = 5.0 * (out["eigh_v"] @ (np.sqrt(np.maximum(out["eigh_w"], 0)) * rng.normal(size=n))) + 1e-4 * rng.normal(size=n)
    y_gen = y_gen + 40.0 * out["eigh_v"][:, -1] * math.sqrt(out["eigh_w"][-1])
    lm_g = ref.LMM.from_spectral(y_gen, x_extra, out["eigh_w"], out["eigh_v"])
    out["y_gen"] = y_gen
    model_attrs(lm_g, "gen")
    # ---- 8. the in-memory API (janusx/assoc/api.py::ASSOC :518) on the same inputs: lmm and fvlmm routes
    api_ok = []
    try:
        api = importlib.import_module("janusx.assoc.api")
        for model in ("lmm", "fvlmm"):
            CALLS.clear()
            a = api.ASSOC(model)
            a.fit(y, x_extra, k1)
            res = a.assoc(snp.T)                                                  # sample-major (n, m)
            tab = res[["beta", "se", "pwald"]].to_numpy() if hasattr(res, "columns") and {"beta", "se", "pwald"} <= set(res.columns) else None
            out[f"api_{model}_columns"] = np.array(";".join(map(str, getattr(res, "columns", []))))
            if tab is not None:
                out[f"api_{model}_table"] = np.array(tab, dtype=np.float64)
            out[f"api_{model}_route"] = np.array(str(a.route_))
            out[f"api_{model}_native_calls"] = np.array(";".join(c[0] for c in CALLS))
            api_ok.append(model)
    except Exception as e:   # noqa: BLE001 - the API layer needs more of the package than the model layer; recorded, not fatal
        out["api_error"] = np.array(repr(e))
    out["api_models"] = np.array(";".join(api_ok))
    np.savez_compressed(os.path.join(HERE, "reference_model.npz"), **out)
    print("wrote reference_model.npz", os.path.getsize(os.path.join(HERE, "reference_model.npz")), "bytes")
    print(json.dumps({k: (float(v) if np.ndim(v) == 0 and np.asarray(v).dtype.kind == "f" else None) for k, v in out.items()
                      if np.ndim(v) == 0 and np.asarray(v).dtype.kind == "f"}, indent=0))
    print("api:", api_ok, out.get("api_error"))


if __name__ == "__main__":
    main()
