#!/usr/bin/env python3
"""bench.py -- SNPs/sec of the full `-lmm` pipeline (GRM + eigh + null REML + per-SNP scan) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched through
torch.distributed.run (one rank per GPU, RCCL).  One "step" = one pass of the whole hot path over one synthetic
panel that is already resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Workload at N = 1: BASELINE.json configs[2], the largest single-GPU configuration of the metric (n = 20 000 samples,
m = 200 000 SNPs, HWE genotypes with MAF ~ U(0.02, 0.45), intercept-only, 100 causal SNPs, pve 0.5), run with the
exact per-SNP REML scan (`-lmm`, what the metric names; `--mode fvlmm` times the fixed-lambda scan instead;
`--n 5000 --m 50000` is configs[1]).
N > 1: SNP-sharded, default `--scaling strong` on BASELINE.json configs[3] (n = 50 000, m = 500 000: the panel the
north star's 1 -> 8 GPU target is quoted on): every rank builds the GRM partial of its contiguous SNP range, the f64
partials are summed with an RCCL all-reduce over xGMI, every rank then holds K, runs the eigendecomposition (its
O(n^3) stages dealt over the ranks, see janusx_amd/csrc/eigh.cpp) + null fit and scans its own SNP range.
`--scaling weak` gives every rank m SNPs of an n x (m N) panel instead.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
# form of the Q2 back-transformation the last decomposition took (jxg_last_kernel_ms(17), csrc/k_sbback.hip)
Q2_FORMS = ("sbback_apply_reg_kernel", "sbback_apply_solo_kernel", "sbback_apply_pair_kernel", "sbback_apply_bal_kernel")
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (spec, no sparsity)
MFMA_I8_PEAK_TOPS = 5000.0     # dense int8 MFMA: 2 x the bf16 rate per clock (MI355X_MICROARCH.md, Matrix cores: I8 row)
F64_VALU_PEAK_TFLOPS = 78.6    # f64 vector peak (public MI355X spec; 256 CUs x 128 flop/clk x 2.4 GHz)


def synth_panel_gpu(n, m, seed, device, m_offset=0, missing_rate=0.0):
    """HWE panel generated directly in HBM: MAF ~ U(0.02,0.45), g ~ Binomial(2,p)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed + 7919 * m_offset)
    bps = (n + 3) // 4
    out = torch.empty((m, bps), dtype=torch.uint8, device=device)
    dos_chunks = []
    chunk = max(1, min(m, (1 << 28) // max(n, 1)))
    for r0 in range(0, m, chunk):
        r1 = min(m, r0 + chunk)
        p = 0.02 + 0.43 * torch.rand((r1 - r0, 1), generator=g, device=device)
        d = (torch.rand((r1 - r0, n), generator=g, device=device) < p).to(torch.uint8)
        d += (torch.rand((r1 - r0, n), generator=g, device=device) < p).to(torch.uint8)
        codes = torch.where(d == 0, 0, d + 1).to(torch.uint8)  # 0->00, 1->10, 2->11
        if missing_rate > 0:
            mask = torch.rand((r1 - r0, n), generator=g, device=device) < missing_rate
            codes = torch.where(mask, torch.ones_like(codes), codes)
            d = torch.where(mask, torch.zeros_like(d), d)
        pad = bps * 4 - n
        if pad:
            codes = torch.nn.functional.pad(codes, (0, pad))
        c4 = codes.view(r1 - r0, bps, 4)
        out[r0:r1] = c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6)
        dos_chunks.append(d[: min(r1 - r0, 128)].clone() if r0 == 0 else None)
    return out, dos_chunks[0]


def family_panel_gpu(n, m, fam, seed, device):
    """Panel with family structure generated directly in HBM (SURVEY.md 8d "family" variant, python/janusx/script/sim.py:18-19):
    sibships of `fam` consecutive samples, every member after the founder copies each SNP of the founder with probability
    1/2 (kinship ~ 0.25-0.5), so a thresholded GRM is block diagonal by family.  No (m x n) array ever exists on the host.
    -> (packed (m, ceil(n / 4)) uint8 on the device, dosages of the first 128 SNPs)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    bps = (n + 3) // 4
    out = torch.empty((m, bps), dtype=torch.uint8, device=device)
    founder = (torch.arange(n, device=device) // fam) * fam
    head = None
    chunk = max(1, min(m, (1 << 27) // max(n, 1)))
    for r0 in range(0, m, chunk):
        r1 = min(m, r0 + chunk)
        p = 0.05 + 0.4 * torch.rand((r1 - r0, 1), generator=g, device=device)
        d = (torch.rand((r1 - r0, n), generator=g, device=device) < p).to(torch.uint8)
        d += (torch.rand((r1 - r0, n), generator=g, device=device) < p).to(torch.uint8)
        share = torch.rand((r1 - r0, n), generator=g, device=device) < 0.5
        d = torch.where(share, d[:, founder], d)
        codes = torch.where(d == 0, 0, d + 1).to(torch.uint8)
        pad = bps * 4 - n
        if pad:
            codes = torch.nn.functional.pad(codes, (0, pad))
        c4 = codes.view(r1 - r0, bps, 4)
        out[r0:r1] = c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6)
        if head is None:
            head = d[:128].clone()
    return out, head


def make_phenotype(dos_head, n, seed, device):
    """y = Z beta + e with up to 100 causal SNPs (the first rows of the panel), pve 0.5."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    z = dos_head[:100].to(torch.float64)
    z = z - z.mean(dim=1, keepdim=True)
    beta = torch.randn(z.shape[0], generator=g, device=device, dtype=torch.float64)
    gv = beta @ z
    vg = float(gv.var())
    e = torch.randn(n, generator=g, device=device, dtype=torch.float64) * math.sqrt(max(vg, 1e-12))
    return (gv + e).cpu().numpy()


def cpu_baseline(packed_cpu, n, y, mode, sample_m, threads, eig_cap=5000, scale_to_n=None, m_full=None):
    """The oracle (C + numpy/OpenBLAS restatement of the reference algorithm) timed on the host cores on a
    bounded SNP sample; GRM and scan are linear in m and are extrapolated; the null fit is timed at full n; dsyevd is timed at
    full n up to 2 * eig_cap rows and extrapolated above that with an exponent fitted from two timed sizes (see below).
    `scale_to_n` (largest config): the panel handed in holds the first n samples of a larger
    configuration; every stage is then scaled to `scale_to_n` samples by its own exponent (SSYRK and the rotation SGEMM n^2,
    decode and the per-SNP Brent evaluations n, eigh n^3) and the rule is printed with the number."""
    from oracle import jx_oracle as O
    from oracle import jx_oracle_c as OC
    from scipy.linalg import blas as sblas
    m_full = int(packed_cpu.shape[0] if m_full is None else m_full)
    ms = min(sample_m, packed_cpu.shape[0])
    sub = np.ascontiguousarray(packed_cpu[:ms])
    n_to = float(n if scale_to_n is None else scale_to_n)
    g1, g2, g3 = n_to / n, (n_to / n) ** 2, (n_to / n) ** 3
    t0 = time.perf_counter()
    mi, he, ho = OC.row_counts(sub, n)
    cnt = np.stack([mi, he, ho], 1)
    from janusx_amd import stats as st  # host-side count logic only (no GPU)
    gkeep, mean_g, scale, flip, var = st.stream_grm_row_prepare(cnt, n, 1, 0.02, 0.05, 0.0)
    rows = np.nonzero(gkeep)[0]
    lut = st.grm_lut_from_mean_scale(mean_g[rows], scale[rows], flip[rows])
    z = OC.decode_rows_lut(sub, n, lut, rows, center=False)
    t_dec = time.perf_counter() - t0
    t0 = time.perf_counter()
    # f32 SSYRK into an f32 scratch (lower triangle: the n(n+1)k flops the reference issues, src/stats/grm.rs:1650-1662) + f64 merge
    acc = sblas.ssyrk(1.0, z.T, trans=0, lower=1).astype(np.float64)
    t_syrk = time.perf_counter() - t0
    t_grm = t_dec * g1 + t_syrk * g2
    k = acc / float(np.sum(var[rows]))  # GRM of the SNP sample: same size/spectrum class as the full one for timing eigh
    k = np.tril(k) + np.tril(k, -1).T
    # LAPACK dsyevd: above 2 * eig_cap rows the decomposition is TIMED on the leading eig_cap and 2 * eig_cap blocks of the
    # sample GRM, the exponent is FITTED from the two times (dsyevd on many cores runs well below its asymptotic rate at
    # these sizes, so the fit comes out under 3) and the time at the full size is extrapolated from the larger block with that
    # exponent (extrapolation with the rule stated; the full-size call would take minutes of host
    # time at n = 20 000).  The two timings are taken once per process (_EIG_FIT) and shared by the configs[2] and configs[3]
    # baselines.  The eigenpairs the scan below uses come from a cheap exact source: the GPU-side result is NOT used, the
    # scan sample only needs *a* spectral basis of full size, so the small block's eigenvectors are embedded into an
    # orthonormal n x n basis (identity on the remaining coordinates).
    n_eig = min(n, eig_cap)
    t0 = time.perf_counter()
    s_b, u_b = O.gwas_eigh_from_grm(np.ascontiguousarray(k[:n_eig, :n_eig]).astype(np.float32))
    t_eig_block = time.perf_counter() - t0
    n_big = min(n, 2 * eig_cap)
    if n_big > n_eig:
        key = (n_eig, n_big)
        if key not in _EIG_FIT:
            t0 = time.perf_counter()
            O.gwas_eigh_from_grm(np.ascontiguousarray(k[:n_big, :n_big]).astype(np.float32))
            _EIG_FIT[key] = (t_eig_block, time.perf_counter() - t0)
        t_small, t_big = _EIG_FIT[key]
        eig_exp = min(3.0, max(2.0, math.log(t_big / t_small) / math.log(n_big / float(n_eig))))
        t_eig = t_big * (n_to / float(n_big)) ** eig_exp
        eig_rule = (f"{t_small:.2f}s at {n_eig} and {t_big:.2f}s at {n_big} rows measured, fitted exponent {eig_exp:.2f}, "
                    f"extrapolated from {n_big} to {int(n_to)} rows -> {t_eig:.1f}s")
    else:
        eig_exp = 3.0
        t_eig = t_eig_block * (n_to / float(n_eig)) ** 3
        eig_rule = (f"{t_eig:.2f}s at full size" if scale_to_n is None else
                    f"{t_eig_block:.2f}s at {n_eig} rows, scaled by (n/{n_eig})^3 -> {t_eig:.1f}s")
    if n_eig < n:
        s = np.concatenate([s_b, np.diag(k)[n_eig:].astype(np.float64) + 1e-6])
        u = np.zeros((n, n), dtype=np.float64)
        u[:n_eig, :n_eig] = u_b
        u[n_eig:, n_eig:] = np.eye(n - n_eig)
    else:
        s, u = s_b, u_b
    t0 = time.perf_counter()
    nm = O.spectral_null_model(y, np.ones((n, 1)), s, u)
    t_null = (time.perf_counter() - t0) * g2
    t0 = time.perf_counter()
    keep, af, miss = st.gwas_scan_row_stats(cnt, n, 0.02, 0.05, 1.0)
    srows = np.nonzero(keep)[0]
    slut = st.scan_lut_from_counts(af[srows], np.zeros(len(srows), bool), cnt[srows], n)
    gd = OC.decode_rows_lut(sub, n, slut, srows, center=False)
    t_sdec = time.perf_counter() - t0
    t0 = time.perf_counter()
    grot = gd @ nm.Dh.T
    t_rot = time.perf_counter() - t0
    t0 = time.perf_counter()
    if mode == "lmm":
        OC.lmm_scan_rotated_block(grot, nm.S, nm.Xcov, nm.y, nm.bounds[0], nm.bounds[1], 30, 1e-2, threads=threads)
    else:
        c = O.fvlmm_prepare_cache(nm.S, nm.Xcov, nm.y, nm.lbd_null)
        OC.fvlmm_assoc_block(grot, c.w, grot @ c.py, grot @ c.wx, c.a_chol, c.ypy, c.df, threads=threads)
    t_assoc = time.perf_counter() - t0
    t_scan = t_sdec * g1 + t_rot * g2 + t_assoc * g1
    scale_m = m_full / float(ms)
    total = t_grm * scale_m + t_eig + t_null + t_scan * scale_m
    rule = ("" if scale_to_n is None else
            f"; timed on the first {n} of {int(n_to)} samples and scaled per stage: decode and per-SNP evaluations x{g1:.2f} (n), "
            f"SSYRK, rotation SGEMM and the null rotation x{g2:.2f} (n^2), eigh (n^3) as stated")
    return {
        "value": m_full / total,
        "unit": "SNPs/s",
        "cores": int(threads),
        "kind": "port",
        "eigh_exponent": eig_exp,
        "sample_short": (f"first {ms} of {m_full} SNPs at n={n}" + ("" if scale_to_n is None else f" (of {int(n_to)})") +
                         f", GRM and scan linear in m; dsyevd: {eig_rule}"),
        "stages_s": {"grm": t_grm * scale_m, "eigh": t_eig, "null": t_null, "rotate": (t_sdec * g1 + t_rot * g2) * scale_m,
                     "assoc": t_assoc * g1 * scale_m},
        "sample": (f"oracle (C restatement + OpenBLAS ssyrk / sgemm + scipy dsyevd) on the first {ms} of {m_full} SNPs at "
                   f"n={n}: grm {t_dec + t_syrk:.2f}s (decode {t_dec:.2f} + SSYRK {t_syrk:.2f}) and scan {t_sdec + t_rot + t_assoc:.2f}s "
                   f"(decode {t_sdec:.2f} + SGEMM {t_rot:.2f} + per-SNP {t_assoc:.2f}) scaled x{scale_m:.1f} (linear in m); eigh "
                   + eig_rule + f"; null {t_null:.2f}s" + rule),
    }


def grm_peak_tflops(i8_share):
    """Dense MFMA peak the GRM is priced against: the exact-integer SNPs run on the int8 pipes (5 POP/s), the rest on the f16
    pipes (2.5 PFLOP/s, three products per algorithmic product); a mixed panel is priced by the time-weighted (harmonic) mix."""
    s = min(max(float(i8_share), 0.0), 1.0)
    # JXGPU_GRM_FP4=1 (opt-in): the exact-integer SNPs of the 256-tile form run on the fp4 pipes, twice the int8 rate
    exact_peak = 2.0 * MFMA_I8_PEAK_TOPS if os.environ.get("JXGPU_GRM_FP4", "0") not in ("", "0") else MFMA_I8_PEAK_TOPS
    return 1.0 / (s / exact_peak + (1.0 - s) / MFMA_F16_PEAK_TFLOPS)


def baseline_config_label(n, m):
    """Which BASELINE.json config an (n, m) panel is, for `config.workload`."""
    return {(5000, 50000): "BASELINE configs[1] shape", (20000, 200000): "BASELINE configs[2] shape",
            (50000, 500000): "BASELINE configs[3] shape"}.get((int(n), int(m)), "not a BASELINE config shape")


_EIG_FIT = {}      # (n_small, n_big) -> (seconds, seconds) of the host dsyevd, timed once per process
_PMC_SHAPE = {"n": None, "m": None}   # shape of the running configuration (set in main)


def _pmc_files(kind):
    """Committed counter summaries profiles/<tag>_pmc_<kind>.json whose recorded shape is the running one, newest tag
    first.  Round-1 summaries carry no shape field: they were collected at n = 5000, m = 50000."""
    import glob
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_{kind}.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        shape = d.get("shape", {"n": 5000, "m": 50000})
        if int(shape.get("n", -1)) == _PMC_SHAPE["n"] and int(shape.get("m", -1)) == _PMC_SHAPE["m"]:
            out.append((os.path.relpath(path, ROOT), d))
    return out


def pmc_traffic_bytes(kernel_prefix, run="fetch", fetch_scale=2.0):
    """(HBM bytes per launch, source file) of a kernel from a committed PMC summary of THIS shape: rocprofv3 --pmc
    FETCH_SIZE and WRITE_SIZE, each in its own pass, KB units; reads x fetch_scale (2 = the gfx950 correction of
    MI355X_MICROARCH.md for 16-byte-per-lane streaming reads; other widths have to be calibrated on a known byte count,
    as that guide says) + writes as reported.  (None, None) when no summary of the running shape is committed."""
    for src, d in _pmc_files("hbm_traffic"):
        runs = d.get("runs", {})
        rd = runs.get(run)
        if not rd:
            continue
        best, best_name = None, None
        for name, rec in rd.items():   # templated kernels are listed as "void jx::name<...>"; take the variant that ran longest
            if kernel_prefix in name:
                b = fetch_scale * rec["mean_KB"] * 1024.0
                if best is None or b > best:
                    best, best_name = b, name
        if best is not None:
            wr = runs.get("write", {}).get(best_name) if run == "fetch" else None
            if wr:
                best += wr["mean_KB"] * 1024.0
            return best, src
    return None, None


def pmc_mfma_util(kernel_substr, kind="mfma"):
    """(MFMA-pipe utilisation, source file) of a kernel from a committed counter pass of THIS shape (kind "mfma": the clean
    panel; "mfma_missing1pct": the same shape with 1 % missing calls):
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs). (None, None) when absent."""
    for src, d in _pmc_files(kind):
        for name, rec in d.get("kernels", {}).items():
            if kernel_substr in name:
                try:
                    return rec["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (rec["GRBM_GUI_ACTIVE"]["mean"] / 8.0 * 1024.0), src
                except Exception:
                    pass
    return None, None

HEADLINE_MAX_BYTES = 6144   # the driver parses the LAST stdout line; r05's ~20 kB line was not parsed


def _num(v, digits=6):
    """A JSON-safe scalar: floats rounded to `digits` significant figures, non-finite -> None."""
    if isinstance(v, (bool, str)) or v is None:
        return v
    if isinstance(v, (int, np.integer)):
        return int(v)
    try:
        f = float(v)
    except Exception:
        return None
    if not math.isfinite(f):
        return None
    return float(f"{f:.{digits}g}")


def _pick(d, keys):
    return {k: _num(d.get(k)) for k in keys if isinstance(d, dict) and k in d}


def headline_record(res):
    """The ONE line the driver parses (< HEADLINE_MAX_BYTES, strict JSON): the contract fields, the dominant kernel's
    roofline, the CPU baseline and one short record per extra leg.  Notes, per-kernel prose and the secondary rooflines
    stay in the detail record (gpurun_out/bench_detail.json + the `BENCH_DETAIL ` stdout line before this one)."""
    out = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data")}
    for k in ("value", "ms_per_step"):
        out[k] = _num(out[k], 9)
    cfg = res.get("config", {})
    out["config"] = {k: cfg.get(k) for k in ("workload", "n", "m", "m_kept", "mode", "parallelism") if k in cfg}
    rf = res.get("roofline") or {}
    out["roofline"] = dict(_pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                                      "avg_launch_ms", "launches_per_decomposition", "mfma_util_pmc")))
    for name in ("roofline_grm", "roofline_rotate", "roofline_scan", "roofline_eigh_gemm"):
        r = res.get(name)
        if isinstance(r, dict):
            out[name] = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic_over_algorithmic", "mfma_util_pmc"))
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = (_pick(cb, ("value", "unit", "cores", "kind", "eigh_exponent")) if "error" not in cb
                               else {"error": str(cb["error"])[:160]})
        if "sample" in cb:
            out["cpu_baseline"]["sample"] = str(cb.get("sample_short") or cb["sample"])[:400]
        if "value" in cb and cb["value"]:
            out["gpu_over_cpu_baseline"] = _num(res["value"] / cb["value"])
    if isinstance(res.get("null"), dict):
        out["null"] = _pick(res["null"], ("lbd", "pve"))
    out["stages_ms_per_step"] = {k: _num(v, 5) for k, v in (res.get("stages_ms_per_step") or {}).items()}
    if "stages_ms_per_step_max_over_ranks" in res:
        out["stages_ms_per_step_max_over_ranks"] = {k: _num(v, 5) for k, v in res["stages_ms_per_step_max_over_ranks"].items()}
    legs = {}
    for short, key in (("c1", "extra_c1_mouse"), ("c2", "extra_c2_fvlmm"), ("c3_miss", "extra_c3_missing1pct"),
                       ("c3_cov5", "extra_c3_cov5"), ("c3_chain", "extra_c3_chain"), ("c4_1gpu", "extra_c4_1gpu"),
                       ("c5_splmm", "extra_c5_splmm"), ("c5_pcg", "extra_c5_pcg")):
        leg = res.get(key)
        if not isinstance(leg, dict):
            continue
        if "error" in leg:
            legs[short] = {"error": str(leg["error"])[:160]}
            continue
        rec = _pick(leg, ("value", "ms_per_step", "steps", "m_kept"))
        lr = leg.get("roofline")
        if isinstance(lr, dict):
            rec["kernel"] = str(lr.get("kernel", ""))[:48]
            rec.update(_pick(lr, ("frac", "traffic_over_algorithmic")))
        if isinstance(leg.get("cpu_baseline"), dict) and leg["cpu_baseline"].get("value"):
            rec["cpu_value"] = _num(leg["cpu_baseline"]["value"])
            rec["gpu_over_cpu"] = _num(leg["value"] / leg["cpu_baseline"]["value"])
        legs[short] = rec
    if legs:
        out["legs"] = legs
    do = res.get("dist_overhead")
    if isinstance(do, dict) and "error" not in do:
        out["dist_overhead_ms"] = {k: _num(v, 5) for k, v in (do.get("delta_ms") or {}).items() if v is not None}
    if isinstance(res.get("one_gpu_same_workload"), dict):
        out["one_gpu_same_workload"] = {k: (_num(v) if k != "source" else str(v)[:120]) for k, v in res["one_gpu_same_workload"].items()}
    out["detail"] = "gpurun_out/bench_detail.json"
    line = json.dumps(out, allow_nan=False)
    if len(line) >= HEADLINE_MAX_BYTES:     # never let an over-long line cost the round's record again: drop the optional parts
        for k in ("dist_overhead_ms", "roofline_eigh_gemm", "roofline_scan", "stages_ms_per_step_max_over_ranks", "legs"):
            out.pop(k, None)
            line = json.dumps(out, allow_nan=False)
            if len(line) < HEADLINE_MAX_BYTES:
                break
    if len(line) >= HEADLINE_MAX_BYTES:     # last resort: the stage table down to the primary stages
        st = out.get("stages_ms_per_step", {})
        out["stages_ms_per_step"] = {k: st[k] for k in ("prep", "grm", "eigh", "null", "scan") if k in st}
        line = json.dumps(out, allow_nan=False)
    return out, line


def one_gpu_same_workload(n, m, mode):
    """The committed ONE-GPU measurement of the shape a multi-rank run times (its default is BASELINE configs[3]; the N = 1 default is
    configs[2], so the N = 1 `value` is NOT the baseline of the N > 1 ones): the leg of the newest profiles/*_bench_default.json that
    ran this shape on one MI355X.  None when no such record is committed."""
    import glob
    if mode != "lmm":
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_default.json")), reverse=True):
        try:
            rec = json.loads(open(path).read().strip().splitlines()[-1])
        except Exception:
            continue
        cfg = rec.get("config", {})
        if (int(cfg.get("n", -1)), int(cfg.get("m", -1))) == (int(n), int(m)) and int(rec.get("n_gpus", 0)) == 1:
            return {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "source": os.path.relpath(path, ROOT) + " (headline)"}
        leg = (rec.get("legs") or {}).get("c4_1gpu")
        if leg and (int(n), int(m)) == (50000, 500000):
            return {"value": leg["value"], "ms_per_step": leg["ms_per_step"], "source": os.path.relpath(path, ROOT) + " legs.c4_1gpu"}
    return None


def emit(res):
    """Detail record to gpurun_out/bench_detail.json and to an EARLIER stdout line, then the headline as the last line."""
    detail = json.dumps(res, default=lambda o: None)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as fh:
            fh.write(detail + "\n")
    except OSError:
        pass
    print("BENCH_DETAIL " + detail, flush=True)
    _, line = headline_record(res)
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--samples", dest="n", type=int, default=None)    # long forms: torchrun's own parser
    ap.add_argument("--m", "--snps", dest="m", type=int, default=None)       # treats a bare --n / --m as its options
    ap.add_argument("--mode", default="lmm", choices=["lmm", "fvlmm"])
    ap.add_argument("--missing", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=20260609)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--leg", default=None, choices=["c1", "c2_fvlmm", "c3_chain", "c5_splmm", "c5_pcg"],
                    help="run ONE extra leg of the default run alone and print its record (profiling)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the two extra legs of the default one-GPU run (1 %% missing calls; BASELINE configs[3] on one GPU)")
    args = ap.parse_args()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env != args.gpus:
        # Started without a launcher (`python bench.py --gpus N`): start one rank per GPU as a CHILD process -- before anything
        # here has touched the GPU (torch is imported below), never an exec -- relay its output (rank 0 prints the JSON line)
        # and leave with its exit code.
        if os.environ.get("JXGPU_BENCH_CHILD"):
            raise SystemExit(f"bench.py --gpus {args.gpus}: the launcher started WORLD_SIZE={world_env} ranks")
        import socket
        import subprocess
        with socket.socket() as sk:                   # a free rendezvous port on the loopback interface
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, JXGPU_BENCH_CHILD="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(key, None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    # BASELINE.json configs: [2] (n = 20k, m = 200k) is the largest one-GPU configuration of the metric, [3] (n = 50k,
    # m = 500k) the panel the multi-GPU target is quoted on
    if args.n is None:
        args.n = 20000 if world_env == 1 else 50000
    if args.m is None:
        args.m = 200000 if world_env == 1 else 500000

    import torch
    import torch.distributed as dist
    from janusx_amd import dist as jd
    from janusx_amd import pipeline as pl
    from janusx_amd import stats as st
    from janusx_amd._lib import lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X (no CPU fallback)")
    # JXGPU_BENCH_BACKEND=gloo: functional check of the multi-rank path on a box with fewer GPUs than ranks (ranks
    # share devices, collectives go through gloo); the measured configuration is always nccl (= RCCL), one GPU per rank
    backend = os.environ.get("JXGPU_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # JXGPU_BENCH_FORCE_DIST=1 exercises the RCCL code path with a single rank (used to validate it on a 1-GPU box)
    distributed = world > 1 or bool(os.environ.get("JXGPU_BENCH_FORCE_DIST"))
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")   # only reached without a launcher (JXGPU_BENCH_FORCE_DIST)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def run_leg(n, m_arg, missing, steps, warmup, covariates=0, mode=None, chain=None):
        """One timed configuration: synthetic panel resident in HBM, `warmup` untimed + `steps` timed passes of the whole
        hot path.  -> dict(elapsed, kept_total, kern, stage, null, packed, y, x, eigh_sharded, m)."""
        # replicated eigendecomposition: from 16384 individuals on, the symv tiles of the tridiagonalisation are dealt over
        # the ranks and summed by one all-reduce per column (JXGPU_DIST_EIGH=0 keeps every rank on the whole matrix)
        mode = args.mode if mode is None else mode
        eigh_min_n = int(os.environ.get("JXGPU_DIST_EIGH_MIN_N", "16384"))
        eigh_ranks = world > 1 or (distributed and os.environ.get("JXGPU_DIST_EIGH_FORCE", "0") != "0")
        eigh_sharded = bool(eigh_ranks and os.environ.get("JXGPU_DIST_EIGH", "1") != "0" and n >= eigh_min_n and
                            pl.enable_distributed_eigh(eigh_min_n))
        m = m_arg * world if args.scaling == "weak" else m_arg   # panel width of the whole job
        # SNP shard of this rank (contiguous range)
        lo = (m * rank) // world
        hi = (m * (rank + 1)) // world
        packed, dos_head = synth_panel_gpu(n, hi - lo, args.seed, dev, m_offset=lo, missing_rate=missing)
        if rank == 0:
            y = make_phenotype(dos_head, n, args.seed, dev)
            y_t = torch.from_numpy(y).to(dev)
        else:
            y_t = torch.empty(n, dtype=torch.float64, device=dev)
        if distributed:
            dist.broadcast(y_t, 0)
        y = y_t.cpu().numpy()
        x = np.ones((n, 1))
        if covariates > 0:      # fixed-effect columns beside the intercept (principal components in a real run): standard normal
            x = np.concatenate([x, np.random.default_rng(args.seed + 17).normal(size=(n, covariates))], axis=1)

        kern = {"grm_ms": 0.0, "rot_ms": 0.0, "scan_ms": 0.0, "grm_flops": 0.0, "rot_flops": 0.0, "scan_bytes": 0.0,
                "launches": 0}
        stage = {}

        p32_keep = [None]           # the P32 image is re-tiled every step into the same allocation (the work is timed, the malloc is not)
        # Placement check (untimed, before the warm-up): in about one process out of four the 1 GB re-tiling kernel runs 25 x slower
        # than its bytes allow (24 ms instead of 1) and every kernel that streams the P32 image pays a little as well -- a property
        # of where the allocation landed, not of the work.  Up to three fresh allocations are tried,
        # the rejected ones held until the choice is made so that the allocator cannot hand the same memory back.
        placement = {"retile_ms": [], "retries": 0}
        if world == 1 and not distributed:
            def retile_ms(buf):
                pnl = pl.Panel(packed, n, p32_buffer=buf)          # first call also pays one-off set-up
                torch.cuda.synchronize()
                t_a = time.perf_counter()
                pnl = pl.Panel(packed, n, p32_buffer=pnl.p32)
                torch.cuda.synchronize()
                return (time.perf_counter() - t_a) * 1e3, pnl.p32
            limit_ms = 6.0 * max(packed.numel() / 1e9, 0.05)      # ~1 ms per GB when placed well
            rejected = []
            best = None
            for attempt in range(4):
                ms_a, buf_a = retile_ms(None)
                placement["retile_ms"].append(round(ms_a, 3))
                if best is None or ms_a < best[0]:
                    best = (ms_a, buf_a)
                if ms_a <= limit_ms:
                    break
                rejected.append(buf_a)
                placement["retries"] = attempt + 1
            p32_keep[0] = best[1]
            del rejected

        step_no = [0]
        diag_sleep_ms = float(os.environ.get("JXGPU_BENCH_STEP_SLEEP_MS", "0") or 0)   # diagnostic: idle before every ODD step

        def one_step(record):
            t = {}
            torch.cuda.synchronize()
            step_no[0] += 1
            if diag_sleep_ms > 0 and (step_no[0] & 1):
                time.sleep(diag_sleep_ms * 1e-3)
            t0 = time.perf_counter()
            panel = pl.Panel(packed, n, p32_buffer=p32_keep[0])
            p32_keep[0] = panel.p32
            t0a = time.perf_counter()      # detail record only: host time of the re-tiling call (its launch is asynchronous) ...
            torch.cuda.synchronize()
            t0b = time.perf_counter()      # ... and the re-tiling kernel's share of `prep`
            counts = panel.counts()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            # the product's own composition (pipeline.build_grm, rank-aware): this rank's SNP shard -> partial Z Z' -> ONE
            # sum-reduction of the f64 lower-triangle tiles + the two denominators over xGMI (RCCL) -> K on every rank
            k32, geff, _ = pl.build_grm(packed, n, 1, 0.02, 0.05, panel=panel, payload_sharded=True)
            grm_ms = lib().jxg_last_kernel_ms(0)
            kern["grm_i8_share"] = float(lib().jxg_last_kernel_ms(12))   # SNPs on the exact int8 path / all kept SNPs
            geff_local = int(panel.grm_rows_local)                        # kept SNPs of this rank's launch
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            s, ut64 = pl.eigh_from_grm(k32, 1e-6, f32_consumer=True)    # as pipeline.run_gwas does: U is kept as the f32 U^T only
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            symv_ms, symv_mb = lib().jxg_last_kernel_ms(2), lib().jxg_last_kernel_ms(3)
            two_stage = lib().jxg_last_kernel_ms(10) > 0.5        # which reduction jxg_eigh_f64 took (two-stage from n = 1500)
            q2_ms, q2_gflop = lib().jxg_last_kernel_ms(4), lib().jxg_last_kernel_ms(5)
            eig_st = [lib().jxg_last_kernel_ms(i) for i in (6, 7, 8, 9)]
            model = pl.SpectralModel(s, ut64, x, y)
            del ut64
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            keep, af, miss = st.gwas_scan_row_stats(counts, n, 0.02, 0.05, 1.0)
            rows = np.nonzero(keep)[0]
            lut = st.scan_lut_from_counts(af[rows], np.zeros(len(rows), dtype=bool), counts[rows], n)
            tm = pl.StageTimes()
            chain_off, init = None, None
            if chain is not None and mode == "lmm":
                # the reference CLI's default scan: warm-start chains over chunks of `chain[0]` rows of the file, cut into
                # `chain[1]` pieces, seeded with log10 lambda0 (workflow_model_stream.py:1436-1480; src/stats/lmm.rs:134-161)
                chain_off = st.warm_chain_offsets(st.warm_chain_blocks_bed(rows, panel.m, chain[0]), len(rows), chain[1])
                lo_b, hi_b = model.null.bounds
                init = min(max(math.log10(model.null.lbd), lo_b), hi_b) if model.null.lbd > 0 else None
            out = pl.scan_rows(panel, model, rows, lut, mode, max_iter=30, tol=1e-2, times=tm,
                               return_evals=(mode == "lmm"), chain_off=chain_off, init_log10_lbd=init)
            if mode == "lmm":
                out, evals = out
                n_evals = float(evals.sum().item())   # Brent objective evaluations over all SNPs of the step
            else:
                n_evals = 0.0
            out = pl.gather_results(out)              # result rows of every rank in BED order (12 MB at configs[3])
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            if record:
                for key, val in (("prep", t1 - t0), ("grm", t2 - t1), ("eigh", t3 - t2), ("null", t4 - t3),
                                 ("scan", t5 - t4), ("rotate_k", tm.t.get("rotate", 0.0)),
                                 ("assoc_k", tm.t.get("scan", 0.0))):
                    stage[key] = stage.get(key, 0.0) + val
                kern.setdefault("prep_ms_by_step", []).append([round((t0a - t0) * 1e3, 3), round((t0b - t0a) * 1e3, 3), round((t1 - t0b) * 1e3, 3)])   # [re-tile call on the host, wait for it, counts + copy]
                kern["symv_ms"] = kern.get("symv_ms", 0.0) + symv_ms
                kern["symv_mb"] = kern.get("symv_mb", 0.0) + symv_mb
                kern["two_stage"] = two_stage
                if two_stage:
                    kern["q2_ms"] = kern.get("q2_ms", 0.0) + q2_ms
                    kern["q2_gflop"] = kern.get("q2_gflop", 0.0) + q2_gflop
                    kern["q2_launches"] = int(round(lib().jxg_last_kernel_ms(16)))
                    kern["q2_form"] = int(round(lib().jxg_last_kernel_ms(17)))
                    for name, v in zip(("eigh_band_reduction", "eigh_bulge_chasing", "eigh_divide_conquer", "eigh_q1_backtransform"), eig_st):
                        stage[name] = stage.get(name, 0.0) + v * 1e-3
                    stage["eigh_q2_backtransform_kernel"] = stage.get("eigh_q2_backtransform_kernel", 0.0) + q2_ms * 1e-3
                kern["grm_ms"] += grm_ms
                kern["grm_flops"] += float(n) * (n + 1) * geff_local
                kern["rot_ms"] += tm.t.get("rotate", 0.0) * 1e3
                # exact rows present: int8 planes (three products per algorithmic product) priced against the int8 peak
                kern["rot_peak"] = MFMA_I8_PEAK_TOPS if float(lib().jxg_last_kernel_ms(13)) > 0.5 else MFMA_F16_PEAK_TFLOPS
                kern["rot_flops"] += 2.0 * len(rows) * float(n) * n
                kern["scan_ms"] += tm.t.get("scan", 0.0) * 1e3
                kern["scan_bytes"] += 4.0 * n * len(rows)
                dim = x.shape[1] + 1
                # (B + 1) n (3 dim (dim + 1) / 2 + 5 dim + 8) flops per SNP, B = Brent evaluations
                kern["scan_flops"] = kern.get("scan_flops", 0.0) + (n_evals + len(rows)) * n * (1.5 * dim * (dim + 1) + 5 * dim + 8)
                kern["scan_evals"] = kern.get("scan_evals", 0.0) + n_evals / max(1, len(rows))
                kern["launches"] += 1
            return len(rows), geff, model.null, out

        for _ in range(warmup):
            one_step(False)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        # host-side pauses of the interpreter's cyclic garbage collector inside the timed steps, by generation (detail record)
        import gc
        gc_log = {"ms": [0.0, 0.0, 0.0], "runs": [0, 0, 0], "t": 0.0}

        def gc_cb(phase, info):
            if phase == "start":
                gc_log["t"] = time.perf_counter()
            else:
                g = min(int(info.get("generation", 0)), 2)
                gc_log["ms"][g] += (time.perf_counter() - gc_log["t"]) * 1e3
                gc_log["runs"][g] += 1
        gc.callbacks.append(gc_cb)
        ms0 = torch.cuda.memory_stats(dev)
        t_start = time.perf_counter()
        kept = 0
        null = None
        for _ in range(steps):
            kept, geff, null, out = one_step(True)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t_start
        gc.callbacks.remove(gc_cb)
        placement["gc_ms_in_timed_steps"] = [round(v, 2) for v in gc_log["ms"]]
        placement["gc_runs_in_timed_steps"] = gc_log["runs"]
        ms1 = torch.cuda.memory_stats(dev)
        # driver-level allocations / releases of the tensor allocator inside the timed steps (a step that fits the cache makes none)
        placement["torch_device_alloc_free_in_timed_steps"] = [int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)),
                                                               int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0))]
        placement["torch_reserved_gib"] = round(ms1.get("reserved_bytes.all.current", 0) / 2**30, 2)
        el_t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        kept_t = torch.tensor([float(kept)], dtype=torch.float64, device=dev)
        if distributed:
            dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
            dist.all_reduce(kept_t)
        # per-stage wall time: rank 0's own and the maximum over the ranks (the slowest rank of every stage is what a scaling
        # curve has to be read against; same key set on every rank)
        stage_max = dict(stage)
        if distributed:
            keys = sorted(stage)
            st_t = torch.tensor([stage[k] for k in keys], dtype=torch.float64, device=dev)
            dist.all_reduce(st_t, op=dist.ReduceOp.MAX)
            stage_max = {k: float(v) for k, v in zip(keys, st_t.tolist())}
        return dict(elapsed=float(el_t[0]), kept_total=float(kept_t[0]), kern=kern, stage=stage, null=null, packed=packed, placement=placement,
                    y=y, x=x, eigh_sharded=eigh_sharded, m=m, stage_max=stage_max)

    def leg_summary(leg, n, steps):
        """Condensed record of an extra leg: whole-step rate, stage times and the two f16-MFMA kernels against the dense peak."""
        k = leg["kern"]
        grm_tf = k["grm_flops"] / max(k["grm_ms"], 1e-9) / 1e9
        rot_tf = k["rot_flops"] / max(k["rot_ms"], 1e-9) / 1e9
        L = max(1, k["launches"])
        grm_peak = grm_peak_tflops(k.get("grm_i8_share", 0.0))
        return {"value": leg["kept_total"] * steps / leg["elapsed"], "unit": "SNPs/s", "steps": steps,
                "ms_per_step": leg["elapsed"] / steps * 1e3, "m_kept": int(leg["kept_total"]),
                "stages_ms_per_step": {kk: v / steps * 1e3 for kk, v in leg["stage"].items()},
                "roofline": ({"bound": "mfma", "kernel": Q2_FORMS[int(k.get("q2_form", 0))],
                              "achieved": k["q2_gflop"] / max(k["q2_ms"], 1e-9), "peak": 78.6, "unit": "TFLOP/s",
                              "frac": k["q2_gflop"] / max(k["q2_ms"], 1e-9) / 78.6,
                              "avg_launch_ms": k["q2_ms"] / L / max(1, int(k.get("q2_launches", 1))),
                              "launches_per_decomposition": int(k.get("q2_launches", 1)),
                              "traffic": None,
                              "note": "Q2 back-transformation of this leg: algorithmic 2 n^3 f64 flops over the kernel's HIP-event "
                                      "duration, against the 78.6 TFLOP/s f64 MFMA peak"} if k.get("two_stage") else None),
                "roofline_grm": {"bound": "mfma", "achieved": grm_tf, "peak": grm_peak, "unit": "TFLOP/s",
                                 "frac": grm_tf / grm_peak, "int8_share": k.get("grm_i8_share", 0.0),
                                 "avg_launch_ms": k["grm_ms"] / L},
                "roofline_rotate": {"bound": "mfma", "achieved": rot_tf, "peak": k.get("rot_peak", MFMA_F16_PEAK_TFLOPS),
                                    "unit": "TFLOP/s", "frac": rot_tf / k.get("rot_peak", MFMA_F16_PEAK_TFLOPS),
                                    "ms_per_step": k["rot_ms"] / L}}

    def leg_c1_mouse(steps=3, warmup=1):
        """BASELINE configs[0]: the reference's own example panel (example/mouse_hs1940: n = 1940, 10 300 sites, trait test0 with
        1410 phenotyped samples; committed as 2-bit codes in tests/golden/mouse_hs1940.npz), `jx gwas -lmm`: GRM on all samples,
        eigendecomposition + null + exact scan on the phenotyped ones (pipeline.run_trait)."""
        d = np.load(os.path.join(ROOT, "tests", "golden", "mouse_hs1940.npz"))
        pk, n1 = np.ascontiguousarray(d["packed"]), len(d["ids"])
        ph = d["pheno"][:, 0]
        pos = {sid: i for i, sid in enumerate(d["pheno_ids"])}
        yfull = np.array([ph[pos[sid]] if sid in pos else np.nan for sid in d["ids"]])
        keep_idx = np.nonzero(np.isfinite(yfull))[0]
        y1 = yfull[keep_idx]
        pt = torch.from_numpy(pk).to(dev)
        x1 = np.ones((len(keep_idx), 1))
        kept = 0
        for it in range(warmup + steps):
            if it == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            k1, eff, _ = pl.build_grm(pt, n1, 1, 0.02, 0.05)
            r1 = pl.run_trait(pt, n1, k1, keep_idx, y1, x1, "lmm")
            kept = int(np.count_nonzero(r1.keep))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return {"value": kept * steps / el, "unit": "SNPs/s", "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3,
                "m_kept": kept, "grm_eff_snps": int(eff), "null": {"lbd": r1.null.lbd, "pve": r1.null.pve},
                "workload": f"example/mouse_hs1940 (BASELINE configs[0]; real panel, n={n1} of which {len(keep_idx)} phenotyped, "
                            f"m={pk.shape[0]}), -lmm, maf 0.02 geno 0.05, intercept only", "data": "reference example panel"}

    def legs_c5(want=("splmm", "pcg")):
        """BASELINE configs[4] on ONE GPU (n = 200 000 in sibships of four, m = 1 000 000; panel generated in HBM): one step of the
        `-splmm` routes (sparse GRM by row panels -> sparse REML null + exact scan; fastGWA null + GRAMMAR-gamma scan) and of the
        `-BLUP` PCG leg (`rrblup_pcg_bed` with 160 000 training samples, `he_pcg_bed`) through the reference's entry points."""
        import tempfile
        import shutil
        from janusx_amd import janusx as jxrs
        n5, m5, n_tr = 200000, 1000000, 160000
        out = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pk, dos = family_panel_gpu(n5, m5, 4, 11, dev)
        y5 = make_phenotype(dos, n5, 7, dev)
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        counts = jxrs.bed_row_counts(pk, n5)
        keep, _miss, maf, _std = st.packed_prep_row_stats(counts, n5, 0.02, 0.05, 0.0)
        if not bool(keep.all()):
            pk = pk[torch.from_numpy(np.nonzero(keep)[0]).to(dev)]
            counts, maf = counts[keep], maf[keep]
            m5 = int(pk.shape[0])
        flip = np.zeros(m5, dtype=bool)
        wl = f"synthetic family panel n={n5} m={m5} (BASELINE configs[4] shape; sibships of four) on ONE GPU"
        if "splmm" in want:
            td = tempfile.mkdtemp()
            try:
                jxrs.spectral_cache_clear()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                path, _nn, nnz = jxrs.spgrm_packed_to_jxgrm(pk, n5, flip, maf, os.path.join(td, "k"), None, 1, 0.05)
                t1 = time.perf_counter()
                res_e, l10, _null = jxrs.splmm_exact_scan_from_jxgrm(path, y5, pk, n5, maf, flip)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                jxrs.spectral_cache_clear()
                yc = y5 - y5.mean()
                vp = float(yc @ yc) / float(n5 - 1)
                nullf = jxrs.spreml_sparse_fastgwa_fixed_vp_brent_from_jxgrm(path, yc, vp, low=-5.0, high=5.0, grid_size=17,
                                                                             tol=1e-3, max_iter=20)
                t3 = time.perf_counter()
                ga = jxrs.splmm_assoc_pcg_bed("device", y5, float(nullf[0]), packed=pk, packed_n_samples=n5, maf=maf, row_flip=flip,
                                              sparse_jxgrm_path=path, rhat_markers=1000, scan_mode="approx")
                torch.cuda.synchronize()
                t4 = time.perf_counter()
                out["extra_c5_splmm"] = {
                    "workload": wl + ", jx grm -sparse (cut-off 0.05) + jx gwas -splmm (the reference's default: fastGWA null + "
                                     "GRAMMAR-gamma scan, 1000 sampled markers) and -splmm-exact (sparse REML null + exact scan)",
                    "steps": 1, "warmup": 0, "unit": "SNPs/s", "m_kept": int(m5), "nnz": int(nnz),
                    "value": m5 / ((t1 - t0) + (t4 - t2)), "ms_per_step": ((t1 - t0) + (t4 - t2)) * 1e3,
                    "value_exact_route": m5 / (t2 - t0), "ms_per_step_exact_route": (t2 - t0) * 1e3,
                    "stages_ms_per_step": {"panel_generation_untimed": t_gen * 1e3, "sparse_grm": (t1 - t0) * 1e3,
                                           "exact_null_and_scan": (t2 - t1) * 1e3, "approx_null": (t3 - t2) * 1e3,
                                           "approx_scan": (t4 - t3) * 1e3},
                    "log10_lambda_exact": float(l10), "lambda_fastgwa": float(nullf[0]), "gamma": float(ga[0]),
                    "peak_hbm_gib": torch.cuda.max_memory_allocated() / 2**30}
                del res_e, ga
            finally:
                shutil.rmtree(td, True)
                jxrs.spectral_cache_clear()
            torch.cuda.empty_cache()
        if "pcg" in want:
            rng = np.random.default_rng(5)
            perm = rng.permutation(n5)
            tr = np.sort(perm[:n_tr]).astype(np.int64)
            te = np.sort(perm[n_tr:]).astype(np.int64)
            ytr = np.ascontiguousarray(y5[tr])
            cnt3 = counts.astype(np.int64)
            nm = n5 - cnt3[:, 0]
            alt = cnt3[:, 1] + 2 * cnt3[:, 2]
            pfr = (alt.astype(np.float32) / (np.float32(2.0) * np.maximum(nm, 1).astype(np.float32))).astype(np.float32)
            maf_l = np.minimum(pfr, np.float32(1.0) - pfr).astype(np.float32)      # `load_bed_2bit_packed` (gfreader.rs:4460-4485)
            flip_l = (alt.astype(np.float64) / (2.0 * np.maximum(nm, 1))) > 0.5      # `bed_packed_row_flip_mask`
            torch.cuda.reset_peak_memory_stats()
            torch.cuda.synchronize()
            # as `jx gs -rrBLUP -rr-solver pcg` runs them: both calls inside one image scope (the second reuses the first's images)
            with jxrs.pcg_image_scope():
                t0 = time.perf_counter()
                o = jxrs.rrblup_pcg_bed("", tr, ytr, te, lambda_value=float(m5), tol=1e-6, max_iter=200, packed=pk,
                                        packed_n_samples=n5, maf=maf_l, row_flip=flip_l)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                loop_ms, iters, op_ms, setup_ms = (float(lib().jxg_last_kernel_ms(i)) for i in (18, 19, 20, 21))
                h = jxrs.he_pcg_bed("", tr, ytr, packed=pk, packed_n_samples=n5, maf=maf_l, row_flip=flip_l, trace_samples=32)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            he_ms, he_apps = float(lib().jxg_last_kernel_ms(22)), float(lib().jxg_last_kernel_ms(23))
            op_bytes = 2.0 * n_tr * float(m5) / 4.0            # both images of the training payload once per application
            iters = max(iters, 1.0)
            _PMC_SHAPE.update(n=n5, m=m5)
            # the int8 forms of round 6 (k_pcg_i8.hip); the table forms when JXGPU_PCG_I8=0
            i8_op = os.environ.get("JXGPU_PCG_I8", "1") != "0"
            tr_a, tr_a_src = pmc_traffic_bytes("jx::pi_dot_kernel" if i8_op else "jx::packed_dot_t32_kernel")
            tr_b, tr_b_src = pmc_traffic_bytes("jx::pi_tdot_kernel" if i8_op else "jx::packed_tdot_f32_kernel")
            _PMC_SHAPE.update(n=int(args.n), m=int(args.m))
            out["extra_c5_pcg"] = {
                "workload": wl + f", jx gs -rrBLUP -rr-solver pcg: rrblup_pcg_bed with {n_tr} training / {n5 - n_tr} test samples, "
                                 "lambda = m, tol 1e-6; then he_pcg_bed (32 Hutchinson probes)",
                "steps": 1, "warmup": 0, "unit": "SNPs/s", "value": m5 / (t1 - t0), "ms_per_step": (t1 - t0) * 1e3,
                "pcg_iterations": int(iters), "converged": bool(o[3]), "rel_res": float(o[5]),
                "stages_ms_per_step": {"setup_images_and_prepass": setup_ms, "iteration_loop": loop_ms,
                                       "operator_kernels": op_ms, "predictions_and_rest": (t1 - t0) * 1e3 - setup_ms - loop_ms,
                                       "he_pcg_bed": (t2 - t1) * 1e3, "he_operator_kernels": he_ms},
                "he": {"sigma_g2": h[0], "sigma_e2": h[1], "h2": h[2], "operator_applications": int(he_apps)},
                "roofline": {"bound": "hbm", "kernel": ("pi_dot_kernel (Z'p, sample-major image) + pi_tdot_kernel (Z (Z'p), SNP-major "
                                                        "image), int8 MFMA on bit planes" if i8_op else
                                                        "packed_dot_t32_kernel (Z'p, sample-major image) + packed_tdot_f32_kernel "
                                                        "(Z (Z'p), SNP-major image)") + ": one operator application",
                             "achieved": op_bytes * iters / max(op_ms, 1e-9) / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": op_bytes * iters / max(op_ms, 1e-9) / 1e6 / HBM_PEAK_GBS,
                             "avg_application_ms": op_ms / iters, "applications": int(iters),
                             "he_achieved": op_bytes * max(he_apps, 1.0) / max(he_ms, 1e-9) / 1e6,
                             "traffic": (tr_a + tr_b) if (tr_a and tr_b) else None, "traffic_source": tr_a_src or tr_b_src,
                             "traffic_over_algorithmic": ((tr_a + tr_b) / op_bytes) if (tr_a and tr_b) else None,
                             "note": "algorithmic bytes per operator application = 2 x n_train x m / 4 (the 2-bit payload of the "
                                     "training samples once per half of (Z_c Z_c' + lambda I) p); duration = HIP events around the "
                                     "two halves (vector quantisation + streaming kernel each) on the launch stream, summed over the "
                                     "iterations (jxg_last_kernel_ms 20); DESIGN.md 3.6; "
                                     "traffic = rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of the two kernels from the committed pass of "
                                     "this shape (null when none)"},
                "peak_hbm_gib": torch.cuda.max_memory_allocated() / 2**30}
        del pk
        torch.cuda.empty_cache()
        return out

    if args.leg:
        # one extra leg alone (profiling: `rocprofv3 ... -- python3 bench.py --leg c5_pcg`); prints that leg's record
        if args.leg == "c1":
            rec = {"extra_c1_mouse": leg_c1_mouse()}
        elif args.leg == "c3_chain":
            lg = run_leg(20000, 200000, 0.0, args.steps, args.warmup, chain=(10000, 1))
            rec = {"extra_c3_chain": {k: v for k, v in leg_summary(lg, 20000, args.steps).items() if k != "roofline"}}
        elif args.leg == "c2_fvlmm":
            lg = run_leg(5000, 50000, 0.0, args.steps, args.warmup, mode="fvlmm")
            rec = {"extra_c2_fvlmm": {k: v for k, v in leg_summary(lg, 5000, args.steps).items() if k != "roofline"}}
        else:
            rec = legs_c5((args.leg[3:],))
        print(json.dumps(rec), flush=True)
        return

    n = args.n
    main_leg = run_leg(n, args.m, args.missing, args.steps, args.warmup)
    elapsed, kept_total, kern, stage, null = (main_leg[k] for k in ("elapsed", "kept_total", "kern", "stage", "null"))
    packed, y, x, eigh_sharded, m = (main_leg[k] for k in ("packed", "y", "x", "eigh_sharded", "m"))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = kept_total * args.steps / elapsed
        info = np.zeros(4, dtype=np.int64)
        lib().jxg_device_info(info.ctypes.data)
        L = max(1, kern["launches"])
        grm_tflops = kern["grm_flops"] / max(kern["grm_ms"], 1e-9) / 1e9
        rot_tflops = kern["rot_flops"] / max(kern["rot_ms"], 1e-9) / 1e9
        scan_gbs = kern["scan_bytes"] / max(kern["scan_ms"], 1e-9) / 1e6
        symv_gbs = kern.get("symv_mb", 0.0) / max(kern.get("symv_ms", 0.0), 1e-9)
        _PMC_SHAPE.update(n=int(n), m=int(m))
        tr_symv, tr_symv_src = pmc_traffic_bytes("jx::sytrd_symv_kernel")
        # the exact scan's kernel is chosen by n and p (jxg_last_kernel_ms(11): 0 LDS-resident, 1 tiled, 2 operands from L2)
        scan_kernel = {1: "jx::lmm_scan_tiled_kernel", 3: "void jx::series_coef_kernel"}.get(int(lib().jxg_last_kernel_ms(11)),
                                                                                             "jx::lmm_scan_fast_kernel")
        tr_scan, tr_scan_src = (pmc_traffic_bytes(scan_kernel, "fetch") if args.mode == "lmm" else
                                pmc_traffic_bytes("jx::fvlmm_scan_kernel", "fetch_fv"))
        i8_share = kern.get("grm_i8_share", 0.0)
        grm_kernel = "grm_i8_kernel" if i8_share >= 0.5 else "grm_f16x2_kernel"
        if i8_share >= 0.5 and os.environ.get("JXGPU_GRM_FP4", "0") not in ("", "0") and n >= 9900:
            grm_kernel = "grm_fp4_kernel"
        grm_peak = grm_peak_tflops(i8_share)
        tr_grm, tr_grm_src = pmc_traffic_bytes("jx::" + grm_kernel)
        mu_grm, mu_grm_src = pmc_mfma_util(grm_kernel)
        rot_i8 = float(lib().jxg_last_kernel_ms(13)) > 0.5      # exact design rows rotated on the int8 planes (k_rotate_i8.hip)
        # "rotate_i8_": rotate_i8_dma_kernel (round 6, the default) or rotate_i8_kernel -- whichever the newest committed pass holds
        rot_kernel = "rotate_i8_" if rot_i8 else ("rotate256_kernel" if n >= 4096 else "rotate_f16x2_kernel")
        rot_peak = MFMA_I8_PEAK_TOPS if rot_i8 else MFMA_F16_PEAK_TFLOPS
        mu_rot, mu_rot_src = pmc_mfma_util(rot_kernel)
        tr_rot, tr_rot_src = pmc_traffic_bytes("jx::" + rot_kernel)
        q1_share = float(world) if eigh_sharded else 1.0      # a rank of a sharded decomposition back-transforms n / world columns
        scan_series = int(lib().jxg_last_kernel_ms(11)) == 3
        # what the association stage is priced on: the series form executes 2 (p + 2) n 64 flops per SNP on the f64 matrix pipes
        # (k_scan_fast.hip series_coef_kernel), the other forms the reference formulation's evaluations
        scan_flops_priced = (2.0 * (x.shape[1] + 2) * 64.0 * n * kern["scan_bytes"] / (4.0 * n)) if scan_series else kern.get("scan_flops", 0.0)
        oz_p = int(pl.LAST_EIGH.get("planes") or lib().jxg_oz_planes())   # digit planes the pipeline's eigendecomposition ran with
        F64_MFMA_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64: one 2048-flop block per 64 cycles per SIMD = the f64 vector rate
        if kern.get("two_stage"):
            # dominant kernel by time of the two-stage eigensolver path: the back-transformation of the bulge-chasing
            # reflectors (one launch per decomposition; profiles/r02*_kernel_stats.csv)
            q2_tflops = kern["q2_gflop"] / max(kern["q2_ms"], 1e-9)
            # calibrated on the kernel's known row bytes (n^3 / 8 = 1.00 TB at n = 20000 vs 1.01 - 1.02 TB raw FETCH_SIZE, with
            # 8-byte row reads in profiles/r03a and 16-byte ones in r03b) the counter is exact here, not halved
            # n >= 32768 (eight 16-column units per CU): one wave per unit, two sweep groups per pass (k_sbback.hip)
            q2_kernel = Q2_FORMS[int(kern.get("q2_form", 0))]
            q2_launches = max(1, int(kern.get("q2_launches", 1)))
            q2_pairs = kern.get("q2_form", 0) in (2, 3)             # two sweep groups per pass over the rows
            tr_q2, tr_q2_src = pmc_traffic_bytes("jx::" + q2_kernel, fetch_scale=1.0)
            mu_q2, mu_q2_src = pmc_mfma_util(q2_kernel)
            roofline_main = {"bound": "mfma", "kernel": q2_kernel, "achieved": q2_tflops,
                             "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": q2_tflops / F64_MFMA_PEAK_TFLOPS,
                             "traffic": tr_q2, "traffic_source": tr_q2_src, "avg_launch_ms": kern["q2_ms"] / L / q2_launches,
                             "launches_per_decomposition": q2_launches,
                             "traffic_over_algorithmic": (tr_q2 * q2_launches / (2.0 * float(n) ** 3 / (16.0 if q2_pairs else 8.0))) if tr_q2 else None,   # n^3 / 8 B each way (n^3 / 16 with two groups per pass)
                             "traffic_over_result": (tr_q2 / (16.0 * float(n) * n)) if tr_q2 else None,
                             "mfma_util_pmc": mu_q2, "mfma_util_source": mu_q2_src,
                             "note": "algorithmic flops = 4 n sum(reflector lengths) ~ 2 n^3 for C <- Q2 C (DESIGN.md 3.5); the "
                                     "kernel issues (64 + 32) / 64 of them on the parallelogram blocks (U = V T' is precomputed per block); "
                                     "duration = HIP start/stop events bound to the dispatch (hipExtLaunchKernelGGL) on the "
                                     "launch stream; peak = f64 MFMA 78.6 TFLOP/s (public MI355X figure, = 64 cycles per "
                                     "16x16x4 block per SIMD; rocBLAS dgemm reaches 75 here); traffic = rocprofv3 FETCH_SIZE "
                                     "(calibrated on the known row bytes of this kernel, n^3 / 8 per launch: the raw counter equals them with 8-byte and with 16-byte row reads alike, so no x2 here) + "
                                     "WRITE_SIZE, each in its own pass, per launch, from the committed "
                                     "summary of this shape (algorithmic: every row of C read and written once per group of "
                                     "32 sweeps = 8 n^3 / 32 B; once per PAIR of groups in sbback_apply_pair_kernel: 8 n^3 / 64 B); "
                                     "mfma_util_pmc = SQ_VALU_MFMA_BUSY_CYCLES share of SIMD cycles"}
        else:
            # one-stage path (n < 1500): the dominant kernel is the symv of the tridiagonalisation, one launch per column
            roofline_main = {"bound": "hbm", "kernel": "sytrd_symv_kernel",
                             "achieved": symv_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": symv_gbs / HBM_PEAK_GBS,
                             "traffic": tr_symv, "traffic_source": tr_symv_src,
                             "avg_launch_ms": kern.get("symv_ms", 0.0) / L,
                             "note": "algorithmic bytes = lower triangle of the trailing matrix (4 nt^2 + 4 nt B) per launch; "
                                     "achieved = mean bytes / mean duration of the mid-panel launch of every 64-column panel "
                                     "(HIP start/stop events bound to the dispatch, hipExtLaunchKernelGGL, on the launch "
                                     "stream); traffic = rocprofv3 FETCH_SIZE (own pass) x2 gfx950 correction"}
        res = {
            "metric": "SNPs/sec full -lmm (GRM+eig+scan)" if args.mode == "lmm" else "SNPs/sec full -fvlmm (GRM+eig+scan)",
            "value": value,
            "unit": "SNPs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64 (eigendecomposition, REML and per-SNP statistics; GRM and rotation products are exact int8 / split-f16 MFMA "
                     "sums merged in f64)",
            "dtype_detail": "eigendecomposition: reduction stages and Q2 on f64 MFMA; Q1 back-transformation and divide-and-conquer "
                            "merges as products of 5 int8 digit planes per operand (6 behind rust_eigh_from_array_f64), exact i32 sums, "
                            "f64 combination; GRM: int8 MFMA with i32 accumulation or fp16 hi+lo split with f32 accumulation, f64 merge; "
                            "rotation: exact design rows x three int8 planes of U, other rows fp16 hi+lo split",
            "data": "synthetic",
            "config": {"workload": f"synthetic HWE panel n={n} m={m} ({baseline_config_label(n, args.m)}{f' x{world} SNPs (weak scaling)' if (args.scaling == 'weak' and world > 1) else ''}), -{args.mode}, "
                                   f"maf 0.02 geno 0.05, intercept only, missing={args.missing}",
                       "n": n, "m": m, "m_kept": int(kept_total), "mode": args.mode,
                       "parallelism": f"snp-shard x{world}" + (
                           "" if not eigh_sharded else
                           ", eigh: band reduction with the trailing matrix sharded over the ranks (two collectives per panel), "
                           "replicated bulge chasing, divide and conquer with the top-level merge per rank window, "
                           "back-transformations sharded by eigenvector" if kern.get("two_stage") else
                           ", eigh symv tiles sharded over ranks (one-stage tridiagonalisation below the two-stage "
                           "threshold)")},
            "roofline": roofline_main,
            "roofline_grm": {"bound": "mfma", "kernel": grm_kernel + " (int8 exact-integer Gram term for SNPs without a missing "
                                                        "call among the selected samples: v_mfma_i32_32x32x32_i8; the fp16 "
                                                        "hi/lo three-product grm_f16x2_kernel for the rest)",
                             "achieved": grm_tflops, "peak": grm_peak, "unit": "TFLOP/s", "int8_share": i8_share,
                             "frac": grm_tflops / grm_peak,
                             "traffic": tr_grm, "traffic_source": tr_grm_src,
                             "traffic_over_algorithmic": (tr_grm / (n * float(kept_total) / 4.0 + 8.0 * n * (n + 1) / 2.0)) if tr_grm else None,
                             "traffic_note": "HBM read bytes per launch, rocprofv3 FETCH_SIZE (own pass) x2 gfx950 "
                                             "correction, from the committed summary of this shape (null when none); "
                                             f"algorithmic input = n*m/4 = {n * m / 4e6:.1f} MB (payload "
                                             "re-read per tile pair is served by L2/MALL)",
                             "note": "algorithmic n(n+1)m flops (one int8 multiply-add = 2 ops) over the duration of the call "
                                     "(classification, affine terms and MFMA kernels; HIP events); peak = dense int8 MFMA "
                                     "5 POP/s for the exact-integer SNPs (one product per algorithmic product), dense f16 "
                                     "2.5 PFLOP/s for the others (three products per algorithmic product), mixed by share",
                             "mfma_util_pmc": mu_grm, "mfma_util_source": mu_grm_src,
                             "mfma_util_note": "SQ_VALU_MFMA_BUSY_CYCLES share of SIMD cycles (own rocprofv3 --pmc pass, "
                                               "profiles/*_pmc_mfma.json): the fraction of time the matrix pipes are busy",
                             "avg_launch_ms": kern["grm_ms"] / L},
            "roofline_rotate": {"bound": "mfma", "kernel": ("rotate_i8_dma_kernel" if rot_i8 else rot_kernel), "achieved": rot_tflops,
                                "peak": rot_peak, "unit": "TFLOP/s", "frac": rot_tflops / rot_peak,
                                "mfma_util_pmc": mu_rot, "mfma_util_source": mu_rot_src,
                                "traffic": tr_rot, "traffic_source": tr_rot_src,
                                "traffic_over_algorithmic": (tr_rot / (min(32768.0, float(kept_total)) * (n / 4.0 + 4.0 * n) + 3.0 * float(n) * n))
                                if tr_rot else None,
                                "traffic_note": "HBM bytes per launch (one launch per block of <= 32768 SNP rows): FETCH_SIZE x2 + "
                                                "WRITE_SIZE; algorithmic = the block's payload + the U planes once + 4 n bytes "
                                                "written per row",
                                "note": "algorithmic 2 m n^2 flops; exact design rows (allele counts, no missing call): THREE "
                                        "int8 MFMA products per algorithmic product (U in three int8 planes, exact i32 sums, "
                                        "f64 combine) against the dense int8 peak 5 POP/s -- 0.75 of the matrix-pipe cycles of "
                                        "the two fp16 products they replace; other rows: three fp16 products (hi/lo split of "
                                        "both operands) against 2.5 PFLOP/s",
                                "ms_per_step": kern["rot_ms"] / L},
            "roofline_eigh_gemm": ({
                "bound": "mfma", "kernel": f"oz_mm_kernel<{oz_p}> (sliced f64 GEMM on v_mfma_i32_32x32x32_i8, csrc/k_ozgemm.hip)",
                "stage": "Q1 back-transformation C <- Q1 C (W = V'C and C -= (V T) W per block of 2048 reflectors)",
                "algorithmic_tflops": 2.0 * float(n) ** 3 / q1_share / max(stage.get("eigh_q1_backtransform", 0.0) / args.steps, 1e-9) / 1e12,
                "columns_per_rank": int(round(n / q1_share)),
                "planes": oz_p,
                "issued_int8_products_per_algorithmic_product": oz_p * (oz_p + 1) // 2,
                "achieved": 2.0 * float(n) ** 3 / q1_share * (oz_p * (oz_p + 1) // 2)
                / max(stage.get("eigh_q1_backtransform", 0.0) / args.steps, 1e-9) / 1e12,
                "peak": MFMA_I8_PEAK_TOPS, "unit": "TOP/s",
                "frac": 2.0 * float(n) ** 3 / q1_share * (oz_p * (oz_p + 1) // 2)
                / max(stage.get("eigh_q1_backtransform", 0.0) / args.steps, 1e-9) / 1e12 / MFMA_I8_PEAK_TOPS,
                "f64_mfma_peak_tflops": F64_MFMA_PEAK_TFLOPS,
                "note": "algorithmic 2 n^3 f64 flops of the stage over its whole duration (slicing of C and W, products, "
                        "HIP events around the stage, C-independent part -- V images, Gram, T^-1, V T -- included: it runs in line; on "
                        "several ranks a rank back-transforms its n / world eigenvector columns and is priced on those); every algorithmic product is issued as planes (planes + 1) / 2 "
                        "int8 digit products (exact i32 sums, f64 combination: 4e-14 relative at 6 planes = 21 products, 1e-11 at the 5 planes = 15 products of the f32-consuming pipeline), priced against the dense int8 "
                        "peak 5 POP/s; algorithmic_tflops is to be read against the 78.6 TFLOP/s f64 MFMA roof this stage "
                        "(rocBLAS dgemm, rounds 1 - 3: 72) no longer sits under; the divide and conquer's merges run on the same kernel"}
                                   if kern.get("two_stage") and n >= 3000 else None),
            "roofline_scan": ({"bound": "f64 mfma" if scan_series else "f64 valu",
                               "kernel": ("lmm_scan_fast_kernel (s / X~ / y~ resident in LDS)",
                                          "lmm_scan_tiled_kernel (LDS tiles of s / X~ / y~, 16 SNPs per workgroup in lock step)",
                                          "lmm_scan_fast_kernel (operands from L2)",
                                          "series_coef_kernel (per-SNP Chebyshev series of the SNP-specific sums, one pass over the "
                                          "rotated rows on v_mfma_f64_16x16x4_f64) + lmm_scan_fast_kernel<SERIES> (Brent on the series)"
                                          )[int(lib().jxg_last_kernel_ms(11))],
                               "achieved": scan_flops_priced / max(kern["scan_ms"], 1e-9) / 1e9,
                               "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": scan_flops_priced / max(kern["scan_ms"], 1e-9) / 1e9 / F64_VALU_PEAK_TFLOPS,
                               "priced_work": ("series_coef_kernel: 2 (p + 2) n 64 f64 MFMA flops per SNP (the (p + 2) rotated rows of "
                                               "a SNP against the n x 64 Chebyshev table), over the whole association time; the "
                                               "reference formulation's (B + 1) n (...) flops are NOT executed by this form"
                                               if scan_series else "SURVEY 8(d): (B + 1) n (3 dim (dim + 1) / 2 + 5 dim + 8) flops per SNP"),
                               "brent_evals_per_snp": kern.get("scan_evals", 0.0) / L,
                               "hbm_gbs": scan_gbs,
                               "traffic": tr_scan, "traffic_source": tr_scan_src,
                               "traffic_over_algorithmic": (tr_scan / (4.0 * n * min(32768.0, float(kept_total)))) if tr_scan else None,
                               "note": "Brent over the exact per-SNP REML: every objective evaluation is a pass over the n "
                                       "rotated samples (one f64 reciprocal per sample, operands s / X~ / y~ resident in "
                                       "LDS); algorithmic flops per SURVEY 8(d): (B + 1) n (3 dim (dim + 1) / 2 + 5 dim + 8) "
                                       "per SNP with the measured B; peak = 78.6 TFLOP/s f64 vector (public MI355X "
                                       "figure); the 4 n bytes per SNP are read once (hbm_gbs), HBM is not the bound",
                               "ms_per_step": kern["scan_ms"] / L} if args.mode == "lmm" else
                              ({"bound": "mfma", "kernel": "rotate_f16x2_kernel<fused epilogue> + fvlmm_finish_kernel",
                                "achieved": rot_tflops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": rot_tflops / MFMA_F16_PEAK_TFLOPS, "traffic": None,
                                "note": "the fixed-lambda scan has no kernel of its own: the rotation kernel reduces every "
                                        "128 x 128 tile of G~ against w / Py~ / WX~ in its epilogue and stores the p + 2 partial "
                                        "sums of every SNP in its column tile's slot (plain stores, no atomics), G~ is never "
                                        "written; the finish kernel adds the column tiles in index order ((p + 2) x 8 B per SNP "
                                        "and column tile read; bit-reproducible). "
                                        "Priced as the rotation (roofline_rotate); JXGPU_FVLMM_FUSED=0 restores the two-kernel "
                                        "form (4 n B per SNP written and read back, HBM-bound)",
                                "ms_per_step": (kern["rot_ms"] + kern["scan_ms"]) / L}
                               if pl._fused_fixed_lambda(int(x.shape[1])) else
                               {"bound": "hbm", "kernel": "fvlmm_scan_kernel",
                                "traffic": tr_scan, "traffic_source": tr_scan_src,
                                "achieved": scan_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": scan_gbs / HBM_PEAK_GBS,
                                "ms_per_step": kern["scan_ms"] / L})),
            "stages_ms_per_step": {k: v / args.steps * 1e3 for k, v in stage.items()},
            "prep_ms_by_step": kern.get("prep_ms_by_step"),
            "placement_check": main_leg.get("placement"),
            **({"stages_ms_per_step_max_over_ranks": {k: v / args.steps * 1e3 for k, v in main_leg["stage_max"].items()}}
               if distributed else {}),
            "null": {"lbd": null.lbd, "pve": null.pve},
            "device": {"cus": int(info[0]), "clock_khz": int(info[1]), "hbm_mib": int(info[2])},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                from oracle import jx_oracle_c as OC
                res["cpu_baseline"] = cpu_baseline(packed.cpu().numpy(), n, y, args.mode, args.cpu_sample,
                                                   os.cpu_count() or 1)
            except Exception as e:  # the baseline is a reported number, never the product path
                res["cpu_baseline"] = {"error": repr(e)}
        # Extra legs of the default one-GPU run (BASELINE configs[2], -lmm, no missing calls): (1) the same shape with 1 %
        # missing calls (every SNP takes the fp16 hi/lo three-product path of the GRM and of the rotation), (2) one step of
        # BASELINE configs[3] (n = 50 000, m = 500 000: the shape the north star's 8-GPU target is quoted on) on this ONE GPU.
        if (world == 1 and not distributed and not args.no_extra and args.mode == "lmm" and args.missing == 0.0
                and (int(n), int(args.m)) == (20000, 200000)):
            del main_leg, packed
            torch.cuda.empty_cache()
            try:
                leg = run_leg(20000, 200000, 0.01, 2, 1)
                sm = leg_summary(leg, 20000, 2)
                mu_g, mu_g_src = pmc_mfma_util("grm_i8_kernel", "mfma_missing1pct")
                mu_r, mu_r_src = pmc_mfma_util("rotate_i8_", "mfma_missing1pct")
                res["roofline_grm_missing1pct"] = dict(sm["roofline_grm"], kernel="grm_i8_kernel<LUT> x 2 (SNPs with missing calls: "
                                                       "the missing call's count in two int8 digits, two int8 Gram products with "
                                                       "per-SNP byte LUTs, exact diagonal; csrc/k_grm.hip dense missing-call path)",
                                                       mfma_util_pmc=mu_g, mfma_util_source=mu_g_src,
                                                       issued_int8_products_per_algorithmic_product=2,
                                                       note="2 timed steps of the configs[2] shape with 1 % missing calls; "
                                                            "algorithmic n(n+1)m flops over the whole accumulate call, priced "
                                                            "against the int8 peak; every SNP holds a missing call at this rate and "
                                                            "takes TWO int8 Gram products ((30 B B' + A A') / 31 with the byte LUTs "
                                                            "B = (0, P1, 56, 112), A = (0, P1 + P2, 56, 112): 2e-7 of the mean diagonal), "
                                                            "so 0.5 x the clean kernel's fraction is the ceiling of this form "
                                                            "(rounds 2 - 3: fp16 three-product kernel 229 ms, sparse correction 165 ms)")
                res["roofline_rotate_missing1pct"] = dict(sm["roofline_rotate"],
                                                          kernel="rotate_i8_dma_kernel<0> + <1> (every row has missing calls "
                                                                 "at this rate: the count operand and the indicator of the missing "
                                                                 "calls against the three int8 planes of U, exact; the fp16 hi/lo "
                                                                 "kernel until round 4)",
                                                          mfma_util_pmc=mu_r, mfma_util_source=mu_r_src,
                                                          issued_int8_products_per_algorithmic_product=6,
                                                          note="same leg; algorithmic 2 m n^2 flops")
                res["extra_c3_missing1pct"] = {k: sm[k] for k in ("value", "unit", "steps", "ms_per_step", "m_kept",
                                                                  "stages_ms_per_step")}
                del leg
                torch.cuda.empty_cache()
            except Exception as e:   # an extra leg never takes the headline line down
                res["extra_c3_missing1pct"] = {"error": repr(e)}
            try:
                # the configs[2] shape with five covariates beside the intercept (what a GWAS with principal components runs):
                # only the scan differs -- block form of the exact per-SNP evaluation
                leg = run_leg(20000, 200000, 0.0, 2, 1, covariates=5)
                sm = leg_summary(leg, 20000, 2)
                res["extra_c3_cov5"] = dict({k: sm[k] for k in ("value", "unit", "steps", "ms_per_step", "m_kept",
                                                                "stages_ms_per_step")},
                                            note="configs[2] shape, -lmm with 5 covariates beside the intercept (dim = 7)")
                del leg
                torch.cuda.empty_cache()
            except Exception as e:
                res["extra_c3_cov5"] = {"error": repr(e)}
            try:
                # the same shape with the reference's DEFAULT exact scan: warm-start chains (`jx gwas -lmm` without
                # JX_LMM_UNIFIED_NO_WARM_START): chunks of 10 000 rows of the file = 20 sequential chains, and cut into 64 pieces
                # each (what rayon's splitter does on a 32-thread pool) = 1280 chains
                res["extra_c3_chain"] = {}
                for tag, ch in (("chunk10000", (10000, 1)), ("chunk10000_pieces64", (10000, 64))):
                    leg = run_leg(20000, 200000, 0.0, 2, 1, chain=ch)
                    sm = leg_summary(leg, 20000, 2)
                    res["extra_c3_chain"][tag] = {k: sm[k] for k in ("value", "unit", "steps", "ms_per_step", "m_kept")}
                    res["extra_c3_chain"][tag]["assoc_k_ms"] = sm["stages_ms_per_step"].get("assoc_k")
                    res["extra_c3_chain"][tag]["brent_evals_per_snp"] = leg["kern"].get("scan_evals", 0.0) / max(1, leg["kern"]["launches"])
                    del leg
                    torch.cuda.empty_cache()
                res["extra_c3_chain"].update(res["extra_c3_chain"]["chunk10000"])
                res["extra_c3_chain"]["note"] = ("configs[2] shape, -lmm with the reference's default warm-start chain (one wave per "
                                                 "chain on the per-SNP series; the headline runs every SNP from the same start)")
            except Exception as e:
                res["extra_c3_chain"] = {"error": repr(e)}
            try:
                leg = run_leg(50000, 500000, 0.0, 3, 1)
                sm = leg_summary(leg, 50000, 3)
                res["extra_c4_1gpu"] = dict({k: sm[k] for k in ("value", "unit", "steps", "ms_per_step", "m_kept",
                                                                "stages_ms_per_step", "roofline")},
                                            warmup=1, n_gpus=1,
                                            workload="synthetic HWE panel n=50000 m=500000 (BASELINE configs[3] shape) on ONE GPU, "
                                                     "-lmm, maf 0.02 geno 0.05, intercept only, missing=0.0",
                                            roofline_grm=sm["roofline_grm"], roofline_rotate=sm["roofline_rotate"])
                _PMC_SHAPE.update(n=50000, m=500000)
                q2k = res["extra_c4_1gpu"]["roofline"]["kernel"] if res["extra_c4_1gpu"].get("roofline") else None
                if q2k:
                    # the two-groups-per-pass kernel's raw FETCH_SIZE is 0.53 x the n^3 / 16 bytes it must read (16-byte row reads:
                    # the guide's x2 applies, calibrated on the known byte count; profiles/r05a_c4_pmc_hbm_traffic.json), WRITE as is
                    tr4, tr4_src = pmc_traffic_bytes("jx::" + q2k, fetch_scale=2.0 if "pair" in q2k else 1.0)
                    mu4, _ = pmc_mfma_util(q2k)
                    nl = max(1, int(res["extra_c4_1gpu"]["roofline"].get("launches_per_decomposition", 1)))
                    res["extra_c4_1gpu"]["roofline"].update(
                        traffic=tr4, traffic_source=tr4_src, mfma_util_pmc=mu4,
                        traffic_over_algorithmic=(tr4 * nl / (2.0 * 50000.0 ** 3 / 16.0)) if tr4 else None)
                _PMC_SHAPE.update(n=int(n), m=int(m))
                if not args.no_cpu_baseline:
                    # the north star's >= 10 x is quoted on THIS configuration: the oracle on the host cores, BASELINE.md section 2's
                    # sub-sample rule (first 20 000 of the 50 000 samples, first --cpu-sample SNPs, every stage scaled by its exponent)
                    try:
                        n_c = 20000
                        sub = leg["packed"][: args.cpu_sample, : n_c // 4].contiguous().cpu().numpy()
                        cb = cpu_baseline(sub, n_c, leg["y"][:n_c], "lmm", args.cpu_sample, os.cpu_count() or 1, scale_to_n=50000,
                                          m_full=500000)
                        res["extra_c4_1gpu"]["cpu_baseline"] = cb
                        res["extra_c4_1gpu"]["gpu_over_cpu_baseline"] = res["extra_c4_1gpu"]["value"] / cb["value"]
                        del sub
                    except Exception as e:   # noqa: BLE001 - a reported number, never the product path
                        res["extra_c4_1gpu"]["cpu_baseline"] = {"error": repr(e)}
                del leg
                torch.cuda.empty_cache()
            except Exception as e:
                res["extra_c4_1gpu"] = {"error": repr(e)}
            try:
                # BASELINE configs[1] (n = 5000, m = 50 000, -fvlmm: GRM + eigendecomposition + null + fixed-lambda scan)
                leg = run_leg(5000, 50000, 0.0, 5, 2, mode="fvlmm")
                sm = leg_summary(leg, 5000, 5)
                res["extra_c2_fvlmm"] = dict({k: sm[k] for k in ("value", "unit", "steps", "ms_per_step", "m_kept",
                                                                 "stages_ms_per_step", "roofline_grm", "roofline_rotate")},
                                             warmup=2, workload="synthetic HWE panel n=5000 m=50000 (BASELINE configs[1] shape), -fvlmm, "
                                                                "maf 0.02 geno 0.05, intercept only, missing=0.0")
                del leg
                torch.cuda.empty_cache()
            except Exception as e:
                res["extra_c2_fvlmm"] = {"error": repr(e)}
            try:
                res["extra_c1_mouse"] = leg_c1_mouse()
            except Exception as e:
                res["extra_c1_mouse"] = {"error": repr(e)}
            try:
                res.update(legs_c5(("splmm",)))
            except Exception as e:
                res["extra_c5_splmm"] = {"error": repr(e)}
            try:
                # the -BLUP PCG leg in a process of its own, as `jx gs` runs it: inside THIS process the first large hipMalloc behind
                # the ~100 GB the legs above released waits 4 - 6 s once (scripts/probes/vram_recycle_probe.py: 6.0 s for a 40 GB
                # block behind 120 GB released, 0.3 ms in a fresh process or ten seconds later), which is this
                # run's history and not the solver's cost
                import subprocess
                torch.cuda.empty_cache()
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", "c5_pcg"], capture_output=True, text=True,
                                    timeout=600, env=dict(os.environ, JXGPU_BENCH_CHILD="1"))
                line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                if cp.returncode != 0 or not line:
                    raise RuntimeError(f"child rc={cp.returncode}: {cp.stderr[-400:]}")
                res.update(json.loads(line[-1]))
                res["extra_c5_pcg"]["process"] = "child process of its own (python bench.py --leg c5_pcg), one GPU"
            except Exception as e:
                res["extra_c5_pcg"] = {"error": repr(e)}
        # What the multi-rank code path costs BEFORE any wire time (no multi-GPU box exists on this pool): the same
        # configuration once more in a child process with ONE rank on the RCCL path -- the GRM partials through the lower-triangle
        # pack / ncclAllReduce / unpack, the band reduction in its sharded launch sequence (per block row, two collectives per
        # panel through the counted callback), the agreement checks, the column-sharded back-transformations with their gather --
        # against the unsharded stages of the main leg above.
        if (world == 1 and not distributed and not args.no_extra and args.mode == "lmm" and args.missing == 0.0
                and (int(n), int(args.m)) == (20000, 200000) and not os.environ.get("JXGPU_BENCH_CHILD")):
            try:
                import socket
                import subprocess
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                env = dict(os.environ, JXGPU_BENCH_FORCE_DIST="1", JXGPU_DIST_EIGH_FORCE="1", JXGPU_EIGH="twostage", JXGPU_BENCH_CHILD="1",
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
                cmd = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]
                cp = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                if cp.returncode != 0 or not line:
                    raise RuntimeError(f"child rc={cp.returncode}: {cp.stderr[-400:]}")
                ch = json.loads(line[-1])
                a, b = res["stages_ms_per_step"], ch["stages_ms_per_step"]
                keys = ("grm", "eigh", "eigh_band_reduction", "eigh_bulge_chasing", "eigh_divide_conquer", "eigh_q1_backtransform",
                        "eigh_q2_backtransform_kernel", "scan")
                panels = int(math.ceil((n - 64 - 1) / 64.0))
                res["dist_overhead"] = {
                    "what": "one rank, RCCL (nccl backend, world size 1): the sharded launch sequences and their collectives against "
                            "the unsharded stages of the main leg; NOT a multi-GPU measurement",
                    "unsharded_ms": {k: a.get(k) for k in keys}, "one_rank_rccl_ms": {k: b.get(k) for k in keys},
                    "delta_ms": {k: (b.get(k) - a.get(k)) if (a.get(k) is not None and b.get(k) is not None) else None for k in keys},
                    "band_reduction_ms_per_panel": (b.get("eigh_band_reduction", 0.0) - a.get("eigh_band_reduction", 0.0)) / max(1, panels),
                    "panels": panels, "ms_per_step": ch["ms_per_step"], "parallelism": ch["config"]["parallelism"]}
            except Exception as e:   # noqa: BLE001 - a reported side measurement
                res["dist_overhead"] = {"error": repr(e)}
        # RCCL writes its version banner through C stdio (buffered when stdout is a pipe): flush it first so that the
        # JSON line is the last line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if world > 1:
            og = one_gpu_same_workload(n, args.m, args.mode)
            if og:
                res["one_gpu_same_workload"] = og
        emit(res)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
