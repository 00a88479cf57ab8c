"""BASELINE configs[4]'s `-BLUP` PCG leg at FULL SIZE, the panel GENERATED ON THE DEVICE (run as its own process by
tests/test_gpu_parity.py::test_c5_full_size_blup_pcg_device_panel so that the host RSS it reports is this run's alone).

n samples in sibships of four, m SNPs, N_TRAIN training samples; `rrblup_pcg_bed` (src/stats/rrblup.rs:3519) and
`he_pcg_bed` (src/stats/he.rs:2101) through the reference's own entry points with the payload as a device tensor
(operator: src/math/pcg.rs:300-575).  No (m x n) array ever exists on the host.  Checks:

  1. the ridge system: `(Z_c Z_c' + lambda I) beta = Z y_c` with the residual RECOMPUTED in f64 on the device from an independent
     decode (torch), over all m markers and all training samples (two chunked passes) -- size-independent property;
  2. predictions: every test sample against `alpha + Z' beta` from the same f64 decode, and a 150-sample slice against the oracle's
     prediction operator (`pcg_x_mul_samples` restated in oracle.rrblup_pcg_packed) on the marker effects the device returned;
  3. the oracle's whole solve on a marker sub-panel (first M_SUB rows, ALL training samples: 1250 sample tiles) against the
     device solve of the same sub-panel: iteration count, beta, predictions;
  4. Haseman-Elston at full size: y'PKPy, y'Py and the Hutchinson sums over the reference's splitmix64 probes against the same
     quantities from the f64 decode (same probes, generated independently here), tr(PKP) against its exact value from the
     genotype counts (within the estimator's own spread), and on a sample sub-block (N_SUB training samples) the
     exact-trace route against dense f64 traces of the sub-block's K.

    python tests/c5_pcg_driver.py N M [N_TRAIN]
Prints one JSON line.
"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

F32 = np.float32
_M64 = np.uint64((1 << 64) - 1)


def _rss_gib():
    now = peak = 0.0
    with open("/proc/self/status") as f:
        for ln in f:
            if ln.startswith("VmRSS:"):
                now = int(ln.split()[1]) / 2**20
            elif ln.startswith("VmHWM:"):
                peak = int(ln.split()[1]) / 2**20
    return round(now, 3), round(peak, 3)


def _splitmix64_vec(x):
    """splitmix64 (src/stats/he.rs:899-906) on a uint64 vector (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x.copy()
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rademacher_probes(seed, n_probes, n):
    """(n, n_probes) f32 of +-1: probe t = the chain he.rs:1871-1881 starts from splitmix64(seed ^ t * 0x517C...), all chains
    advanced together (the oracle's scalar `he_rademacher_probe` is the definition; checked against it in main)."""
    with np.errstate(over="ignore"):
        t = np.arange(n_probes, dtype=np.uint64)
        state = _splitmix64_vec(np.uint64(seed) ^ (t * np.uint64(0x517CC1B727220A95)))
    out = np.empty((n, n_probes), dtype=np.float32)
    for i in range(n):
        state = _splitmix64_vec(state)
        out[i] = np.where((state & np.uint64(1)) == 0, 1.0, -1.0)
    return out


def stats_from_counts(cnt, n_samples):
    """(maf f32, flip) as `load_bed_2bit_packed` / `bed_packed_row_flip_mask` derive them (src/io/gfreader.rs:4460-4485,
    src/stats/packed.rs:81-117) from the per-SNP genotype counts."""
    mi, he, ho = (cnt[:, k].astype(np.int64) for k in range(3))
    nm = n_samples - mi
    alt = he + 2 * ho
    ok = nm > 0
    p = np.zeros(cnt.shape[0], dtype=F32)
    p[ok] = alt[ok].astype(F32) / (F32(2.0) * nm[ok].astype(F32))
    maf = np.where(ok, np.minimum(p, F32(1.0) - p), F32(0.0)).astype(F32)
    flip = np.zeros(cnt.shape[0], dtype=bool)
    flip[ok] = (alt[ok].astype(np.float64) / (2.0 * nm[ok].astype(np.float64))) > 0.5
    return maf, flip


class Decoder:
    """Independent f64 decode of the payload on the device (torch only: shifts, masks, selects)."""

    def __init__(self, packed_t, n, cols):
        import torch
        self.t, self.pk, self.n = torch, packed_t, n
        self.cols = torch.from_numpy(np.asarray(cols, dtype=np.int64)).to(packed_t.device)

    def codes(self, r0, r1, cols=None):
        t = self.t
        b = self.pk[r0:r1]
        c = t.stack([(b >> (2 * k)) & 3 for k in range(4)], dim=2).reshape(r1 - r0, -1)[:, : self.n]
        return c.index_select(1, self.cols if cols is None else cols)

    def values(self, codes, lut64):
        """lut64 (rows, 4) f64 on the device -> (rows, cols) f64; code 1 (missing) -> lut[:, 1]."""
        t = self.t
        z = t.where(codes == 0, lut64[:, 0:1], t.where(codes == 2, lut64[:, 2:3], t.where(codes == 3, lut64[:, 3:4], lut64[:, 1:2])))
        return z


def main():
    import torch
    import bench
    from janusx_amd import janusx as jxrs
    from janusx_amd._lib import lib
    from oracle import jx_oracle as O
    n, m = int(sys.argv[1]), int(sys.argv[2])
    n_train = int(sys.argv[3]) if len(sys.argv) > 3 else int(0.8 * n)
    m_sub = int(os.environ.get("JX_C5PCG_MSUB", "1000"))
    n_sub = int(os.environ.get("JX_C5PCG_NSUB", "1200"))
    chunk = int(os.environ.get("JX_C5PCG_CHUNK", "4096"))
    dev = torch.device("cuda", 0)
    lib()
    (torch.ones(8, device=dev) @ torch.ones(8, device=dev)).item()
    torch.cuda.synchronize()
    rss = {"init": _rss_gib()}
    # the vectorised probe generator is the oracle's chain
    pr = rademacher_probes(20260512, 3, 50)
    for t_ in range(3):
        assert np.array_equal(pr[:, t_], O.he_rademacher_probe(20260512, t_, 50))
    t0 = time.perf_counter()
    packed_t, dos = bench.family_panel_gpu(n, m, 4, 11, dev)
    y_all = bench.make_phenotype(dos, n, 7, dev)
    torch.cuda.synchronize()
    res = {"n": n, "m": m, "n_train": n_train, "gen_s": time.perf_counter() - t0}
    cnt = jxrs.bed_row_counts(packed_t, n)
    maf, flip = stats_from_counts(cnt, n)
    rng = np.random.default_rng(5)
    perm = rng.permutation(n)
    tr = np.sort(perm[:n_train]).astype(np.int64)
    te = np.sort(perm[n_train:]).astype(np.int64)
    y = np.ascontiguousarray(y_all[tr])
    lam = float(m)                      # h2 = 0.5 on standardised markers: lambda = m (1 - h2) / h2
    tol, max_iter = 1e-6, 200
    rss["prepared"] = _rss_gib()

    # ---------------- the device route, reference entry point, payload as a device tensor
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = jxrs.rrblup_pcg_bed("", tr, y, te, lambda_value=lam, tol=tol, max_iter=max_iter, packed=packed_t, packed_n_samples=n,
                              maf=maf, row_flip=flip)
    torch.cuda.synchronize()
    t_pcg = time.perf_counter() - t0
    L = lib()
    loop_ms, iters_k, op_ms, setup_ms = (float(L.jxg_last_kernel_ms(k)) for k in (18, 19, 20, 21))
    pred_tr, pred_te, _pve_tv, converged, iters, rel_res, m_eff, pve_vc, k_trace, beta = out
    res.update(pcg_s=t_pcg, pcg_setup_ms=setup_ms, pcg_loop_ms=loop_ms, pcg_operator_kernels_ms=op_ms, iters=int(iters),
               converged=bool(converged), rel_res=float(rel_res), m_effective=int(m_eff), k_trace_mean=float(k_trace),
               pve_lambda_vc=float(pve_vc),
               # HBM roofline of the streaming operator: 2 x n_train x m / 4 payload bytes per application
               operator_payload_gbs=(2.0 * n_train * m / 4.0) * max(iters, 1) / max(op_ms, 1e-9) / 1e6,
               snps_per_s=m / t_pcg)
    rss["solved"] = _rss_gib()

    # ---------------- Haseman-Elston at full size (same operator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    he = jxrs.he_pcg_bed("", tr, y, packed=packed_t, packed_n_samples=n, maf=maf, row_flip=flip, trace_samples=32)
    torch.cuda.synchronize()
    t_he = time.perf_counter() - t0
    he_ms, he_apps = float(L.jxg_last_kernel_ms(22)), float(L.jxg_last_kernel_ms(23))
    res.update(he_s=t_he, he_operator_kernels_ms=he_ms, he_operator_applications=int(he_apps),
               he_operator_payload_gbs=(2.0 * n_train * m / 4.0) * max(he_apps, 1.0) / max(he_ms, 1e-9) / 1e6,
               he_sigma_g2=he[0], he_sigma_e2=he[1], he_h2=he[2], he_m_effective=int(he[6]))
    rss["he"] = _rss_gib()

    # ---------------- checker: independent f64 decode on the device, two chunked passes over all m markers
    dec = Decoder(packed_t, n, tr)
    te_t = torch.from_numpy(te).to(dev)
    # rrBLUP value table (oracle: rrblup_row_standardization + rrblup_value_lut) and the HE one (oracle: he_row_standardization
    # with the training samples' own frequency, from the device counts of the training samples)
    rm, ri, m_eff_ref = O.rrblup_row_standardization(maf, F32(1e-12))
    lut_rr = torch.from_numpy(O.rrblup_value_lut(rm, ri, flip).astype(np.float64)).to(dev)
    cnt_tr = jxrs.bed_row_counts(packed_t, n, tr).astype(np.int64)
    nm = n_train - cnt_tr[:, 0]
    alt = cnt_tr[:, 1] + 2 * cnt_tr[:, 2]
    dosg = np.where(flip, 2 * nm - alt, alt)
    p_he = np.clip(np.where(nm > 0, dosg.astype(F32) / (F32(2.0) * np.maximum(nm, 1).astype(F32)), maf), F32(0), F32(1)).astype(F32)
    he_mean = (F32(2.0) * p_he).astype(F32)
    he_var = np.maximum((F32(2.0) * p_he * (F32(1.0) - p_he)).astype(F32), F32(0.0))
    he_good = he_var > F32(1e-12)
    he_inv = np.zeros_like(he_var)
    he_inv[he_good] = (F32(1.0) / np.sqrt(he_var[he_good])).astype(F32)
    lut_he = torch.from_numpy(O.rrblup_value_lut(he_mean, he_inv, flip).astype(np.float64)).to(dev)
    m_he = int(np.count_nonzero(he_good))
    beta_t = torch.from_numpy(beta.astype(np.float64)).to(dev)
    y_mean = float(np.sum(y)) / n_train
    yc = torch.from_numpy((y - y_mean).astype(np.float32).astype(np.float64)).to(dev)
    # HE right-hand sides: [P y | P v_t] rounded to f32 as the reference holds them (he.rs:1633-1700, :1871-1881)
    probes = rademacher_probes(20260512, 32, n_train).astype(np.float64)
    yp = y - y.mean()
    pv = probes - probes.mean(axis=0, keepdims=True)
    rhs = torch.from_numpy(np.concatenate([yp[:, None], pv], axis=1).astype(np.float32).astype(np.float64)).to(dev)   # (n_train, 33)
    t_vec = torch.zeros(n_train, dtype=torch.float64, device=dev)
    mu = torch.zeros(m, dtype=torch.float64, device=dev)
    bvec = torch.zeros(m, dtype=torch.float64, device=dev)
    pred_te_ref = torch.zeros(len(te), dtype=torch.float64, device=dev)
    he_t = torch.zeros((m, 33), dtype=torch.float64, device=dev)
    ss_he = 0.0
    t0 = time.perf_counter()
    for r0 in range(0, m, chunk):
        r1 = min(m, r0 + chunk)
        c = dec.codes(r0, r1)
        z = dec.values(c, lut_rr[r0:r1])
        t_vec += z.T @ beta_t[r0:r1]
        mu[r0:r1] = z.mean(dim=1)
        bvec[r0:r1] = z @ yc
        zh = dec.values(c, lut_he[r0:r1])
        he_t[r0:r1] = zh @ rhs
        zc = zh - zh.mean(dim=1, keepdim=True)
        ss_he += float((zc * zc).sum())
        del z, zh, zc, c
        ct = dec.codes(r0, r1, te_t)
        pred_te_ref += dec.values(ct, lut_rr[r0:r1]).T @ beta_t[r0:r1]
        del ct
    mu_beta = float(torch.dot(mu, beta_t))
    ab = torch.zeros(m, dtype=torch.float64, device=dev)
    he_kv = torch.zeros((n_train, 33), dtype=torch.float64, device=dev)
    he_t32 = he_t.to(torch.float32).to(torch.float64)          # the reference's GEMV output is f32
    for r0 in range(0, m, chunk):
        r1 = min(m, r0 + chunk)
        c = dec.codes(r0, r1)
        z = dec.values(c, lut_rr[r0:r1])
        ab[r0:r1] = z @ t_vec
        zh = dec.values(c, lut_he[r0:r1])
        he_kv += zh.T @ he_t32[r0:r1]
        del z, zh, c
    torch.cuda.synchronize()
    res["checker_s"] = time.perf_counter() - t0
    ab = ab - float(n_train) * mu * mu_beta + lam * beta_t
    res["ridge_residual"] = float(torch.linalg.norm(bvec - ab) / torch.linalg.norm(bvec))
    alpha = y_mean - mu_beta
    scale = float(np.std(y))
    res["pred_test_err"] = float(np.max(np.abs(pred_te.ravel() - (pred_te_ref.cpu().numpy() + alpha))) / scale)
    res["pred_train_err"] = float(np.max(np.abs(pred_tr.ravel() - (t_vec.cpu().numpy() + alpha))) / scale)
    # Haseman-Elston sufficient statistics from the f64 decode
    inv_m = 1.0 / max(float(m_he), 1.0)
    kv = he_kv * inv_m
    y32 = rhs[:, 0]
    y_ky_ref = float(torch.dot(y32, kv[:, 0]))
    y_y_ref = float(torch.dot(y32, y32))
    kvp = kv[:, 1:] - kv[:, 1:].mean(dim=0, keepdim=True)
    tr_k_ref = float((rhs[:, 1:] * kvp).sum()) / 32.0
    tr_k2_ref = float((kvp * kvp).sum()) / 32.0
    tr_k_exact = ss_he * inv_m                                     # tr(P K P) = sum_j |P z_j|^2 / m, P = centring
    res.update(he_y_ky_err=abs(he[8] - y_ky_ref) / abs(y_ky_ref), he_y_y_err=abs(he[9] - y_y_ref) / abs(y_y_ref),
               he_tr_k2_err=abs(he[7] - tr_k2_ref) / abs(tr_k2_ref), he_m_effective_equal=bool(int(he[6]) == m_he),
               he_tr_k_exact=tr_k_exact, he_tr_k_hutchinson_ref=tr_k_ref)
    # sigma_g2 / sigma_e2 from the reference's 2x2 system on the checker's statistics
    tr_p = max(float(n_train) - 1.0, 1.0)
    k2s = max(tr_k2_ref, tr_k_ref * tr_k_ref / tr_p + tr_p * 1e-6)
    sg_u, se_u = O.he_solve_2x2(k2s, tr_k_ref, tr_p, y_ky_ref, y_y_ref)
    sg, se, _pj, _stt = O.he_project_nnls_2x2(k2s, tr_k_ref, tr_p, y_ky_ref, y_y_ref, sg_u, se_u)
    res["he_sigma_err"] = max(abs(he[0] - sg), abs(he[1] - se)) / (abs(sg) + abs(se))
    res["he_tr_k_vs_exact"] = abs(tr_k_ref - tr_k_exact) / tr_k_exact
    del he_t, he_t32, he_kv, kv, kvp, ab, bvec
    rss["checked"] = _rss_gib()

    # ---------------- 150-sample slice of the test predictions against the oracle's prediction operator
    n_pick = min(150, len(te))
    te_pick = te[:n_pick]
    tp = torch.from_numpy(te_pick).to(dev)
    small = torch.empty((m, (n_pick + 3) // 4), dtype=torch.uint8, device=dev)
    for r0 in range(0, m, 65536):
        r1 = min(m, r0 + 65536)
        c = dec.codes(r0, r1, tp)
        pad = (-n_pick) % 4
        if pad:
            c = torch.nn.functional.pad(c, (0, pad))
        c4 = c.view(r1 - r0, -1, 4)
        small[r0:r1] = c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6)
    small_h = small.cpu().numpy()
    del small
    lut_h = O.rrblup_value_lut(rm, ri, flip)
    acc = np.zeros(n_pick, dtype=np.float64)
    for r0 in range(0, m, 100000):       # f32 GEMV of the reference, in row blocks of the payload as `pcg_x_mul_samples` streams them
        r1 = min(m, r0 + 100000)
        cz = O.unpack_codes(small_h[r0:r1], n_pick).astype(np.int64)
        zz = np.take_along_axis(lut_h[r0:r1], cz, axis=1).astype(np.float32)
        acc += (zz.T @ beta[r0:r1]).astype(np.float32)
    mean32 = mu.cpu().numpy().astype(np.float32)
    alpha32 = F32(F32(y_mean) - F32(np.dot(mean32.astype(np.float64), beta.astype(np.float64))))
    pred_or = acc.astype(np.float32).astype(np.float64) + float(alpha32)
    idx_pick = np.searchsorted(te, te_pick)
    res["pred_oracle_slice_err"] = float(np.max(np.abs(pred_te.ravel()[idx_pick] - pred_or)) / scale)
    rss["oracle_slice"] = _rss_gib()

    # ---------------- the oracle's whole solve on a marker sub-panel with ALL training samples
    sub_h = packed_t[:m_sub].cpu().numpy()
    lam_s = float(m_sub) * 20.0
    ref = O.rrblup_pcg_packed(sub_h, n, maf[:m_sub], flip[:m_sub], tr, y, te_pick, None, None, lam_s, 1e-7, 200)
    got = jxrs.rrblup_pcg_bed("", tr, y, te_pick, lambda_value=lam_s, tol=1e-7, max_iter=200, packed=packed_t[:m_sub],
                              packed_n_samples=n, maf=maf[:m_sub], row_flip=flip[:m_sub])
    bmax = float(np.max(np.abs(ref[9])))
    res.update(sub_iters_ref=int(ref[4]), sub_iters=int(got[4]), sub_converged=bool(got[3] and ref[3]),
               sub_beta_err=float(np.max(np.abs(got[9] - ref[9])) / bmax),
               sub_pred_train_err=float(np.max(np.abs(got[0] - ref[0])) / scale),
               sub_pred_test_err=float(np.max(np.abs(got[1] - ref[1])) / scale),
               sub_k_trace_err=float(abs(got[8] - ref[8]) / abs(ref[8])))
    del ref, sub_h
    rss["oracle_sub"] = _rss_gib()

    # ---------------- HE exact-trace route on a sample sub-block against dense f64 traces of that block's K
    tr_s = tr[:n_sub]
    y_s = np.ascontiguousarray(y[:n_sub])
    he_s = jxrs.he_pcg_bed("", tr_s, y_s, packed=packed_t, packed_n_samples=n, maf=maf, row_flip=flip, exact_trace_debug=True,
                           exact_trace_max_n=n_sub)
    cnt_s = jxrs.bed_row_counts(packed_t, n, tr_s).astype(np.int64)
    nm_s = n_sub - cnt_s[:, 0]
    alt_s = cnt_s[:, 1] + 2 * cnt_s[:, 2]
    dos_s = np.where(flip, 2 * nm_s - alt_s, alt_s)
    p_s = np.clip(np.where(nm_s > 0, dos_s.astype(F32) / (F32(2.0) * np.maximum(nm_s, 1).astype(F32)), maf), F32(0), F32(1)).astype(F32)
    var_s = np.maximum((F32(2.0) * p_s * (F32(1.0) - p_s)).astype(F32), F32(0.0))
    good_s = var_s > F32(1e-12)
    inv_s = np.zeros_like(var_s)
    inv_s[good_s] = (F32(1.0) / np.sqrt(var_s[good_s])).astype(F32)
    lut_s = torch.from_numpy(O.rrblup_value_lut((F32(2.0) * p_s).astype(F32), inv_s, flip).astype(np.float64)).to(dev)
    ts = torch.from_numpy(tr_s).to(dev)
    k_s = torch.zeros((n_sub, n_sub), dtype=torch.float64, device=dev)
    for r0 in range(0, m, 32768):
        r1 = min(m, r0 + 32768)
        zs = dec.values(dec.codes(r0, r1, ts), lut_s[r0:r1])
        k_s += zs.T @ zs
    k_s /= float(int(np.count_nonzero(good_s)))
    cen = torch.eye(n_sub, dtype=torch.float64, device=dev) - 1.0 / n_sub
    pkp = cen @ k_s @ cen
    ys = torch.from_numpy(y_s - y_s.mean()).to(dev)
    fro2 = float((pkp * pkp).sum())
    yky_s = float(ys @ k_s @ ys)
    res.update(he_sub_tr_k2_err=abs(fro2 - he_s[7]) / fro2, he_sub_y_ky_err=abs(yky_s - he_s[8]) / abs(yky_s),
               he_sub_y_y_err=abs(float(ys @ ys) - he_s[9]) / float(ys @ ys),
               he_sub_m_effective_equal=bool(int(he_s[6]) == int(np.count_nonzero(good_s))))
    res["peak_hbm_gib"] = torch.cuda.max_memory_allocated() / 2**30
    rss["done"] = _rss_gib()
    res["host_rss_gib"] = rss
    res["host_maxrss_gib"] = rss["done"][1]
    res["host_rss_growth_gib"] = res["host_maxrss_gib"] - rss["init"][1]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
