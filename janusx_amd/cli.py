"""Thin `jx gwas` / `jx grm` / `jx gs` command line for the accelerated path.

Flag names, defaults and output file names follow the reference (python/janusx/assoc/workflow.py:6599-7047,
python/janusx/script/grm.py:18-23, 1874-1975, python/janusx/assoc/workflow_model_stream.py:989-994):

  python -m janusx_amd gwas -bfile PREFIX -p PHENO.tsv [-n TRAIT ...] (-lmm | -lmm2 | -fvlmm) [-k 1|2|GRM.npy] [-c COV.tsv]
                            [-maf 0.02] [-geno 0.05] [-het 1.0] [-o OUT] [-force-model]
  python -m janusx_amd grm  -bfile PREFIX [-m 1|2] [-maf 0.02] [-geno 0.05] [-o OUT]
  python -m janusx_amd gs   -bfile PREFIX -p PHENO.tsv [-n TRAIT ...] -BLUP [-cv K] [-seed 42] [-k GRM.npy]
  python -m janusx_amd gs   -bfile PREFIX -p PHENO.tsv [-n TRAIT ...] -rrBLUP [-lambda L] [-tol 1e-4] [-max-iter 100] [-cv K]
                            [-maf 0.02] [-geno 0.05] [-o OUT]

Outputs: `{out}.{trait}.lmm.tsv` / `.lmm2.tsv` / `.fvlmm.tsv` / `.splmm2.tsv` (`gwas -splmm-exact [cutoff]`, exact SparseLMM scan) / `.splmm.tsv` (`gwas -splmm [cutoff]`, GRAMMAR-gamma SparseLMM scan); `{out}.cGRM.npy` (method 1) or `.sGRM.npy` (method 2) + `.npy.id`; `{out}.spgrm` + `.spgrm.id` with `grm -sparse [cutoff]`;
`{out}.{trait}.gs.GBLUP.tsv` (sample, observed, predicted, fold) for `gs`.
Only PLINK BED input, the additive model, the -lmm / -fvlmm scans and the GBLUP branch of `-BLUP`
(python/janusx/gs/blup.py:72-163 routes n <= BLUP_SMALL_N there; `gblup_reml_npy_grm` call of
python/janusx/gs/workflow.py:9122) are built; VCF/HMP readers, PCs (-q), plots, the history DB, the
other GS model families are out of scope; `-rrBLUP` runs the exact marker-space route (`rrblup_exact_snp_packed`, REML
lambda from the spectrum) up to 15 000 kept markers and the PCG route (`rrblup_pcg_bed`, HE / manual / subsample-REML
lambda) beyond, `-rr-solver exact|fast|pcg` forces one (fast = exact sample-space route for n_train <= 10 000).
"""
from __future__ import annotations

import argparse
import contextlib
import math
import os
import sys
import time

import numpy as np


SPLMM_APPROX_RHAT_MARKERS = 1000      # python/janusx/assoc/workflow_model_packed.py:99


def _read_table(path):
    """Tab/space delimited table with a header; first column = sample id. -> (ids, names, values float with NaN)."""
    with open(path) as fh:
        header = fh.readline().rstrip("\n").replace(",", "\t").split()
        ids, rows = [], []
        for line in fh:
            parts = line.rstrip("\n").replace(",", "\t").split()
            if not parts:
                continue
            ids.append(parts[0])
            vals = []
            for v in parts[1:]:
                try:
                    vals.append(float(v))
                except ValueError:
                    vals.append(float("nan"))
            rows.append(vals)
    width = max(len(r) for r in rows) if rows else 0
    arr = np.full((len(rows), width), np.nan)
    for i, r in enumerate(rows):
        arr[i, :len(r)] = r
    names = header[1:] if len(header) == width + 1 else [f"trait{i}" for i in range(width)]
    return ids, names, arr


def _select_traits(names, specs):
    if not specs:
        return list(range(len(names)))
    out = []
    for spec in specs:
        for tok in str(spec).split(","):
            tok = tok.strip()
            if ":" in tok and all(t.isdigit() for t in tok.split(":")):
                a, b = [int(t) for t in tok.split(":")]
                out.extend(range(a, b + 1))
            elif tok.isdigit():
                out.append(int(tok))
            elif tok in names:
                out.append(names.index(tok))
            else:
                raise SystemExit(f"unknown trait selector '{tok}'")
    return out


def _load_grm(path, fam_ids):
    k = np.load(path)
    id_path = path + ".id"
    if os.path.exists(id_path):
        ids = [ln.split()[0] for ln in open(id_path) if ln.strip()]
        if ids != list(fam_ids):
            pos = {s: i for i, s in enumerate(ids)}
            try:
                order = np.array([pos[s] for s in fam_ids])
            except KeyError as e:
                raise SystemExit(f"GRM id file lacks sample {e}") from None
            k = k[np.ix_(order, order)]
    if k.shape != (len(fam_ids), len(fam_ids)):
        raise SystemExit(f"GRM shape {k.shape} does not match {len(fam_ids)} samples")
    return np.ascontiguousarray(k, dtype=np.float32)


def _resolve_out(args, src):
    """Output prefix as the reference resolves it (python/janusx/assoc/workflow.py:6907-6921, script/_common/cli_args.py:727-760):
    -o PREFIX or DIR/PREFIX (an existing directory or a trailing separator: the input's basename inside it), -prefix NAME names
    the files inside -o DIR; nothing given: the input's basename in the current directory."""
    base = os.path.basename(src[:-4] if src.lower().endswith((".bed", ".bim", ".fam", ".npy")) else src)
    o, pfx = getattr(args, "out", None), getattr(args, "prefix", None)
    if pfx:
        res = os.path.join(o or ".", pfx)
    elif o is None:
        res = base
    elif o.endswith(os.sep) or os.path.isdir(o):
        res = os.path.join(o, base)
    else:
        res = o
    parent = os.path.dirname(res)
    if parent:
        os.makedirs(parent, exist_ok=True)
    return res


def _dist_setup():
    """Under a launcher (`python -m torch.distributed.run --nproc-per-node N -m janusx_amd gwas ...`: WORLD_SIZE > 1) bind this
    process to its GPU, join the process group (backend JXGPU_DIST_BACKEND, default "nccl" = RCCL over xGMI; "gloo" lets several
    ranks share one GPU in the functional tests) and switch the rank-aware pipeline on: SNP-sharded GRM + one reduction, the
    eigenvectors shared out over the ranks, SNP-sharded scan, rows gathered in BED order.  Rank 0 alone prints
    and writes.  -> (rank, world); (0, 1) without a launcher."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    from . import pipeline as pl
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("JXGPU_DIST_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    if backend == "nccl" and local >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local} has no GPU of its own ({ndev} visible); one process per GPU")
    torch.cuda.set_device(local % ndev)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local % ndev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    pl.init_distributed()
    if rank != 0:
        sys.stdout = open(os.devnull, "w")      # one voice: rank 0 reports
    return rank, world


def cmd_grm(args):
    from . import janusx as jxrs
    from .bed import read_fam_ids
    if args.grm is None and not args.bfile:
        raise SystemExit("grm needs -bfile PREFIX (or -grm FILE.npy -sparse [cutoff])")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and args.grm is not None:
        raise SystemExit("thresholding an existing dense GRM runs on one GPU: start it without the launcher")
    out = _resolve_out(args, args.bfile or args.grm)
    t0 = time.perf_counter()
    if args.grm is not None:
        # python/janusx/script/grm.py:1806-1868, 1896: an existing dense GRM (`.npy` + sibling `.id`) thresholded into `.spgrm`
        if args.sparse is None:
            raise SystemExit("-grm FILE.npy must be used with -sparse [cutoff]")
        if not os.path.exists(args.grm + ".id"):
            raise SystemExit(f"{args.grm}.id not found (sample ids of the dense GRM)")
        path, n, nnz = jxrs.spgrm_dense_npy_to_jxgrm(args.grm, out, float(args.sparse))
        ids_in = open(args.grm + ".id").read().split()
        if len(ids_in) != n:
            raise SystemExit(f"{args.grm}.id lists {len(ids_in)} samples, the matrix has {n}")
        with open(path + ".id", "w") as fh:
            for sid in ids_in:
                fh.write(f"{sid}\n")
        print(f"Sparse GRM from {args.grm}: n={n} nnz={nnz} cutoff={args.sparse} -> {path} "
              f"({time.perf_counter() - t0:.2f}s)")
        return 0
    if args.sparse is not None:
        # python/janusx/script/grm.py:1574-1675 (`-sparse [cutoff]`): thresholded lower-triangle CSC `.spgrm` + `.id`; under
        # the launcher the row panels of the matrix are dealt over the ranks (janusx._spgrm_packed)
        rank, world = _dist_setup()
        path, n, nnz = jxrs.spgrm_bed_to_jxgrm(args.bfile, out_prefix=out, method=args.method,
                                               threshold=float(args.sparse), maf_threshold=args.maf,
                                               max_missing_rate=args.geno, het_threshold=0.0,
                                               snps_only=bool(getattr(args, "snps_only", False)))
        if rank == 0:
            with open(path + ".id", "w") as fh:
                for sid in read_fam_ids(args.bfile):
                    fh.write(f"{sid}\n")
        print(f"Sparse GRM method {args.method}: n={n} nnz={nnz} cutoff={args.sparse} -> {path} "
              f"({time.perf_counter() - t0:.2f}s)")
        return 0
    rank, world = _dist_setup()
    if world > 1:
        # one process per GPU: every rank accumulates its share of the kept SNPs, one reduction over xGMI (pipeline.build_grm)
        import torch
        from . import pipeline as pl
        from .bed import snps_only_mask, stage_bed_payload
        packed_t, n, bim = stage_bed_payload(args.bfile, None)
        if getattr(args, "snps_only", False):
            packed_t = packed_t[torch.from_numpy(np.nonzero(snps_only_mask(bim))[0]).to(packed_t.device)]
        k_t, eff, _ = pl.build_grm(packed_t, n, args.method, args.maf, args.geno)
        if rank != 0:
            return 0
        k = k_t.cpu().numpy()
    else:
        k, eff, n = jxrs.grm_stream_bed_f32(args.bfile, method=args.method, maf_threshold=args.maf,
                                            max_missing_rate=args.geno, het_threshold=0.0,
                                            snps_only=bool(getattr(args, "snps_only", False)))
    tag = "cGRM" if args.method == 1 else "sGRM"
    if getattr(args, "txt", False):             # python/janusx/script/grm.py:2684-2690
        path = f"{out}.{tag}.txt"
        tmp = f"{path}.tmp.{os.getpid()}"
        np.savetxt(tmp, k, fmt="%.6f")
    else:
        path = f"{out}.{tag}.npy"
        tmp = f"{path}.tmp.{os.getpid()}"
        with open(tmp, "wb") as fh:
            np.lib.format.write_array(fh, k, version=(1, 0))
    os.replace(tmp, path)
    with open(path + ".id", "w") as fh:
        for sid in read_fam_ids(args.bfile):
            fh.write(f"{sid}\n")
    print(f"GRM method {args.method}: n={n} eff_m={eff} -> {path} ({time.perf_counter() - t0:.2f}s)")
    return 0


def build_cv_splits(n_samples, n_splits, seed=42):
    """`build_cv_splits` (python/janusx/gs/workflow.py:3950-3980) over the `KFold` of python/janusx/pyBLUP/kfold.py:28-88:
    one permutation of arange(n) from numpy's default_rng(seed), cut into n_splits runs whose sizes differ by at most one
    (the first n % k folds are the longer ones); each item is (test_idx, train_idx), train in ascending order."""
    n, k = int(n_samples), int(n_splits)
    if n < 2:
        raise ValueError(f"CV requires at least 2 samples, got {n}.")
    if k < 2:
        raise ValueError(f"CV folds must be >=2, got {k}.")
    if k > n:
        raise ValueError(f"CV folds ({k}) cannot exceed sample size ({n}).")
    idx = np.asarray(np.random.default_rng(int(seed)).permutation(np.arange(n, dtype=np.int64)), dtype=np.int64)
    sizes = np.full(k, n // k, dtype=np.int64)
    sizes[: n % k] += 1
    out, cur = [], 0
    for fs in sizes:
        te = idx[cur:cur + int(fs)]
        cur += int(fs)
        mask = np.zeros(n, dtype=bool)
        mask[te] = True
        out.append((te, np.nonzero(~mask)[0].astype(np.int64)))
    return out


def cv_fold_metrics(y_true, y_pred):
    """Per-fold metrics of the reference's GS summary (python/janusx/gs/workflow.py:880-930): Pearson r, Spearman rho,
    R2 = 1 - SS_res / SS_tot of the held-out fold."""
    from scipy.stats import pearsonr, spearmanr
    yt, yp = np.asarray(y_true, dtype=np.float64), np.asarray(y_pred, dtype=np.float64)
    ss_res = float(np.sum((yt - yp) ** 2))
    ss_tot = float(np.sum((yt - float(np.mean(yt))) ** 2))
    r2 = 1.0 - ss_res / ss_tot if ss_tot > 0.0 else 0.0
    return float(pearsonr(yt, yp).statistic), float(spearmanr(yt, yp).statistic), r2


def _parse_qcov_dim(qcov_opt) -> int:
    """`_parse_qcov_dim` (python/janusx/assoc/workflow.py:1813-1828): -q takes a PC count, not a file."""
    s = str(qcov_opt if qcov_opt is not None else "").strip()
    if s == "":
        raise SystemExit("Invalid -q/--qcov: empty value. Use an integer PC dimension (>=0).")
    try:
        q = int(s)
    except ValueError:
        raise SystemExit("External Q matrix via -q/--qcov is no longer supported. Use -c <file> for external covariates and "
                         "-q <int> for the PC dimension.")
    if q < 0:
        raise SystemExit(f"Invalid -q/--qcov: {q}. Q/PC dimension must be >= 0.")
    return q


def _parse_cov_site_token(token):
    """`_parse_cov_site_token` (python/janusx/assoc/workflow.py:1742-1776): chr:pos or chr:start:end with start = end."""
    parts = [p.strip() for p in str(token).strip().replace("\uff1a", ":").split(":")]
    if len(parts) not in (2, 3) or parts[0] == "":
        return None
    try:
        start = int(float(parts[1]))
    except Exception:
        return None
    if start <= 0:
        raise SystemExit(f"Invalid site position in --cov: {token}")
    if len(parts) == 3:
        try:
            end = int(float(parts[2]))
        except Exception:
            return None
        if end <= 0:
            raise SystemExit(f"Invalid site position in --cov: {token}")
        if end != start:
            raise SystemExit(f"--cov site token must specify a single site (start=end), got: {token}")
    return parts[0], start


def _canon_site_key(chrom, pos):
    c = str(chrom).strip().lower()
    return (c[3:] if c.startswith("chr") else c), int(pos)


def _warm_start_mode(args):
    """'chain' | 'none' for `jx gwas -lmm`: the flag, else the reference's rule (chain unless JX_LMM_UNIFIED_NO_WARM_START)."""
    from .stats import env_truthy
    if getattr(args, "warm_start", None) is not None:
        return args.warm_start
    return "none" if env_truthy("JX_LMM_UNIFIED_NO_WARM_START") else "chain"


def cmd_gwas(args):
    import torch
    from . import janusx as jxrs
    from . import pipeline as pl
    from .bed import read_bed_payload, read_fam_ids
    from .tsv import AsyncAssocTsvWriter, write_assoc_tsv
    # SparseLMM flags of the reference (python/janusx/assoc/workflow.py:6689-6725, 6992-7017): -splmm-exact = exact g'Pg scan,
    # raw REML null objective, result stem "splmm2"; -splmm = fastGWA fixed-Vp null objective + residualised GRAMMAR-gamma
    # denominator from 1000 sampled markers, stem "splmm" (a trait with fewer kept markers than that falls back to the exact
    # scan, workflow_model_packed.py:8087-8107)
    sp_stems = []
    if getattr(args, "splmm_exact", None) is not None:
        sp_stems.append("splmm2")
    if args.splmm is not None:
        sp_stems.append("splmm")
        if getattr(args, "splmm_exact", None) is not None and float(args.splmm_exact) != float(args.splmm):
            raise SystemExit("-splmm and -splmm-exact in one run must use the same sparse-GRM cut-off")
    if args.splmm is None and sp_stems:
        args.splmm = float(args.splmm_exact)
    if not (args.lmm or args.fvlmm or args.lmm2 or args.splmm is not None):
        raise SystemExit("select at least one model: -lmm, -lmm2, -fvlmm, -splmm and/or -splmm-exact")
    rank, world = _dist_setup()
    # the payload goes to HBM in windows (bed.stage_bed_payload: `mmap_window_mb` of the reference's BED routes); nothing
    # below holds a host copy of it
    from .bed import stage_bed_payload
    packed_t, n_fam, bim = stage_bed_payload(args.bfile, getattr(args, "mmap_window_mb", None))
    if getattr(args, "snps_only", False):
        # -snps-only: sites whose two alleles are not single A/C/G/T leave the run altogether -- GRM, QC and scan
        # (`snps_only` of the reference's BED routes, src/io/gfreader.rs:7013-7019)
        from .bed import Bim, snps_only_mask
        mask = snps_only_mask(bim)
        if not mask.all():
            sel = np.nonzero(mask)[0]
            packed_t = packed_t[torch.from_numpy(sel).to(packed_t.device)]
            bim = Bim([bim.chrom[j] for j in sel], [bim.snp[j] for j in sel], [bim.pos[j] for j in sel],
                      [bim.a0[j] for j in sel], [bim.a1[j] for j in sel])
            print(f"-snps-only: {len(sel)} of {len(mask)} sites kept")
    packed = packed_t
    fam = read_fam_ids(args.bfile)
    ids, names, ph = _read_table(args.pheno)
    pos = {s: i for i, s in enumerate(ids)}
    # -c may be repeated; an item is a covariate table or a SNP site `chr:pos` (`chr:start:end` with start = end) whose additive
    # genotype -- missing calls at the site's mean over all genotyped samples -- becomes a covariate (conditional analysis;
    # `_load_covariates_for_models` / `_load_site_covariates`, python/janusx/assoc/workflow.py:1742-1789, 1979-2145).  The
    # columns of all items are joined on the samples they share.
    cov_items = args.cov if isinstance(args.cov, list) else ([args.cov] if args.cov else [])
    args.cov = bool(cov_items)
    cv, cpos = None, {}
    if cov_items:
        parts = []
        for item in cov_items:
            site = _parse_cov_site_token(item)
            if site is None:
                if not os.path.isfile(item):
                    print(f"Covariate file not found: {item}; skipped.")
                    continue
                cids_i, _, cv_i = _read_table(item)
                parts.append((list(cids_i), np.asarray(cv_i, dtype=np.float64).reshape(len(cids_i), -1)))
            else:
                key = _canon_site_key(*site)
                hit = next((j for j in range(len(bim.chrom)) if _canon_site_key(bim.chrom[j], bim.pos[j]) == key), None)
                if hit is None:
                    raise SystemExit(f"Some --cov SNP site(s) were not found in genotype: {site[0]}:{site[1]}")
                row = packed_t[hit].cpu().numpy()
                codes = ((row[:, None] >> np.array([0, 2, 4, 6], dtype=np.uint8)) & 3).reshape(-1)[:n_fam]
                gv = np.array([0.0, np.nan, 1.0, 2.0])[codes]
                gv[np.isnan(gv)] = float(np.nanmean(gv)) if np.isfinite(gv).any() else 0.0
                parts.append((list(fam), gv.reshape(-1, 1)))
        if parts:
            common = set(parts[0][0])
            for ids_i, _m in parts[1:]:
                common &= set(ids_i)
            cids = [sid for sid in fam if sid in common]
            if not cids:
                raise SystemExit("No overlapping samples between genotype and covariates.")
            cols = []
            for ids_i, m_i in parts:
                ix = {sid: i for i, sid in enumerate(ids_i)}
                cols.append(m_i[[ix[sid] for sid in cids]])
            cv = np.concatenate(cols, axis=1)
            cpos = {s: i for i, s in enumerate(cids)}
        else:
            args.cov = False
    traits = _select_traits(names, args.ncol)
    out = _resolve_out(args, args.bfile)
    dev = packed_t.device
    t0 = time.perf_counter()
    dense_models = args.lmm or args.fvlmm or args.lmm2
    k = None
    if dense_models and str(args.grm).lower().endswith((".spgrm", ".jxgrm")):
        raise SystemExit("-k FILE.spgrm is a sparse GRM: it serves -splmm only; -lmm / -lmm2 / -fvlmm need a dense GRM "
                         "(-k 1 | 2 | FILE.npy)")
    if dense_models and args.grm in ("1", "2"):
        k, eff, _ = pl.build_grm(packed_t, n_fam, int(args.grm), args.maf, args.geno)
        print(f"GRM method {args.grm}: eff_m={eff} ({time.perf_counter() - t0:.2f}s)")
    elif dense_models:
        k = torch.from_numpy(_load_grm(args.grm, fam)).to(dev)    # every rank reads the same file
    # -q N: the N leading principal components of the whole-cohort GRM as fixed-effect columns beside the intercept
    # (`load_or_build_q_with_cache` -> `build_pcs_from_grm`, python/janusx/assoc/workflow.py:3389-3422, 3577-3789: the last N
    # columns of the ascending eigendecomposition of the GRM without a ridge, stored as f32; the reference switches to a
    # randomised SVD of the genotypes above 15 000 samples, here the GRM route serves every size)
    qdim = _parse_qcov_dim(getattr(args, "qcov", "0"))
    qmat = None
    if qdim > 0:
        if qdim >= n_fam:
            raise SystemExit(f"Q/PC dimension out of range: {qdim}. valid=[0..{max(0, n_fam - 1)}]")
        kq = k
        if kq is None:
            kq, _eff_q, _ = pl.build_grm(packed_t, n_fam, int(args.grm) if args.grm in ("1", "2") else 1, args.maf, args.geno)
        tq = time.perf_counter()
        _s_all, ut_all = pl.eigh_from_grm(kq, ridge=0.0)
        qmat = ut_all[-qdim:].T.to(torch.float32).cpu().numpy().astype(np.float64)
        del _s_all, ut_all, kq
        print(f"Q matrix: {qdim} principal components of the GRM ({time.perf_counter() - tq:.2f}s)")
    sparse_path = None
    sparse_pos = None
    if args.splmm is not None:
        # SparseLMM, exact mode: thresholded sparse GRM of all genotyped samples once (`.spgrm`, or an existing one given
        # with -grm FILE.spgrm), then per trait the sparse REML null model and the exact g'Pg scan on its samples
        # -spk / --grm-sparse (python/janusx/assoc/workflow.py:6747-6754): 1 | 2 = method of the sparse GRM built here, or the
        # path of a precomputed .spgrm / .jxgrm; -k FILE.spgrm is this build's older way of naming the same file
        spk = str(getattr(args, "grm_sparse", "1") or "1").strip()
        spk_file = spk if spk.lower().endswith((".spgrm", ".jxgrm")) else None
        if spk_file is None and spk not in ("1", "2"):
            raise SystemExit(f"-spk {spk}: expected 1 (centering), 2 (standardization) or a .spgrm / .jxgrm file "
                             "(GCTA / fastGWA .grm.sp inputs are not read by this build)")
        if spk_file is not None or str(args.grm).lower().endswith((".spgrm", ".jxgrm")):
            sparse_path = spk_file if spk_file is not None else args.grm
            # the sparse GRM's own sample order: map by id, like the dense -grm FILE path (_load_grm)
            id_path = sparse_path + ".id"
            if not os.path.exists(id_path):
                raise SystemExit(f"{id_path} not found: a sparse GRM given with -k needs its sample-id file "
                                 "(one id per line, in the order of the matrix)")
            sparse_ids = [ln.split()[0] for ln in open(id_path) if ln.strip()]
            sparse_pos = {sid: i for i, sid in enumerate(sparse_ids)}
            if len(sparse_pos) != len(sparse_ids):
                raise SystemExit(f"{id_path} lists duplicate sample ids")
        else:
            method = int(spk) if spk == "2" else (int(args.grm) if args.grm in ("1", "2") else 1)
            sparse_path, _, nnz = jxrs.spgrm_bed_to_jxgrm(args.bfile, out_prefix=out, method=method,
                                                         threshold=float(args.splmm), maf_threshold=args.maf,
                                                         max_missing_rate=args.geno, het_threshold=0.0,
                                                         snps_only=bool(getattr(args, "snps_only", False)))
            if rank == 0:
                with open(sparse_path + ".id", "w") as fh:
                    for sid in fam:
                        fh.write(f"{sid}\n")
            print(f"Sparse GRM cutoff={args.splmm}: nnz={nnz} -> {sparse_path} ({time.perf_counter() - t0:.2f}s)")
    for ti in traits:
        name = names[ti]
        rows_ok = []
        for j, sid in enumerate(fam):
            i = pos.get(sid)
            if i is None or not math.isfinite(ph[i, ti]):
                continue
            if args.cov and (sid not in cpos or not np.all(np.isfinite(cv[cpos[sid]]))):
                continue
            rows_ok.append(j)
        keep_idx = np.array(rows_ok, dtype=np.int64)
        n = len(keep_idx)
        if n < 10:
            print(f"[{name}] only {n} phenotyped samples, skipped")
            continue
        y = np.array([ph[pos[fam[j]], ti] for j in keep_idx])
        x = np.ones((n, 1))
        if qmat is not None:                    # design = [1 | Q | C] (workflow.py:1545-1560)
            x = np.concatenate([x, qmat[keep_idx]], axis=1)
        if args.cov:
            x = np.concatenate([x, np.array([cv[cpos[fam[j]]] for j in keep_idx])], axis=1)
        for mode in (["lmm"] if args.lmm else []) + (["lmm2"] if args.lmm2 else []) + (["fvlmm"] if args.fvlmm else []):
            t1 = time.perf_counter()
            # rows are formatted and written block by block on a writer thread while the device scans the next block
            # (src/stats/lmm.rs:975-1477 with its AsyncTsvWriter, src/stats/common.rs:374)
            wr = {}

            def open_writer(keep_mask, af_k, miss_k, ncol, model_tag):
                kept_rows = np.nonzero(keep_mask)[0]
                wr["path"] = f"{out}.{name}.{model_tag}.tsv"
                wr["w"] = AsyncAssocTsvWriter(wr["path"], ncol, [bim.chrom[j] for j in kept_rows], [bim.pos[j] for j in kept_rows],
                                              [bim.snp[j] for j in kept_rows], [bim.a0[j] for j in kept_rows],
                                              [bim.a1[j] for j in kept_rows], af_k, miss_k, miss_count=(model_tag == "lm"))
                return wr["w"].put

            try:
                # `-lmm`: the reference's default scan -- warm-start chains over chunks of `-chunksize` rows of the file, seeded
                # with log10 lambda0 (workflow_model_stream.py:1436-1480; src/stats/lmm.rs:2627); JX_LMM_UNIFIED_NO_WARM_START=1
                # or `-warm-start none` give every SNP the same start (and the faster one-wave-per-SNP scan)
                chain = None
                if mode == "lmm" and _warm_start_mode(args) == "chain":
                    chain = (max(1, int(args.chunksize)), max(1, int(args.warm_chain_pieces)))
                res = pl.run_trait(packed_t, n_fam, k, keep_idx, y, x, mode, args.maf, args.geno, args.het,
                                   on_rows=open_writer, force_model=bool(args.force_model), warm_chain=chain)
            except BaseException:
                if "w" in wr:
                    wr["w"].abort()
                raise
            if "w" in wr:
                wr["w"].close()
            kept = np.nonzero(res.keep)[0]
            path = wr.get("path", f"{out}.{name}.{res.model_tag}.tsv")
            if res.model_tag == "lm":
                # LMM -> LM fallback (src/stats/gwas_unified.rs:121-175, workflow_model_stream.py:930-963)
                _sw, stat, pv = res.null_lrt
                print(f"[{name}] Warning: -{mode} switch to LM: null LRT stat={stat:.4g}, p={pv:.4g} (>=0.05); "
                      f"pve(null)={res.null.pve:.4f}")
                print(f"[{name}] -lm: n={n} snps={len(kept)} -> {path} ({time.perf_counter() - t1:.2f}s)")
            else:
                print(f"[{name}] -{mode}: n={n} snps={len(kept)} lambda0={res.null.lbd:.5g} pve={res.null.pve:.4f} "
                      f"-> {path} ({time.perf_counter() - t1:.2f}s)")
        if sparse_path is not None:
            from . import stats as st
            t1 = time.perf_counter()
            full = n == n_fam and np.array_equal(keep_idx, np.arange(n_fam))
            counts = jxrs.bed_row_counts(packed, n_fam, None if full else keep_idx)
            keep, af, miss = st.gwas_scan_row_stats(counts, n, args.maf, args.geno, args.het)
            kept = np.nonzero(keep)[0]
            maf_all = np.zeros(packed.shape[0], dtype=np.float32)
            maf_all[kept] = af[kept]
            grm_idx = None
            if sparse_pos is not None:
                missing_ids = [fam[j] for j in keep_idx if fam[j] not in sparse_pos]
                if missing_ids:
                    raise SystemExit(f"{sparse_path}.id lacks {len(missing_ids)} phenotyped sample(s), e.g. {missing_ids[0]}")
                grm_idx = np.array([sparse_pos[fam[j]] for j in keep_idx], dtype=np.int64)
            xc = x[:, 1:] if x.shape[1] > 1 else None
            sidx = keep_idx if (grm_idx is not None or not full) else None
            meta = ([bim.chrom[j] for j in kept], [bim.pos[j] for j in kept], [bim.snp[j] for j in kept],
                    [bim.a0[j] for j in kept], [bim.a1[j] for j in kept])
            for stem in sp_stems:
                t2 = time.perf_counter()
                path = f"{out}.{name}.{stem}.tsv"
                approx = stem == "splmm" and len(kept) >= SPLMM_APPROX_RHAT_MARKERS
                if stem == "splmm" and not approx:
                    # python/janusx/assoc/workflow_model_packed.py:8087-8107
                    print(f"[{name}] Warning: SparseLMM approx scan has fewer filtered markers than the default gamma sample size "
                          f"({len(kept)} < {SPLMM_APPROX_RHAT_MARKERS}); falling back to splmm-exact for this trait.")
                if approx:
                    # null: fastGWA fixed-Vp objective on the OLS residual (workflow_model_packed.py:3134-3156, 3430-3451), then
                    # the residualised GRAMMAR-gamma scan (`splmm_assoc_pcg_bed_to_tsv`, scan_mode "approx", :8527-8563)
                    xd = np.ones((n, 1)) if xc is None else np.concatenate([np.ones((n, 1)), xc], axis=1)
                    yc = y - xd @ np.linalg.solve(xd.T @ xd, xd.T @ y)
                    yc = yc - float(np.mean(yc))
                    vp = float(yc @ yc) / float(n - 1)
                    null = jxrs.spreml_sparse_fastgwa_fixed_vp_brent_from_jxgrm(sparse_path, yc, vp, sample_indices=grm_idx if grm_idx is not None else sidx,
                                                                               low=-5.0, high=5.0, grid_size=17, tol=1e-3, max_iter=20)
                    res = jxrs.splmm_assoc_pcg_bed_to_tsv(args.bfile, y, float(null[0]), *meta, path, x_cov=xc, sample_indices=sidx,
                                                          packed=packed, packed_n_samples=n_fam, maf=af[kept], row_flip=np.zeros(len(kept), bool),
                                                          row_missing=miss[kept], row_indices=kept, sparse_sample_indices=grm_idx,
                                                          sparse_jxgrm_path=sparse_path, rhat_markers=SPLMM_APPROX_RHAT_MARKERS,
                                                          rhat_seed=20260527, scan_mode="approx")
                    print(f"[{name}] -splmm: n={n} snps={len(kept)} lambda0={null[0]:.5g} sigma_g2={null[1]:.4g} sigma_e2={null[2]:.4g} "
                          f"gamma={res[0]:.6g} (markers used {res[8]}) -> {path} ({time.perf_counter() - t2:.2f}s)")
                else:
                    stats, l10, null = jxrs.splmm_exact_scan_from_jxgrm(
                        sparse_path, y, packed, n_fam, maf_all, np.zeros(packed.shape[0], dtype=bool), xc, sidx, kept,
                        grid_size=17, tol=1e-3, max_iter=20, grm_sample_indices=grm_idx)
                    if rank == 0:
                        write_assoc_tsv(path, *meta, af[kept], miss[kept], stats)
                    print(f"[{name}] -{'splmm-exact' if stem == 'splmm2' else 'splmm (exact scan)'}: n={n} snps={len(kept)} "
                          f"lambda0={null[0]:.5g} sigma_g2={null[1]:.4g} sigma_e2={null[2]:.4g} -> {path} "
                          f"({time.perf_counter() - t2:.2f}s)")
    return 0


BLUP_SMALL_N = 15000      # python/janusx/gs/blup.py:8-9
BLUP_SMALL_M = 15000


def resolve_blup_dispatch(n_samples, n_markers, force=None):
    """`resolve_blup_dispatch` (python/janusx/gs/blup.py:73-163) -> (effective method, rrBLUP solver or None).  `force` = the
    value of GS_BLUP ("0" GBLUP, "1" exact rrBLUP, "2" PCG rrBLUP; None / "" = automatic)."""
    force = (force or "").strip()
    if force not in ("", "0", "1", "2"):
        raise ValueError(f"Invalid GS_BLUP={force!r}; expected 0 (GBLUP), 1 (rrBLUP exact), or 2 (rrBLUP PCG).")
    if force == "0":
        return "GBLUP", None
    if force == "1":
        return "rrBLUP", "exact"
    if force == "2":
        return "rrBLUP", "pcg"
    if max(0, int(n_samples)) <= BLUP_SMALL_N:
        return "GBLUP", None
    return "rrBLUP", ("exact" if max(0, int(n_markers)) <= BLUP_SMALL_M else "pcg")


def cmd_gs(args):
    """`jx gs -BLUP`: centred GRM of all genotyped samples (once), then per trait a GBLUP fit on the phenotyped
    samples (spectral REML, src/stats/gblup.rs:1105-1240) -- K-fold cross-validated with `-cv` (folds of
    `build_cv_splits`, seeded by `-seed` = 42, python/janusx/gs/workflow.py:3950-3980, 18753-18760) -- and predictions for every genotyped sample."""
    from . import janusx as jxrs
    from .bed import read_fam_ids
    if args.rrblup:
        return cmd_gs_rrblup(args)
    if not (args.blup or args.gblup):
        raise SystemExit("select a model: -BLUP (or -GBLUP, -rrBLUP)")
    fam = read_fam_ids(args.bfile)
    ids, names, ph = _read_table(args.pheno)
    pos = {s: i for i, s in enumerate(ids)}
    traits = _select_traits(names, args.ncol)
    if args.blup and not args.gblup:
        # -BLUP = automatic dispatch (`resolve_blup_dispatch`, python/janusx/gs/blup.py:8-163): n_train <= 15000 -> GBLUP;
        # beyond that rrBLUP, exact (marker space) up to 15000 kept markers and PCG above; GS_BLUP=0/1/2 forces a route
        force = os.environ.get("GS_BLUP", "").strip()
        n_train_max = 0
        for ti in traits:
            n_train_max = max(n_train_max, sum(1 for sid in fam if sid in pos and math.isfinite(ph[pos[sid], ti])))
        try:
            method, _solver = resolve_blup_dispatch(n_train_max, 0, force)
        except ValueError as e:
            raise SystemExit(str(e))
        if method == "rrBLUP":
            # the kept-marker count is known only behind the QC of the rrBLUP route: its "auto" solver applies the same
            # 15000-marker rule (exact up to it, PCG above)
            args.rr_solver = {"1": "exact", "2": "pcg"}.get(force, "auto")
            print(f"-BLUP dispatch: n_train={n_train_max} -> rrBLUP ({args.rr_solver})"
                  + (f" (forced by GS_BLUP={force})" if force else ""))
            return cmd_gs_rrblup(args)
        print(f"-BLUP dispatch: n_train={n_train_max} -> GBLUP" + (" (forced by GS_BLUP=0)" if force == "0" else ""))
    out = _resolve_out(args, args.bfile)
    t0 = time.perf_counter()
    if args.grm:
        k = _load_grm(args.grm, fam)
        print(f"GRM loaded from {args.grm}: n={k.shape[0]}")
    else:
        k, eff, _ = jxrs.grm_stream_bed_f32(args.bfile, method=1, maf_threshold=args.maf, max_missing_rate=args.geno,
                                            het_threshold=0.0)
        print(f"GRM method 1: n={k.shape[0]} eff_m={eff} ({time.perf_counter() - t0:.2f}s)")
    n_all = len(fam)
    for ti in traits:
        name = names[ti]
        yv = np.array([ph[pos[s], ti] if s in pos else np.nan for s in fam])
        train = np.nonzero(np.isfinite(yv))[0].astype(np.int64)
        test = np.nonzero(~np.isfinite(yv))[0].astype(np.int64)
        if len(train) < 10:
            print(f"[{name}] only {len(train)} phenotyped samples, skipped")
            continue
        t1 = time.perf_counter()
        fold = np.full(n_all, -1, dtype=np.int64)
        pred = np.full(n_all, np.nan)
        if args.cv and args.cv > 1:
            print("Fold Method     Pearsonr Spearmanr R2")
            for f, (te_loc, tr_loc) in enumerate(build_cv_splits(len(train), args.cv, args.seed)):
                r = jxrs.gblup_reml_grm(k, train[tr_loc], yv[train[tr_loc]], train[te_loc], estimate_only=False)
                pred[train[te_loc]] = r[1].ravel()
                fold[train[te_loc]] = f
                pe, sp, r2f = cv_fold_metrics(yv[train[te_loc]], r[1].ravel())
                print(f"{f + 1:<4d} BLUP       {pe:.3f}    {sp:.3f}     {r2f:.3f}")
            yo, po = yv[train], pred[train]
            rr = float(np.corrcoef(yo, po)[0, 1])
            r2 = 1.0 - float(np.sum((yo - po) ** 2) / np.sum((yo - yo.mean()) ** 2))
            print(f"[{name}] GBLUP {args.cv}-fold CV: pearson={rr:.4f} R2={r2:.4f}")
        full = jxrs.gblup_reml_grm(k, train, yv[train], test if len(test) else None, return_variance_components=True)
        if not (args.cv and args.cv > 1):
            pred[train] = full[0].ravel()
        if len(test):
            pred[test] = full[1].ravel()
        path = f"{out}.{name}.gs.GBLUP.tsv"
        tmp = f"{path}.tmp.{os.getpid()}"
        with open(tmp, "w") as fh:
            fh.write("sample\tobserved\tpredicted\tfold\n")
            for j, sid in enumerate(fam):
                obs = "NA" if not math.isfinite(yv[j]) else f"{yv[j]:.6g}"
                fh.write(f"{sid}\t{obs}\t{pred[j]:.6g}\t{'NA' if fold[j] < 0 else fold[j]}\n")
        os.replace(tmp, path)
        print(f"[{name}] GBLUP: n_train={len(train)} n_pred={len(test)} lambda={full[3]:.5g} pve={full[2]:.4f} "
              f"sigma_g2={full[9]:.5g} sigma_e2={full[10]:.5g} -> {path} ({time.perf_counter() - t1:.2f}s)")
    return 0


def cmd_gs_rrblup(args):
    """`jx gs -rrBLUP`: marker effects of the standardised genotypes, by the exact marker-space route
    (`rrblup_exact_snp_packed`, src/stats/rrblup.rs:3179-3490) or by PCG over the packed payload
    (`rrblup_pcg_bed`, src/stats/rrblup.rs:3494-4307; the route python/janusx/gs/blup.py:146-163 takes for large n
    and m).  lambda (equation scale, sigma_e^2 / sigma_beta^2) is `-lambda` when given; otherwise
    m_effective * sigma_e^2 / sigma_g^2 from the spectral GBLUP REML of a random subsample of at most 2000 training
    samples (seed `-seed`) on the standardised GRM -- the GRM-REML branch of the reference's
    `_estimate_rrblup_lambda_subsample_reml` (python/janusx/gs/workflow.py:5481) without its HE variants."""
    from . import janusx as jxrs
    from .bed import read_fam_ids
    fam = read_fam_ids(args.bfile)
    ids, names, ph = _read_table(args.pheno)
    pos = {s: i for i, s in enumerate(ids)}
    traits = _select_traits(names, args.ncol)
    out = _resolve_out(args, args.bfile)
    packed, miss, maf, _std, n_all = jxrs.load_bed_2bit_packed(args.bfile)
    flip = jxrs.bed_packed_row_flip_mask(packed, n_all)
    keep = (maf >= np.float32(args.maf)) & (miss <= np.float32(args.geno))
    # solver choice of the reference (`_resolve_rrblup_solver` + `_resolve_rrblup_exact_backend`, python/janusx/gs/workflow.py:
    # 5230-5276): up to 15 000 markers the exact marker-space route ("snp": REML lambda from the m x m spectrum); beyond that
    # the exact SAMPLE-space route ("fast": REML on the spectrum of the n x n kernel of the standardised markers) while the
    # phenotyped samples number at most 10 000, PCG above
    solver = args.rr_solver
    if solver in ("exact", "fast") and args.lam is not None:
        raise SystemExit("-lambda needs -rr-solver pcg: the exact routes estimate lambda by REML on the spectrum and would "
                         "ignore the given value")
    n_pheno_max = max((int(np.isfinite([ph[pos[s], ti] if s in pos else np.nan for s in fam]).sum()) for ti in traits), default=0)
    if solver == "auto":
        # a given -lambda only exists on the PCG route (the exact routes re-estimate it): honour it
        if args.lam is not None:
            solver = "pcg"
        elif int(keep.sum()) <= 15000:
            solver = "exact"
        else:
            solver = "fast" if n_pheno_max <= 10000 else "pcg"
    print(f"rrBLUP-{solver.upper()}: n={n_all} m={packed.shape[0]} kept={int(keep.sum())} (maf {args.maf}, geno {args.geno})")
    k_std = None
    if solver == "fast":
        # sample-space exact route (the reference's "fast" backend: pyBLUP.BLUP on the implicit kinship of the standardised
        # markers, python/janusx/pyBLUP/mlm.py): ridge regression on Z is GBLUP on K = Z'Z / m_eff with lambda_equation =
        # m_eff * sigma_e^2 / sigma_g^2.  K over ALL genotyped samples once (global allele frequencies, like the marker-space
        # routes), then per fit eigendecomposition of K[train, train] + Brent on its spectrum + K[test, train] alpha
        # (`gblup_reml_grm`: the kernels of the `-BLUP` branch)
        p_std = np.clip(maf[keep], 0.0, 0.5)
        m_eff_std = int(np.count_nonzero(2.0 * p_std * (1.0 - p_std) > 1e-12))
        t0k = time.perf_counter()
        k_std = jxrs.grm_packed_f32(np.ascontiguousarray(packed[keep]), n_all, flip[keep], maf[keep], None, method=2)
        print(f"standardised kernel of {n_all} samples over {m_eff_std} markers ({time.perf_counter() - t0k:.2f}s)")
    pcg_payload = packed
    if solver == "pcg":
        # the PCG route streams the payload from HBM for the whole solve: upload it ONCE and hand the device tensor to both
        # he_pcg_bed and rrblup_pcg_bed (their images of the training payload are shared inside `pcg_image_scope`)
        try:
            import warnings
            import torch
            if torch.cuda.is_available() and 3.2 * packed.nbytes < torch.cuda.mem_get_info()[0]:
                with warnings.catch_warnings():
                    # a read-only view of the mapped file: it is only read (copied to the device), never written through
                    warnings.filterwarnings("ignore", message="The given NumPy array is not writable")
                    pcg_payload = torch.from_numpy(np.ascontiguousarray(packed)).cuda()
        except Exception:   # noqa: BLE001 - the host array works everywhere (uploaded per call)
            pcg_payload = packed
    for ti in traits:
        name = names[ti]
        yv = np.array([ph[pos[s], ti] if s in pos else np.nan for s in fam])
        train = np.nonzero(np.isfinite(yv))[0].astype(np.int64)
        test = np.nonzero(~np.isfinite(yv))[0].astype(np.int64)
        if len(train) < 10:
            print(f"[{name}] only {len(train)} phenotyped samples, skipped")
            continue
        t1 = time.perf_counter()
        scope = jxrs.pcg_image_scope() if solver == "pcg" else contextlib.nullcontext()
        with scope:   # the images are released when the block is left, also by an exception
            if solver in ("exact", "fast"):
                lam, src = None, "REML on the spectrum"
            elif args.lam is not None:
                lam, src = float(args.lam), "manual"
            elif not args.lambda_reml:
                # Haseman-Elston first (python/janusx/gs/workflow.py:5564 `he_first`): lambda_equation = lambda_k * m_effective
                try:
                    he = jxrs.he_pcg_bed("", train, yv[train], site_keep=keep, seed=args.seed if args.seed != 42 else 20260512,
                                         packed=pcg_payload, packed_n_samples=n_all, maf=maf, row_flip=flip)
                except RuntimeError as e:   # e.g. the stochastic traces violate the PSD bound on a small panel
                    he = None
                    lam, src = None, f"HE failed ({e})"
                if he is not None:
                    if np.isfinite(he[10]) and he[10] >= 0.0:
                        lam, src = max(1e-8, float(he[10]) * float(max(1, he[6]))), f"HE (h2={he[2]:.4f}, lambda_k={he[10]:.5g})"
                    else:
                        lam, src = None, "HE on the boundary"
            else:
                lam, src = None, ""
            if lam is None and solver == "pcg":
                rng = np.random.default_rng(args.seed)
                sub = np.sort(rng.permutation(len(train))[:min(len(train), 2000)])
                ks = jxrs.grm_packed_f32(np.ascontiguousarray(packed[keep]), n_all, flip[keep], maf[keep], train[sub], method=2)
                fit = jxrs.gblup_reml_grm(ks, np.arange(len(sub), dtype=np.int64), yv[train[sub]], None,
                                          return_variance_components=True)
                lam_k = float(fit[3])
                p = np.clip(maf[keep], 0.0, 0.5)
                m_eff = int(np.count_nonzero(2.0 * p * (1.0 - p) > 1e-12))
                lam, src = lam_k * m_eff, (src + " -> " if src else "") + f"subsample REML (n_sub={len(sub)}, lambda_k={lam_k:.5g})"
            fold = np.full(n_all, -1, dtype=np.int64)
            pred = np.full(n_all, np.nan)

            def fit_predict(tr, te):
                if solver == "fast":
                    r = jxrs.gblup_reml_grm(k_std, tr, yv[tr], te if len(te) else None, return_variance_components=True)
                    # (pred_train, pred_test, pve, lambda_k, ml, reml, ..., sigma_g2, sigma_e2): same slots as the marker-space fit
                    return (r[0], r[1], r[2], float(r[3]) * float(m_eff_std), r[5], (r[9], r[10]), m_eff_std)
                if solver == "exact":
                    return jxrs.rrblup_exact_snp_packed(packed, n_all, tr, yv[tr], te if len(te) else None, site_keep=keep,
                                                        maf=maf, row_flip=flip)
                return jxrs.rrblup_pcg_bed("", tr, yv[tr], te if len(te) else None, site_keep=keep, lambda_value=lam,
                                           tol=args.tol, max_iter=args.max_iter, packed=pcg_payload, packed_n_samples=n_all,
                                           maf=maf, row_flip=flip)

            if args.cv and args.cv > 1:
                print("Fold Method     Pearsonr Spearmanr R2")
                for f, (te_loc, tr_loc) in enumerate(build_cv_splits(len(train), args.cv, args.seed)):
                    r = fit_predict(train[tr_loc], train[te_loc])
                    pred[train[te_loc]] = r[1].ravel()
                    fold[train[te_loc]] = f
                    pe, sp, r2f = cv_fold_metrics(yv[train[te_loc]], r[1].ravel())
                    print(f"{f + 1:<4d} rrBLUP     {pe:.3f}    {sp:.3f}     {r2f:.3f}")
                yo, po = yv[train], pred[train]
                rr = float(np.corrcoef(yo, po)[0, 1])
                r2 = 1.0 - float(np.sum((yo - po) ** 2) / np.sum((yo - yo.mean()) ** 2))
                print(f"[{name}] rrBLUP {args.cv}-fold CV: pearson={rr:.4f} R2={r2:.4f}")
            full = fit_predict(train, test)
        if not (args.cv and args.cv > 1):
            pred[train] = full[0].ravel()
        if len(test):
            pred[test] = full[1].ravel()
        path = f"{out}.{name}.gs.rrBLUP.tsv"
        tmp = f"{path}.tmp.{os.getpid()}"
        with open(tmp, "w") as fh:
            fh.write("sample\tobserved\tpredicted\tfold\n")
            for j, sid in enumerate(fam):
                obs = "NA" if not math.isfinite(yv[j]) else f"{yv[j]:.6g}"
                fh.write(f"{sid}\t{obs}\t{pred[j]:.6g}\t{'NA' if fold[j] < 0 else fold[j]}\n")
        os.replace(tmp, path)
        if solver in ("exact", "fast"):
            print(f"[{name}] rrBLUP-{'EXACT' if solver == 'exact' else 'FAST (sample space)'}: n_train={len(train)} n_pred={len(test)} lambda={full[3]:.5g} [{src}] "
                  f"reml={full[4]:.6g} var_g={full[5][0]:.5g} sigma_e2={full[5][1]:.5g} m_effective={full[6]} "
                  f"pve={full[2]:.4f} -> {path} ({time.perf_counter() - t1:.2f}s)")
        else:
            print(f"[{name}] rrBLUP-PCG: n_train={len(train)} n_pred={len(test)} lambda={lam:.5g} [{src}] "
                  f"converged={full[3]} iters={full[4]} rel_res={full[5]:.3g} m_effective={full[6]} "
                  f"pve={full[7]:.4f} -> {path} ({time.perf_counter() - t1:.2f}s)")
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="jx", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    g = sub.add_parser("gwas")
    g.add_argument("-bfile", "--bfile", required=True)
    g.add_argument("-p", "--pheno", required=True)
    g.add_argument("-n", "--n", dest="ncol", action="append", default=None)
    g.add_argument("-lmm", "--lmm", action="store_true", default=False)
    g.add_argument("-lmm2", "--lmm2", action="store_true", default=False)
    g.add_argument("-fvlmm", "--fvlmm", action="store_true", default=False)
    g.add_argument("-k", "--grm", dest="grm", type=str, default="1")
    g.add_argument("-c", "--cov", dest="cov", action="append", default=None,
                   help="covariate table (sample id + columns) or a SNP site chr:pos taken as a covariate; may be repeated")
    g.add_argument("-snps-only", "--snps-only", dest="snps_only", action="store_true", default=False,
                   help="drop sites whose alleles are not single A/C/G/T")
    g.add_argument("-mem", "--memory", dest="memory", type=float, default=None,
                   help="accepted for compatibility; unused (the working set lives in HBM)")
    g.add_argument("-v", "--verbose", action="store_true", default=False, help="accepted for compatibility")
    g.add_argument("-q", "--qcov", dest="qcov", type=str, default="0",
                   help="number of principal components of the GRM added as covariates (integer >= 0; external covariates "
                        "go through -c)")
    g.add_argument("-mmap-window-mb", "--mmap-window-mb", dest="mmap_window_mb", type=int, default=None,
                   help="stage the .bed payload to the device in windows of this many MiB (default 256)")
    g.add_argument("-maf", "--maf", type=float, default=0.02)
    g.add_argument("-geno", "--geno", type=float, default=0.05)
    g.add_argument("-het", "--het", type=float, default=1.0)
    g.add_argument("-o", "--out", default=None)
    g.add_argument("-prefix", "--prefix", default=None, help="file name prefix inside the -o directory")
    g.add_argument("-force-model", "--force-model", dest="force_model", action="store_true", default=False)
    g.add_argument("-chunksize", "--chunksize", dest="chunksize", type=int, default=10000,
                   help="SNP rows per scan chunk (the reference's default 10000): the length of a warm-start chain of -lmm")
    g.add_argument("-warm-start", "--warm-start", dest="warm_start", choices=["chain", "none"], default=None,
                   help="-lmm: 'chain' (the reference's default: each SNP's Brent starts from the previous SNP's optimum inside "
                        "a chunk) or 'none'; default chain unless JX_LMM_UNIFIED_NO_WARM_START is set")
    g.add_argument("-warm-chain-pieces", "--warm-chain-pieces", dest="warm_chain_pieces", type=int, default=1,
                   help="cut every chunk's chain into this many pieces (power of two) by halving, as rayon's splitter does on "
                        "the reference's thread pool (2 x threads pieces); 1 = one chain per chunk")
    g.add_argument("-t", "--thread", type=int, default=0, help="accepted for compatibility; unused")
    g.set_defaults(func=cmd_gwas)
    g.add_argument("-splmm", "--splmm", "-splmm-approx", "--splmm-approx", dest="splmm", nargs="?", const=0.05, default=None,
                   type=float,
                   help="SparseLMM with the GRAMMAR-gamma scan approximation (fastGWA null + residualised scan) on a sparse GRM "
                        "thresholded at this kinship cut-off (default 0.05; negative: keep every entry); -k FILE.spgrm reuses "
                        "an existing sparse GRM")
    g.add_argument("-spk", "--grm-sparse", dest="grm_sparse", type=str, default="1",
                   help="sparse GRM of the SparseLMM models: 1 (centering), 2 (standardization) or a precomputed .spgrm / .jxgrm "
                        "file with its .id sibling")
    g.add_argument("-splmm-exact", "--splmm-exact", dest="splmm_exact", nargs="?", const=0.05, default=None, type=float,
                   help="SparseLMM with the exact g'Pg denominator for every SNP -> {out}.{trait}.splmm2.tsv")
    r = sub.add_parser("grm")
    r.add_argument("-bfile", "--bfile", default=None)
    r.add_argument("-m", "--method", type=int, default=1, choices=[1, 2])
    r.add_argument("-maf", "--maf", type=float, default=0.02)
    r.add_argument("-geno", "--geno", type=float, default=0.05)
    r.add_argument("-o", "--out", default=None)
    r.add_argument("-prefix", "--prefix", default=None, help="file name prefix inside the -o directory")
    r.add_argument("-grm", "--grm", "-k", "--dense-grm", dest="grm", default=None,
                   help="existing dense GRM (.npy with a sibling .id); with -sparse it is thresholded into a .spgrm "
                        "(-k / --dense-grm is the reference's name of this option)")
    r.add_argument("-snps-only", "--snps-only", dest="snps_only", action="store_true", default=False,
                   help="drop sites whose alleles are not single A/C/G/T")
    r.add_argument("-mem", "--memory", dest="memory", type=float, default=None, help="accepted for compatibility; unused")
    r.add_argument("-v", "--verbose", action="store_true", default=False, help="accepted for compatibility")
    r.add_argument("-txt", "--txt", action="store_true", default=False,
                   help="write the dense GRM as plain text ({out}.cGRM.txt, %%.6f) instead of NPY")
    r.add_argument("-sparse", "--sparse", nargs="?", const=0.05, default=None, type=float,
                   help="write a sparse `.spgrm` keeping off-diagonal kinship > cutoff (negative: keep everything)")
    r.add_argument("-t", "--thread", type=int, default=0, help="accepted for compatibility; unused")
    r.set_defaults(func=cmd_grm)
    q = sub.add_parser("gs")
    q.add_argument("-bfile", "--bfile", required=True)
    q.add_argument("-p", "--pheno", required=True)
    q.add_argument("-n", "--n", dest="ncol", action="append", default=None)
    q.add_argument("-BLUP", "--BLUP", dest="blup", action="store_true", default=False)
    q.add_argument("-GBLUP", "--GBLUP", dest="gblup", action="store_true", default=False)
    q.add_argument("-rrBLUP", "--rrBLUP", dest="rrblup", action="store_true", default=False)
    q.add_argument("-rr-solver", "--rr-solver", "--rrblup-solver", dest="rr_solver", choices=["auto", "exact", "fast", "pcg"], default="auto",
                   help="rrBLUP: exact marker-space route (auto up to 15000 kept markers) or PCG")
    q.add_argument("-lambda", "--lambda", "--rrblup-lambda", dest="lam", type=float, default=None)
    q.add_argument("-lambda-reml", "--lambda-reml", dest="lambda_reml", action="store_true", default=False,
                   help="rrBLUP: take lambda from the subsample GBLUP REML instead of Haseman-Elston")
    q.add_argument("-tol", "--tol", type=float, default=1e-4)
    q.add_argument("-max-iter", "--max-iter", dest="max_iter", type=int, default=100)
    q.add_argument("-cv", "--cv", type=int, default=None)
    q.add_argument("-seed", "--seed", type=int, default=42)
    q.add_argument("-k", "--grm", dest="grm", type=str, default=None)
    q.add_argument("-maf", "--maf", type=float, default=0.02)
    q.add_argument("-geno", "--geno", type=float, default=0.05)
    q.add_argument("-o", "--out", default=None)
    q.add_argument("-prefix", "--prefix", default=None, help="file name prefix inside the -o directory")
    q.add_argument("-t", "--thread", type=int, default=0, help="accepted for compatibility; unused")
    q.set_defaults(func=cmd_gs)
    args = ap.parse_args(argv)
    return args.func(args)


if __name__ == "__main__":
    sys.exit(main())
