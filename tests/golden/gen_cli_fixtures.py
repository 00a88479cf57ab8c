#!/usr/bin/env python3
"""Reference-produced values for the command-line helpers the mirror restates (janusx_amd/cli.py).

Run in the BUILD container only:  python tests/golden/gen_cli_fixtures.py
The reference's pure-Python helpers are imported from /root/reference/python with a stub `janusx.janusx` module (the native
extension cannot be built here) and evaluated on a list of inputs; inputs and outputs are stored as data in
tests/golden/cli_helpers.json.  Nothing under /root/reference is copied.
  python/janusx/gs/blup.py::resolve_blup_dispatch (+ the GS_BLUP override), python/janusx/assoc/workflow.py::
  _parse_cov_site_token, _parse_qcov_dim, _canon_site_key, _GWAS_PCA_GRM_EIGH_SAMPLE_THRESHOLD; python/janusx/gs/workflow.py::
  build_cv_splits."""
import importlib
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PY = "/root/reference/python"


def main():
    sys.path.insert(0, REF_PY)

    class _Stub(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return None

    sys.modules["janusx.janusx"] = _Stub("janusx.janusx")
    blup = importlib.import_module("janusx.gs.blup")
    wf = importlib.import_module("janusx.assoc.workflow")
    out = {"blup_small_n": int(blup.BLUP_SMALL_N), "blup_small_m": int(blup.BLUP_SMALL_M),
           "pca_grm_eigh_threshold": int(wf._GWAS_PCA_GRM_EIGH_SAMPLE_THRESHOLD), "dispatch": [], "cov_site": [], "qcov": [],
           "canon": []}
    for force in ("", "0", "1", "2"):
        if force:
            os.environ["GS_BLUP"] = force
        else:
            os.environ.pop("GS_BLUP", None)
        for n in (10, 14999, 15000, 15001, 60000):
            for m in (100, 15000, 15001, 900000):
                d = blup.resolve_blup_dispatch(n, m)
                out["dispatch"].append({"force": force, "n": n, "m": m, "method": d.effective_method,
                                        "solver": d.rrblup_solver})
    os.environ.pop("GS_BLUP", None)
    for tok in ("1:58", "chr1:58:58", "X:1e3", "7：123", "cov.tsv", "a:b", "1:5:9", "1:0", ":5", "1:5:x", "1:2:3:4", "2:7.0:7"):
        try:
            r = wf._parse_cov_site_token(tok)
            out["cov_site"].append({"token": tok, "result": None if r is None else [r[0], int(r[1])], "error": None})
        except Exception as e:
            out["cov_site"].append({"token": tok, "result": None, "error": type(e).__name__})
    for q in ("0", "3", " 12 ", "-1", "", "pcs.txt", "2.5"):
        try:
            out["qcov"].append({"value": q, "result": int(wf._parse_qcov_dim(q)), "error": None})
        except Exception as e:
            out["qcov"].append({"value": q, "result": None, "error": type(e).__name__})
    for c, p in (("Chr7", 12), ("chr01", 5), ("X", 9), (" 3 ", 1)):
        k = wf._canon_site_key(c, p)
        out["canon"].append({"chrom": c, "pos": p, "key": [k[0], int(k[1])]})
    # K-fold splits of `jx gs -cv` (python/janusx/gs/workflow.py::build_cv_splits over pyBLUP/kfold.py::KFold)
    gsw = importlib.import_module("janusx.gs.workflow")
    out["cv_splits"] = []
    for n, k, seed in ((23, 4, 7), (10, 5, 42), (101, 3, 0)):
        sp = gsw.build_cv_splits(n, k, seed)
        out["cv_splits"].append({"n": n, "k": k, "seed": seed,
                                 "folds": [[[int(v) for v in te], [int(v) for v in tr]] for te, tr in sp]})
    with open(os.path.join(HERE, "cli_helpers.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "cli_helpers.json"), {k: (len(v) if isinstance(v, list) else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
