"""Turns the round-end collection of scripts/collect_profiles.sh (gpurun_out/prof_end) into the committed summaries:
profiles/<tag>_bench_c2_lmm_kernel_stats.csv, <tag>_pmc_hbm_traffic.json, <tag>_pmc_mfma.json.
usage: assemble_profiles.py <prof_end dir> <tag> "<state note>" [n m config-label]   (default 20000 200000 c3)"""
import csv, glob, json, os, sys


def main():
    src, tag, note = sys.argv[1], sys.argv[2], sys.argv[3]
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
    m = int(sys.argv[5]) if len(sys.argv) > 5 else 200000
    label = sys.argv[6] if len(sys.argv) > 6 else "c3"
    shape = {"n": n, "m": m}
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    # kernel stats
    cand = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if cand:
        cand.sort(key=os.path.getmtime)                    # gpurun merges into gpurun_out: older collections stay beside the new one
        rows = list(csv.DictReader(open(cand[-1])))
        out = os.path.join(root, f"{tag}_bench_{label}_lmm_kernel_stats.csv")
        with open(out, "w") as fh:
            fh.write(f'"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 '
                     f'--no-cpu-baseline --no-extra (MI355X, {note}; n={n} m={m} -lmm; 3 pipeline passes incl. warmup)"\n')
            fh.write("kernel,calls,total_ns,avg_ns,pct\n")
            for r in rows:
                fh.write('"%s",%s,%s,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
        print("wrote", out, len(rows), "kernels")
    runs = {}
    for run in ("fetch", "write", "fetch_fv"):
        path = os.path.join(src, f"{run}.json")
        if not os.path.exists(path):
            continue
        ks = json.load(open(path))["kernels"]
        cname = "WRITE_SIZE" if run == "write" else "FETCH_SIZE"
        runs[run] = {k: {"calls": v[cname]["calls"], "mean_KB": v[cname]["mean"]} for k, v in ks.items()
                     if cname in v and "jx::" in k}
    if runs:
        out = os.path.join(root, f"{tag}_pmc_hbm_traffic.json")
        json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (each in its own pass) -- python3 bench.py --steps 1 "
                           f"--warmup 0 --no-cpu-baseline --no-extra [--mode fvlmm for fetch_fv]; MI355X, {note} (n={n} m={m}); unit "
                           "KB as reported; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide streaming reads by 2x on "
                           "gfx950 (double before comparing with bytes). Per-kernel means over dispatches "
                           "(scripts/pmc_summarize.py).", "shape": shape, "runs": runs}, open(out, "w"), indent=1)
        print("wrote", out, {k: len(v) for k, v in runs.items()})
    for fname, kind, extra in (("mfma.json", "mfma", ""), ("mfma_missing.json", "mfma_missing1pct", " --missing 0.01")):
        path = os.path.join(src, fname)
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        d["note"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_I8 "
                     f"GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra{extra}; MI355X, {note} (n={n} "
                     f"m={m} -lmm). MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 "
                     "SIMDs). Per-kernel means over dispatches.")
        d["shape"] = shape
        # keep the summary small: only this library's kernels
        d["kernels"] = {k: v for k, v in d["kernels"].items() if "jx::" in k}
        out = os.path.join(root, f"{tag}_pmc_{kind}.json")
        json.dump(d, open(out, "w"), indent=1)
        print("wrote", out)


main()
