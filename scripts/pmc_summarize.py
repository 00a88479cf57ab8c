"""Aggregate a rocprofv3 --pmc run (counter_collection.csv files under a directory) into per-kernel sums.
usage: pmc_summarize.py <rocprof output dir> <out.json> [note]"""
import csv, glob, json, os, sys

def main():
    root, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    agg = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            c = r["Counter_Name"]
            d = agg.setdefault(k, {}).setdefault(c, {"dispatches": set(), "sum": 0.0})
            d["dispatches"].add(r["Dispatch_Id"])
            d["sum"] += float(r["Counter_Value"])
    res = {"note": note, "kernels": {}}
    for k, cs in agg.items():
        res["kernels"][k] = {c: {"calls": len(v["dispatches"]), "sum": v["sum"], "mean": v["sum"] / max(1, len(v["dispatches"]))}
                             for c, v in cs.items()}
    json.dump(res, open(out, "w"), indent=1)
    for k in sorted(res["kernels"]):
        if "grm_f16x2" in k or "rotate_f16x2" in k or "scan" in k:
            print(k, {c: round(v["mean"], 1) for c, v in res["kernels"][k].items()})

main()
