"""Compact view of bench.py JSON lines: python scripts/show_bench.py file.json [file.json ...]"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    print("value %.1f %s  ms/step %.1f  n_gpus %d  %s" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["config"]["workload"][:70]))
    print("stages", {k: round(v, 1) for k, v in d["stages_ms_per_step"].items()})
    for k in ("roofline", "roofline_grm", "roofline_rotate", "roofline_scan"):
        r = d.get(k)
        if r:
            print(k, r.get("kernel", "")[:40], "achieved %.1f %s frac %.3f" % (r["achieved"], r["unit"], r["frac"]), "traffic", r.get("traffic"))
    if "cpu_baseline" in d:
        print("cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cores"))
