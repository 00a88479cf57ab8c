"""Per-launch durations of the two tridiagonalisation kernels by column bucket, from a rocprofv3 kernel trace:
    python scripts/prof_update_cols.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for key in ("sytrd_update_kernel", "sytrd_symv_kernel"):
    d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if key in r["Kernel_Name"]]
    d = d[-4999:]                                    # last decomposition of the run
    dur = [b - a for a, b in d]
    gap = [d[k + 1][0] - d[k][1] for k in range(len(d) - 1)]
    print(key, "launches", len(d))
    nb = 10
    for b in range(nb):
        lo, hi = b * len(dur) // nb, (b + 1) * len(dur) // nb
        seg = sorted(dur[lo:hi])
        print(f"  columns {lo:5d}-{hi:5d}: mean {sum(seg) / len(seg) / 1e3:6.2f} us  median {seg[len(seg) // 2] / 1e3:6.2f}  "
              f"p10 {seg[len(seg) // 10] / 1e3:6.2f}  p90 {seg[9 * len(seg) // 10] / 1e3:6.2f}")
    # by position inside the 64-column panel
    for i in (0, 1, 8, 31, 62, 63):
        seg = [dur[k] for k in range(len(dur)) if k % 64 == i and k < 2000]
        if seg:
            print(f"  panel column {i:2d} (first 2000 columns): mean {sum(seg) / len(seg) / 1e3:6.2f} us")
