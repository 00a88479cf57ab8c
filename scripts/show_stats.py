"""Print the head of a rocprofv3 kernel_stats.csv compactly: python scripts/show_stats.py <dir or csv> [rows]"""
import csv, glob, os, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].replace("void ", "").replace("jx::", "")
    print(name[:64].ljust(66), r["Calls"].rjust(7), "%10.2f ms" % (float(r["TotalDurationNs"]) / 1e6), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"].rjust(6))
