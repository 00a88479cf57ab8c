"""Timeline of the divide and conquer from a rocprofv3 kernel trace: window from the end of the bulge chasing (sb2st_owned_kernel) to the
first sbback_vu_kernel of the LAST decomposition: wall time, union busy time, idle gaps, per-kernel totals."""
import csv, glob, os, sys, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ch = [i for i, r in enumerate(rows) if "sb2st_owned" in r["Kernel_Name"]][-1]
vu = [i for i, r in enumerate(rows) if "sbback_vu" in r["Kernel_Name"] and i > ch][0]
t0 = int(rows[ch]["End_Timestamp"]); t1 = int(rows[vu]["Start_Timestamp"])
win = rows[ch + 1:vu]
print("window %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(win)))
ev = []
for r in win: ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort(); lvl = 0; last = t0; idle = 0; gaps = []
for t, d in ev:
    if lvl == 0 and t > last: idle += t - last; gaps.append(t - last)
    last = max(last, t); lvl += d
idle += max(0, t1 - last)
print("idle %.1f ms; gaps > 100 us: %d (sum %.1f ms); > 20 us: %d (sum %.1f ms)" % (idle / 1e6, sum(g > 1e5 for g in gaps), sum(g for g in gaps if g > 1e5) / 1e6,
      sum(g > 2e4 for g in gaps), sum(g for g in gaps if g > 2e4) / 1e6))
tot = collections.Counter(); cnt = collections.Counter()
for r in win:
    k = r["Kernel_Name"].replace("void ", "").replace("jx::", "")[:44]
    tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
for k, v in tot.most_common(16): print("  %-46s %5d %8.2f ms" % (k, cnt[k], v / 1e6))
big = sorted(gaps, reverse=True)[:12]
print("largest gaps (us):", [round(g / 1e3) for g in big])
