"""Timeline of the band reduction from a rocprofv3 kernel trace: python scripts/trace_band.py <dir with *_kernel_trace.csv>
Prints, for the window from the first panel kernel to the band extraction of the LAST decomposition in the trace: wall time,
busy time per queue, idle gaps of the busiest queue, and per-kernel totals inside the window."""
import csv, glob, os, sys, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
def nm(r): return r["Kernel_Name"].replace("void ", "").replace("jx::", "")[:40]
ends = [i for i, r in enumerate(rows) if "sb_extract_band" in r["Kernel_Name"]]
last = ends[-1]
prev = ends[-2] if len(ends) > 1 else -1
win = rows[prev + 1:last + 1]
first = next(i for i, r in enumerate(win) if "sb_chol" in r["Kernel_Name"] or "dgemm_kernel" in r["Kernel_Name"])
win = win[first:]
t0 = min(int(r["Start_Timestamp"]) for r in win); t1 = max(int(r["End_Timestamp"]) for r in win)
print("window %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(win)))
byq = collections.defaultdict(list)
for r in win: byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print("queue", q, "kernels", len(rs), "busy %.1f ms" % (busy / 1e6))
tot = collections.Counter(); cnt = collections.Counter()
for r in win:
    tot[nm(r)] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[nm(r)] += 1
for k, v in tot.most_common(14): print("  %-42s %5d %8.1f ms %8.1f us" % (k, cnt[k], v / 1e6, v / 1e3 / cnt[k]))
# union busy time over all queues and the time with exactly one / two queues busy
ev = []
for r in win: ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort(); lvl = 0; lastt = t0; hist = collections.Counter()
for t, d in ev:
    hist[lvl] += t - lastt; lastt = t; lvl += d
print("time with k kernels running:", {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
# per-panel sample: kernels between two consecutive sb_recon launches in the middle of the window
rec = [i for i, r in enumerate(win) if "sb_recon" in r["Kernel_Name"]]
for mid in (len(rec) // 8, len(rec) // 2, 7 * len(rec) // 8):
    a, b = rec[mid], rec[mid + 1]
    base = int(win[a]["Start_Timestamp"])
    print("--- panel", mid, "span %.0f us" % ((int(win[b]["Start_Timestamp"]) - base) / 1e3))
    for r in win[a:b + 1]:
        print("   q%s %-40s start %7.0f dur %6.0f" % (r["Queue_Id"], nm(r), (int(r["Start_Timestamp"]) - base) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
