"""Timeline of the LAST decomposition in a rocprofv3 --kernel-trace CSV of scripts/time_eigh.py: per stage window (band reduction /
bulge chasing / divide and conquer / Q2 / Q1, cut at the first and last launch of the stage's marker kernels) the busy time per
kernel, the union of busy intervals and the idle time.  usage: trace_q1_stage.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["k"] = r["Kernel_Name"].split("(")[0].replace("void jx::", "").replace("jx::", "")
rows.sort(key=lambda r: r["s"])
q2 = [i for i, r in enumerate(rows) if "sbback_apply" in r["k"]]
# last decomposition: the last run of sbback_apply launches
last_q2_end = q2[-1]
i = len(q2) - 1
while i > 0 and q2[i] - q2[i - 1] < 50:
    i -= 1
first_q2 = q2[i]
def window(a, b, title):
    seg = [r for r in rows if r["s"] >= a and r["e"] <= b]
    if not seg: return
    busy = collections.Counter(); cnt = collections.Counter()
    for r in seg:
        busy[r["k"]] += r["e"] - r["s"]; cnt[r["k"]] += 1
    ivs = sorted((r["s"], r["e"]) for r in seg)
    union = 0; cs, ce = ivs[0]
    for s, e in ivs[1:]:
        if s > ce: union += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    union += ce - cs
    span = b - a
    print(f"== {title}: span {span/1e6:.1f} ms, GPU busy (union) {union/1e6:.1f} ms, idle {(span-union)/1e6:.1f} ms, {len(seg)} launches")
    for k, v in busy.most_common(12):
        print(f"   {k[:60]:60s} {cnt[k]:5d} x  {v/1e6:8.2f} ms")
end_all = max(r["e"] for r in rows[last_q2_end:])
window(rows[last_q2_end]["e"], end_all, "after the last Q2 launch (Q1 back-transformation)")
window(rows[first_q2]["s"], rows[last_q2_end]["e"], "Q2")
# the stage before Q2: back to the last sb2st kernel
bc = [i for i, r in enumerate(rows[:first_q2]) if "sb2st" in r["k"]]
if bc:
    window(rows[bc[-1]]["e"], rows[first_q2]["s"], "between bulge chasing and Q2 (divide and conquer)")
