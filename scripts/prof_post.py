"""From a rocprofv3 kernel trace of scripts/time_sytrd.py: time line of everything after the last sytrd_update launch of the
last repetition (divide-and-conquer + back-transformation)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last_upd = max(i for i, r in enumerate(rows) if 'sytrd_update_kernel' in r['Kernel_Name'])
post = rows[last_upd + 1:]
t0 = int(rows[last_upd]['End_Timestamp'])
agg = {}
for r in post:
    nm = r['Kernel_Name'].split('(')[0][-44:]
    agg.setdefault(nm, [0, 0.0])
    agg[nm][0] += 1
    agg[nm][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
span = (int(post[-1]['End_Timestamp']) - t0) / 1e6
busy = sum(v[1] for v in agg.values())
print(f"post-sytrd span {span:.2f} ms, kernel busy {busy:.2f} ms, launches {len(post)}")
for nm, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
    print(f"  {nm:46s} x{c:5d} {t:8.3f} ms")
# locate dormtr start: first 'larf'/'trmm' style kernel after the merge GEMMs is hard to know; print the time of the big gaps
prev = t0
gaps = []
for r in post:
    s = int(r['Start_Timestamp'])
    if s - prev > 30000:
        gaps.append(((s - t0) / 1e6, (s - prev) / 1e3, r['Kernel_Name'].split('(')[0][-40:]))
    prev = max(prev, int(r['End_Timestamp']))
print("idle gaps > 30 us (at ms, gap us, next kernel):")
for g in gaps[:25]:
    print("   %.2f  %.0f  %s" % g)
