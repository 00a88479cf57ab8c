"""Idle gaps in a rocprofv3 kernel trace (last eigh of scripts/time_sytrd.py): where the GPU waits for the host."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last repetition: from the last sytrd_panel_start_kernel with the largest grid (first panel) onwards
starts = [i for i, r in enumerate(rows) if 'add_diag' in r['Kernel_Name']]
i0 = starts[-1]
seg = rows[i0:]
t0 = int(seg[0]['Start_Timestamp'])
end = t0
tot_gap = 0
gaps = []
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > end:
        g = s - end
        tot_gap += g
        if g > 40000:
            gaps.append(((end - t0) / 1e6, g / 1e3, r['Kernel_Name'].split('(')[0][-36:]))
    end = max(end, e)
print(f"span {(end - t0)/1e6:.2f} ms, total idle {tot_gap/1e6:.2f} ms in {len(seg)} launches")
small = tot_gap - sum(g[1] * 1e3 for g in gaps)
print(f"idle in gaps <= 40 us: {small/1e6:.2f} ms")
for g in gaps[:40]:
    print("  at %8.2f ms  gap %7.0f us  before %s" % g)
