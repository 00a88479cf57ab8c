#!/bin/bash
# kernel-trace stats of one bench leg: prof_leg.sh <leg> -> gpurun_out/prof_r06/leg_<leg>
set -u
LEG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06/leg_$LEG
rm -rf $O; mkdir -p $O
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --leg $LEG > $O.log 2>&1
rm -f $O/*/*kernel_trace.csv
python3 - "$O" <<'P'
import csv,glob,sys
c=sorted(glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(c)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:25]: print(f'{r["Name"][:70]:70s} {int(r["Calls"]):7d} {float(r["TotalDurationNs"])/1e6:10.2f} ms  avg {float(r["AverageNs"])/1e3:10.1f} us')
P
tail -1 $O.log | cut -c1-1500
