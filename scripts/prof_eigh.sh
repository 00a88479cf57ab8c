#!/bin/bash
# kernel-trace stats of one eigendecomposition (time_eigh.py N), top kernels printed.  usage: prof_eigh.sh N [tag]
set -u
N=${1:-20000}; TAG=${2:-eigh}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 scripts/time_eigh.py $N > $O/stats.log 2>&1
tail -2 $O/stats.log
rm -f $O/stats/*/*kernel_trace.csv
F=$(ls $O/stats/*/*kernel_stats.csv | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):6d} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
