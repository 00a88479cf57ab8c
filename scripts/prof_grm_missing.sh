#!/bin/bash
# kernel-trace stats of two bench steps at 1 % missing calls (GRM kernels of the dense two-Gram form).  GPU box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_grm_miss
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --missing 0.01 > $O/stats.log 2>&1
tail -c 600 $O/stats.log
rm -f $O/stats/*/*kernel_trace.csv
F=$(ls $O/stats/*/*kernel_stats.csv | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    nm = r['Name']
    if any(k in nm for k in ("grm", "gm_", "packed_dot", "partition", "classify", "gather", "lut")):
        print(f"{nm[:110]:110s} calls {int(r['Calls']):5d} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
