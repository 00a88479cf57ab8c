#!/bin/bash
# kernel trace of a few bench steps: what the GPU does in the 40 ms in front of every re-tiling kernel (repack_p32_kernel).
# Repeats until a process shows the slow prep (or 5 tries).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06/prepgap
for try in 1 2 3 4 5; do
  rm -rf $O; mkdir -p $O
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --no-extra --no-cpu-baseline --steps 4 --warmup 2 > $O.log 2>&1
  F=$(ls $O/*/*kernel_trace.csv | head -1)
  python3 - "$F" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
idx = [i for i, r in enumerate(rows) if "repack_p32" in r["Kernel_Name"]]
bad = False
for i in idx:
    r = rows[i]
    prev_end = max(x["e"] for x in rows[max(0, i - 50):i]) if i else r["s"]
    dur = (r["e"] - r["s"]) / 1e6
    gap = (r["s"] - prev_end) / 1e6
    print(f"repack #{idx.index(i)}: duration {dur:.3f} ms, gap since the previous kernel's end {gap:.3f} ms")
    if dur > 5 or gap > 200: bad = bad or dur > 5
    if dur > 5:
        print("   kernels overlapping it:")
        for x in rows:
            if x is not r and x["s"] < r["e"] and x["e"] > r["s"]:
                print("     ", x["Kernel_Name"][:80], (x["e"] - x["s"]) / 1e6, "ms", "queue", x.get("Queue_Id"), "stream", x.get("Stream_Id"))
sys.exit(0 if bad else 1)
P
  rc=$?
  rm -f $O/*/*kernel_trace.csv
  if [ $rc -eq 0 ]; then echo "slow process caught in try $try"; break; fi
  echo "try $try: no slow re-tile"
done
