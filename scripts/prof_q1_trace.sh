#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06/q1trace
rm -rf $O; mkdir -p $O
JXGPU_OZ_PLANES=5 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 scripts/time_eigh.py 20000 > $O.log 2>&1
tail -1 $O.log
F=$(ls $O/*/*kernel_trace.csv | head -1)
python3 scripts/trace_q1_stage.py "$F"
rm -f $O/*/*kernel_trace.csv
