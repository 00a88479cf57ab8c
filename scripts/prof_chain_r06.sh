#!/bin/bash
# kernel-trace stats of the chain-scan leg (bench.py --leg c3_chain) -> gpurun_out/prof_r06/chain
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06
mkdir -p $O; rm -rf $O/chain
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chain -- python3 bench.py --leg c3_chain --steps 2 --warmup 1 > $O/chain.log 2>&1
rm -f $O/chain/*/*kernel_trace.csv
ls -la $O/chain/* | head
