#!/bin/bash
# matrix-pipe counters of the sliced int8 GEMM at the eigensolver's shapes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_oz
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 scripts/bench_ozgemm.py big > $O/mfma.log 2>&1
tail -3 $O/mfma.log
python3 scripts/pmc_summarize.py $O/mfma $O/mfma.json mfma | head -60
python3 - <<'PY'
import json
r = json.load(open("gpurun_out/pmc_oz/mfma.json"))
for k, v in r["kernels"].items():
    if "oz_" in k:
        print(k, {c: (round(x["mean"], 1), x["calls"]) for c, x in v.items()})
PY
