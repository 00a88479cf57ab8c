#!/bin/bash
# LDS counters of the Q2 kernel with parts switched off (JXGPU_QB_SKIP masks: results WRONG, timing / counters only)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06/q2lds
rm -rf $O; mkdir -p $O
for skip in 0; do
  JXGPU_QB_SKIP=$skip JXGPU_OZ_PLANES=5 timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-include-regex "sbback_apply" --output-format csv -d $O/s$skip -- python3 scripts/time_eigh.py 20000 > $O/s$skip.log 2>&1
  python3 scripts/pmc_summarize.py $O/s$skip $O/s$skip.json s$skip > /dev/null 2>&1; rm -rf $O/s$skip
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/prof_r06/q2lds/s*.json')):
    d=json.load(open(f))
    for k,v in d.get('kernels',{}).items():
        if 'sbback_apply' in k:
            print(f.split('/')[-1], k[:40], {c: '%.3e' % x['mean'] for c,x in v.items()})
P
