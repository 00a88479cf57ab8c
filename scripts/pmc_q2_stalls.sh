#!/bin/bash
# stall attribution of the Q2 kernel: SQ counters of sbback_apply_bal_kernel over one n = 20000 decomposition (own passes, no trace)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06/q2pmc
rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  JXGPU_OZ_PLANES=5 timeout 600 rocprofv3 --pmc $set --kernel-include-regex "sbback_apply" --output-format csv -d $O/p$i -- python3 scripts/time_eigh.py 20000 > $O/p$i.log 2>&1
  python3 scripts/pmc_summarize.py $O/p$i $O/p$i.json p$i > /dev/null 2>&1; tail -2 $O/p$i.log | cut -c1-200; rm -rf $O/p$i
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/prof_r06/q2pmc/p*.json')):
    d=json.load(open(f))
    for k,v in d.get('kernels',{}).items():
        if 'sbback_apply' in k:
            print(f, k[:40], {c: round(x['mean'],1) for c,x in v.items()})
P
