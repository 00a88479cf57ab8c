#!/bin/bash
# LDS counters of every kernel of one C3 step (own pass): which kernels lose LDS cycles to bank conflicts
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r06/ldsall
rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $O/p.log 2>&1
python3 scripts/pmc_summarize.py $O/p $O/p.json lds > /dev/null 2>&1; rm -rf $O/p
python3 - <<'P'
import json
d=json.load(open('gpurun_out/prof_r06/ldsall/p.json'))['kernels']
rows=[]
for k,v in d.items():
    try:
        g=v['GRBM_GUI_ACTIVE']['sum']; idx=v['SQ_LDS_IDX_ACTIVE']['sum']; bc=v['SQ_LDS_BANK_CONFLICT']['sum']; n=v['SQ_INSTS_LDS']['sum']
        rows.append((g,k,idx,bc,n,v['GRBM_GUI_ACTIVE']['calls'], v['SQ_WAIT_INST_ANY']['sum']/max(v['SQ_WAVE_CYCLES']['sum'],1)))
    except Exception: pass
rows.sort(reverse=True)
for g,k,idx,bc,n,c,w in rows[:22]:
    print(f"{k[:56]:56s} calls {c:5d} gui {g:.2e} lds_active/cu_cycles {idx/(g/8*256+1):.2f} conflict/active {bc/max(idx,1):.2f} cyc/inst {idx/max(n,1):.1f} wait_inst {w:.2f}")
P
