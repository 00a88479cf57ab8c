"""Association-stage time of the exact scan with covariates (block form, dim 7) at n = 20 000, m = 100 000: per-SNP scan and the
reference-default chain scan; run once as is and once with JXGPU_SCAN_INTERP=0 (direct evaluations)."""
import os,sys,time,math
sys.path.insert(0,".")
import numpy as np, torch, bench
from janusx_amd import pipeline as pl, stats as st
dev=torch.device("cuda",0)
n,m=20000,100000
packed,dos=bench.synth_panel_gpu(n,m,20260609,dev)
y=bench.make_phenotype(dos,n,20260609,dev)
k,_,panel=pl.build_grm(packed,n,1,0.02,0.05)
s,ut=pl.eigh_from_grm(k,1e-6,f32_consumer=True)
for ncov in (5,):
    x=np.concatenate([np.ones((n,1)),np.random.default_rng(3).normal(size=(n,ncov))],1)
    model=pl.SpectralModel(s,ut,x,y)
    counts=panel.counts(); keep,af,miss=st.gwas_scan_row_stats(counts,n,0.02,0.05,1.0); rows=np.nonzero(keep)[0]
    lut=st.scan_lut_from_counts(af[rows],np.zeros(len(rows),bool),counts[rows],n)
    lo,hi=model.null.bounds; init=min(max(math.log10(model.null.lbd),lo),hi)
    res={}
    for tag,co in (("nochain",None),("chain10000",st.warm_chain_offsets(st.warm_chain_blocks_bed(rows,m,10000),len(rows),1))):
        for rep in range(2):
            tm=pl.StageTimes(); out,ev=pl.scan_rows(panel,model,rows,lut,"lmm",init_log10_lbd=(init if co is not None else None),chain_off=co,times=tm,return_evals=True)
        res[tag]=out.cpu().numpy()
        print("cov",ncov,tag,"assoc ms %.1f"%(tm.t["scan"]*1e3),"evals/snp %.3f"%float(ev.float().mean()), flush=True)
