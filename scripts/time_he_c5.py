import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
from janusx_amd import janusx as jxrs
dev=torch.device("cuda",0)
n,m,ntr=200000,1000000,160000
pk,dos=bench.family_panel_gpu(n,m,4,11,dev)
y=bench.make_phenotype(dos,n,7,dev)
cnt=jxrs.bed_row_counts(pk,n).astype(np.int64)
nm=n-cnt[:,0]; alt=cnt[:,1]+2*cnt[:,2]
p=(alt/(2.0*np.maximum(nm,1))).astype(np.float32); maf=np.minimum(p,1-p).astype(np.float32); flip=p>0.5
tr=np.sort(np.random.default_rng(5).permutation(n)[:ntr]).astype(np.int64)
torch.cuda.synchronize(); t0=time.perf_counter()
c2=jxrs.bed_row_counts(pk,n,tr)
torch.cuda.synchronize(); t1=time.perf_counter()
h=jxrs.he_pcg_bed("",tr,y[tr],packed=pk,packed_n_samples=n,maf=maf,row_flip=flip,trace_samples=32)
torch.cuda.synchronize(); t2=time.perf_counter()
print("counts(train)",t1-t0,"he total",t2-t1)
