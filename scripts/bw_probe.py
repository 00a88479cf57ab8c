import torch, time
dev=torch.device('cuda:0')
for n in (5000, 10000, 20000):
    a=torch.randn((n,n),dtype=torch.float64,device=dev)
    for name,fn in (('sum',lambda: a.sum()),('clone',lambda: a.clone()),('tril_sum',lambda: torch.tril(a).sum())):
        fn(); torch.cuda.synchronize()
        t0=time.perf_counter()
        for _ in range(10): r=fn()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
        print(n,name,'%.1f us'%(dt*1e6),'read GB/s %.0f'%(n*n*8/dt/1e9))
    del a
