import csv, statistics, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ks={}
for r in rows:
    nm=r['Kernel_Name'].split('(')[0][-40:]
    ks.setdefault(nm,[]).append((int(r['Start_Timestamp']),int(r['End_Timestamp'])))
sy=ks.get('jx::sytrd_symv_kernel',[]); up=ks.get('jx::sytrd_update_kernel',[])
n=len(sy)//2; sy=sy[n:]; up=up[n:]
for a in (0,1000,2500,4000,4900):
    b=a+50
    d=[e-s for s,e in sy[a:b]]; du=[e-s for s,e in up[a:b]]
    per=[(sy[k+1][0]-sy[k][0]) for k in range(a,b)]
    print(a,'symv us %.1f'%(statistics.mean(d)/1e3),'upd us %.1f'%(statistics.mean(du)/1e3),'period us %.1f'%(statistics.mean(per)/1e3))
print('sytrd span ms %.1f'%((up[-1][1]-sy[0][0])/1e6))
# everything after the last update kernel of the second run
t_end=up[-1][1]
rest={}
for nm,v in ks.items():
    tot=sum(e-s for s,e in v if s>=t_end)
    if tot>0: rest[nm]=tot
for nm,t in sorted(rest.items(), key=lambda x:-x[1])[:8]:
    print('after sytrd: %-42s %.2f ms'%(nm,t/1e6))
last=max(e for v in ks.values() for s,e in v)
print('after-sytrd span ms %.1f'%((last-t_end)/1e6))
