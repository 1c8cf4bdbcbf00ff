import numpy as np, heapq, sys
rng=np.random.default_rng(1)
it8=np.load('/tmp/it_full.npy').astype(int)
R={k:np.load('/tmp/res_%d.npy'%k) for k in (64,96,128,160,192,224,288)}
REP=8
perm=np.concatenate([rng.permutation(len(it8)) for _ in range(REP)])
L=it8[perm]
N=len(L); S=2048
T_IT=2.6e-3
PARK=0.047
def feat(q1):
    r=np.maximum(R[q1][0],R[q1][1])[perm]
    return np.log10(np.maximum(r,1e-12)/1e-6)
def simulate(policy,q1=288,q2=96,nb=8,score=None,edges=None):
    free=[(0.0,i) for i in range(S)]; heapq.heapify(free)
    nxt=0
    done=np.zeros(N,int)
    avail=[]   # heap of (prio_key, seq, idx) available now
    pend=[]    # heap of (time, prio_key, seq, idx)
    seq=0; t_end=0.0; parks=0
    while free:
        t,sv=heapq.heappop(free)
        while pend and pend[0][0]<=t:
            ta,pk,sq,ix=heapq.heappop(pend); heapq.heappush(avail,(pk,sq,ix))
        if nxt<N:
            idx=nxt; nxt+=1; n=min(L[idx],q1); cost=0.0
        elif avail:
            pk,sq,idx=heapq.heappop(avail); rem=L[idx]-done[idx]
            n=min(rem,q2) if policy=='rr' else rem; cost=PARK/2
        elif pend:
            heapq.heappush(free,(pend[0][0],sv)); continue
        else:
            continue
        tf=t+cost+n*T_IT
        done[idx]+=n
        if done[idx]<L[idx]:
            tf+=PARK/2; parks+=1; seq+=1
            if policy=='rr': pk=0
            else: pk=-int(np.searchsorted(edges,score[idx]))
            heapq.heappush(pend,(tf,pk,seq,idx))
        t_end=max(t_end,tf)
        heapq.heappush(free,(tf,sv))
    return round(t_end,3),parks
W=L.sum()
print("ideal packed ms", W*T_IT/S, "N",N)
print("rr 288/96", simulate('rr',288,96))
print("rr 288/288", simulate('rr',288,288))
print("no slices", simulate('rr',100000,96))
edges=np.array([50,100,150,200,300,450,650])
for q1 in (64,96,128,160,192,224,288):
    print(q1,"perfect buckets", simulate('prio',q1,score=(L-q1).astype(float),edges=edges), end='  ')
    f=feat(q1)
    surv=L>q1
    rem=(L-q1)[surv]; fs=f[surv]
    qs=np.quantile(fs,np.linspace(0,1,21))
    binmean=np.array([rem[(fs>=qs[i])&(fs<=qs[i+1])].mean() for i in range(20)])
    bi=np.clip(np.searchsorted(qs,f)-1,0,19); pred=binmean[bi]
    print("residual predictor", simulate('prio',q1,score=pred,edges=edges), "bin means", np.round(binmean[::4]))
print("---- priority by log primal residual at park time")
for q1 in (64,96,128,160,192,224,288):
    lp=np.log10(np.maximum(R[q1][0][perm],1e-12))
    e=np.arange(-5.0,-1.4,0.5)
    print(q1, "primal buckets", simulate('prio',q1,score=lp,edges=e), " max(p,d):", simulate('prio',q1,score=np.log10(np.maximum(np.maximum(R[q1][0],R[q1][1])[perm],1e-12)),edges=e))
