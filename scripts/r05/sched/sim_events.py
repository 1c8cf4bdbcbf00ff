import numpy as np, heapq, sys
rng=np.random.default_rng(1)
it8=np.load('/tmp/it_full.npy').astype(int)
TS=np.load('/tmp/traj_S.npy'); TP=np.load('/tmp/traj_P.npy'); TD=np.load('/tmp/traj_D.npy')
row=-np.ones(len(it8),int); row[TS]=np.arange(len(TS))
LR=np.log10(np.maximum(np.maximum(TP,TD),1e-12))   # log residual per iteration (index k-1)
REP=8
perm=np.concatenate([rng.permutation(len(it8)) for _ in range(REP)])
L=it8[perm]; ROW=row[perm]
N=len(L); S=2048
T_IT=2.6e-3; PARK=0.047; LTOL=-6.0
def predict(idx,k,q):
    r=ROW[idx]
    now=LR[r,k-1]; prev=LR[r,k-q-1] if k-q-1>=0 else now+1.0
    rate=(prev-now)/q
    if rate<=1e-5: return 2000.0
    return (now-LTOL)/rate
def simulate(q=64,LONG=300,SHORT=48,rr=True,mode='pred'):
    free=[(0.0,i) for i in range(S)]; heapq.heapify(free)
    nxt=0; done=np.zeros(N,int)
    avail=[]; pend=[]; seq=0; t_end=0.0; parks=0
    # each server holds at most one instance in progress: (idx) stored in cur
    cur=[-1]*S
    while free:
        t,sv=heapq.heappop(free)
        while pend and pend[0][0]<=t:
            ta,sq,ix=heapq.heappop(pend); heapq.heappush(avail,(sq,ix))
        idx=cur[sv]; cost=0.0
        if idx<0:
            if nxt<N: idx=nxt; nxt+=1
            elif avail: sq,idx=heapq.heappop(avail); cost=PARK/2
            elif pend: heapq.heappush(free,(pend[0][0],sv)); continue
            else: continue
        # run to next event or completion
        k0=done[idx]; k1=min(L[idx],(k0//q+1)*q)
        tf=t+cost+(k1-k0)*T_IT
        done[idx]=k1
        if k1>=L[idx]: cur[sv]=-1
        else:
            if mode=='perfect': pr=L[idx]-k1
            else: pr=predict(idx,k1,q)
            fresh_left=nxt<N
            waiting=fresh_left or len(avail)>0 or len(pend)>0
            if pr>=LONG or pr<=SHORT or not waiting: cur[sv]=idx   # continue
            elif fresh_left or rr:
                tf+=PARK/2; parks+=1; seq+=1; heapq.heappush(pend,(tf,seq,idx)); cur[sv]=-1
            else: cur[sv]=idx
        t_end=max(t_end,tf)
        heapq.heappush(free,(tf,sv))
    return round(t_end,3),parks
print("ideal", L.sum()*T_IT/S)
for q in (32,48,64,96,128):
  for LONG in (200,300,400):
    for SHORT in (32,64):
        print("q",q,"LONG",LONG,"SHORT",SHORT,"pred",simulate(q,LONG,SHORT),"perfect",simulate(q,LONG,SHORT,mode='perfect'))
