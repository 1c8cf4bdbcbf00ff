import numpy as np, heapq, sys
exec(open('/tmp/sim2.py').read().split("def simulate")[0])
def predict2(idx,k,q):
    r=ROW[idx]
    now=LR[r,k-1]; prev=LR[r,max(k-q-1,0)]
    rate=(prev-now)/q
    if rate<=2e-4: return 2000.0   # stalled: will not converge within max_iter at this rate
    return min(2000.0,(now-LTOL)/rate)
def simulate(q1=128,qr=64,mode='pred',slack=1.5,fresh_long_continue=False,LONG=400):
    free=[(0.0,i) for i in range(S)]; heapq.heapify(free)
    nxt=0; done=np.zeros(N,int)
    avail=[]; pend=[]; seq=0; t_end=0.0; parks=0
    budget=np.zeros(N,int)
    while free:
        t,sv=heapq.heappop(free)
        while pend and pend[0][0]<=t:
            ta,pk,sq,ix=heapq.heappop(pend); heapq.heappush(avail,(pk,sq,ix))
        cost=0.0
        if nxt<N: idx=nxt; nxt+=1; n=min(L[idx],q1)
        elif avail: pk,sq,idx=heapq.heappop(avail); cost=PARK/2; n=min(L[idx]-done[idx],budget[idx])
        elif pend: heapq.heappush(free,(pend[0][0],sv)); continue
        else: continue
        tf=t+cost+n*T_IT
        done[idx]+=n
        while done[idx]<L[idx]:
            k=done[idx]
            pr=(L[idx]-k) if mode=='perfect' else predict2(idx,k,qr)
            others = nxt<N or len(avail)>0 or len(pend)>0
            if not others or (fresh_long_continue and nxt<N and pr>=LONG):
                # carry on without parking for another chunk
                n=min(L[idx]-k,qr); tf+=n*T_IT; done[idx]+=n; continue
            budget[idx]=int(slack*pr)+32
            tf+=PARK/2; parks+=1; seq+=1
            heapq.heappush(pend,(tf,-pr,seq,idx)); break
        t_end=max(t_end,tf)
        heapq.heappush(free,(tf,sv))
    return round(t_end,3),parks
if __name__=='__main__':
    print("ideal", L.sum()*T_IT/S)
    for q1 in (96,128,160,192):
        for qr in (32,64):
            for flc in (False,True):
                print("q1",q1,"qr",qr,"long continue",flc,"pred",simulate(q1,qr,fresh_long_continue=flc),"perfect",simulate(q1,qr,mode='perfect',fresh_long_continue=flc))
