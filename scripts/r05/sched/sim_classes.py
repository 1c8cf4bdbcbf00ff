import numpy as np, heapq, sys
exec(open('/tmp/sim3.py').read().split("if __name__")[0])
def simulateK(q1=128,qr=64,edges=(300,),slack=1.5):
    edges=np.array(edges)
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
            pr=predict2(idx,k,qr)
            others = nxt<N or len(avail)>0 or len(pend)>0
            if not others:
                n=min(L[idx]-k,qr); tf+=n*T_IT; done[idx]+=n; continue
            budget[idx]=int(slack*pr)+32
            tf+=PARK/2; parks+=1; seq+=1
            heapq.heappush(pend,(tf,-int(np.searchsorted(edges,pr)),seq,idx)); break
        t_end=max(t_end,tf)
        heapq.heappush(free,(tf,sv))
    return round(t_end,3),parks
for q1 in (128,160):
    print(q1,"K=1 (fifo, run to budget)",simulateK(q1,64,edges=()))
    print(q1,"K=2 @300",simulateK(q1,64,edges=(300,)))
    print(q1,"K=2 @500",simulateK(q1,64,edges=(500,)))
    print(q1,"K=3 @150,450",simulateK(q1,64,edges=(150,450)))
    print(q1,"K=4 @100,250,500",simulateK(q1,64,edges=(100,250,500)))
    print(q1,"K=8",simulateK(q1,64,edges=(50,100,150,200,300,450,650)))
    print(q1,"K=8 slack 1.2",simulateK(q1,64,edges=(50,100,150,200,300,450,650),slack=1.2))
    print(q1,"K=8 slack 2.5",simulateK(q1,64,edges=(50,100,150,200,300,450,650),slack=2.5))
print("----")
for q1 in (128,160):
    for K in (16,32):
        e=np.exp(np.linspace(np.log(24),np.log(2000),K-1))
        print(q1,"K",K,"log-spaced",simulateK(q1,64,edges=e,slack=1.5), "slack 2.5", simulateK(q1,64,edges=e,slack=2.5), "qr=48", simulateK(q1,48,edges=e,slack=1.5),"qr=96", simulateK(q1,96,edges=e,slack=1.5))
