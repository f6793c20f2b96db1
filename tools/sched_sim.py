#!/usr/bin/env python
"""Wave-level model of traversal scheduling: 64 lanes, rays as random node / leaf step sequences with the measured
means (11.4 nodes, 3.6 leaves per ray of the 1 M-triangle scene); lanes per step for a pool of R rays per wave (64 = one ray per lane)
and for two private rays per lane.  Costs in VALU instructions per step: node 300, leaf 180, event 250."""
import random
random.seed(2)
CN, CL, CEV = 300, 180, 250
def make_ray():
    nN = max(3, int(random.gauss(11.4, 4))); nL = max(1, int(random.gauss(3.6, 1.5)))
    seq = ['N']*min(4,nN); rest = ['N']*(nN-len(seq)) + ['L']*nL; random.shuffle(rest); return seq+rest
def sim_pool(R, nrays=64*300, refill_frac=0.2, ovh=1.15):
    rays=[make_ray() for _ in range(nrays)]; nxt=0
    pool=[None]*R; cost=0; work=0
    def refill():
        nonlocal nxt
        for i in range(R):
            if pool[i] is None and nxt<nrays: pool[i]=[rays[nxt],0]; nxt+=1
    refill()
    while True:
        act=[l for l in pool if l is not None]
        if not act and nxt>=nrays: break
        if (R-len(act) >= refill_frac*R and nxt<nrays) or not act:
            cost+=CEV; refill(); continue
        N=[l for l in act if l[0][l[1]]=='N']; L=[l for l in act if l[0][l[1]]=='L']
        if len(N)>=len(L): sel=N[:64]; c=CN
        else: sel=L[:64]; c=CL
        cost+=c*ovh; work+=len(sel)*c
        for l in sel: l[1]+=1
        for i in range(R):
            l=pool[i]
            if l is not None and l[1]>=len(l[0]): pool[i]=None
    return cost/(nrays/64), work/cost
def sim_2perlane(nrays=64*300, ovh=1.13, refill=12):
    rays=[make_ray() for _ in range(nrays)]; nxt=0
    lanes=[[None,None] for _ in range(64)]; cost=0; work=0
    def refill_():
        nonlocal nxt
        for ln in lanes:
            for k in (0,1):
                if ln[k] is None and nxt<nrays: ln[k]=[rays[nxt],0]; nxt+=1
    refill_()
    while True:
        nact=sum(1 for ln in lanes for r in ln if r is not None)
        if nact==0 and nxt>=nrays: break
        if (128-nact>=2*refill and nxt<nrays) or nact==0:
            cost+=CEV; refill_(); continue
        nN=sum(1 for ln in lanes if any(r is not None and r[0][r[1]]=='N' for r in ln))
        nL=sum(1 for ln in lanes if any(r is not None and r[0][r[1]]=='L' for r in ln))
        kind='N' if nN>=nL else 'L'; c=CN if kind=='N' else CL
        n=0
        for ln in lanes:
            for k in (0,1):
                r=ln[k]
                if r is not None and r[0][r[1]]==kind:
                    r[1]+=1; n+=1
                    if r[1]>=len(r[0]): ln[k]=None
                    break
        cost+=c*ovh; work+=n*c
    return cost/(nrays/64), work/cost
print('pool 64', sim_pool(64, ovh=1.0))
for R in (96,128,192,256): print('pool',R, sim_pool(R))
print('2 per lane', sim_2perlane())
