import numpy as np
def run(lens, W=640, K=11, G=256, dbg=False):
    H=6*(K-1); TT=W-2*H; V1=W-H; B=len(lens)
    rows=sum(lens)
    if dbg:
        cA=((rows+G-1)//G+TT-1)//TT; cA = 0 if cA<4 else cA; LA=cA*TT
    else:
        cA=((rows+G-1)//G+H+V1-1)//V1; cA = 0 if cA<4 else cA; LA=cA*V1-H
    na=[(l//LA if cA else 0) for l in lens]
    nb=[(l-a*LA+TT-1)//TT for l,a in zip(lens,na)]
    pre=np.concatenate([[0],np.cumsum(nb)]); cpre=np.concatenate([[0],np.cumsum(na)])
    total=pre[B]; nA=cpre[B]
    assert nA<=G,(nA,G)
    cover=[np.zeros(l,int) for l in lens]
    ctr=[0]
    def claim():
        v=ctr[0]; ctr[0]+=1; return v
    def btile(j):
        b=0
        while pre[b+1]<=j: b+=1
        return b,(cpre[b+1]-cpre[b])*LA+(j-pre[b])*TT
    # all WGs "in parallel": order of claims arbitrary; emulate sequentially
    wgs=[]
    for w in range(G):
        if w<nA:
            b=0
            while cpre[b+1]<=w: b+=1
            i=w-cpre[b]; a_end=(i+1)*LA; a_left=cA-1
            if i==0 and not dbg: t0,olo,ohi=H,0,V1
            else: t0,olo,ohi=i*LA,H,W-H
            wgs.append(dict(b=b,t0=t0,olo=olo,ohi=ohi,in_a=True,a_left=a_left,a_end=a_end))
        else:
            j=claim()
            if j>=total: continue
            b,t0=btile(j)
            wgs.append(dict(b=b,t0=t0,olo=H,ohi=H+TT,in_a=False,a_left=0,a_end=0))
    ntiles=0
    active=wgs
    while active:
        nxt=[]
        for s in active:
            ntiles+=1
            base=s['t0']-H
            lo=max(base+s['olo'],0); hi=min(base+s['ohi'],lens[s['b']])
            if hi>lo: cover[s['b']][lo:hi]+=1
            chained=s['in_a'] and s['a_left']>0
            if chained:
                if dbg:
                    t0n=s['t0']+TT; olon=H; ohin=(H+s['a_end']-t0n) if s['a_end']-t0n<TT else H+TT
                else:
                    t0n=s['t0']+V1; olon=0; d=s['a_end']-(t0n-H); ohin=d if d<V1 else V1
                nxt.append(dict(b=s['b'],t0=t0n,olo=olon,ohi=ohin,in_a=True,a_left=s['a_left']-1,a_end=s['a_end']))
            else:
                j=claim()
                if j<total:
                    b,t0=btile(j)
                    nxt.append(dict(b=b,t0=t0,olo=H,ohi=H+TT,in_a=False,a_left=0,a_end=0))
        active=nxt
    bad=[(i,(c!=1).sum()) for i,c in enumerate(cover) if (c!=1).any()]
    print("dbg" if dbg else "carry","cA",cA,"LA",LA,"nA",nA,"total B",total,"tiles",ntiles,"bad",bad[:5])
run([8192]*70,dbg=True); run([8192]*70)
rng=np.random.default_rng(0); lens=np.clip(rng.normal(364,110,60),120,740).astype(int); lens[0]=740
run(list(lens*128),dbg=True); run(list(lens*128)); run(list(lens*128),K=7)
