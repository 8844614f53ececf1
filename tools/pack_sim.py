import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from ava256_amd.scene import make_scene
from oracle.mvp_oracle import Oracle
from helpers import scene_rays
o = Oracle("f64")
N,H,W,K = 1,512,512,4096
s = make_scene(N,H,W,K,device="cpu",seed=1112)
rp, rd, tm = scene_rays(o, s)
rp=rp[0].reshape(-1,3); rd=rd[0].reshape(-1,3); tm=tm[0].reshape(-1,2)
pos=s["primpos"][0].numpy().astype(np.float64); rot=s["primrot"][0].numpy().astype(np.float64); sc=s["primscale"][0].numpy().astype(np.float64)
dt=float(s["stepsize"])
rng=np.random.default_rng(0)
res={"cur":[], "ideal":[], "split3":[], "split4":[], "split5":[], "split6":[], "nq":[], "samples":[], "cur2w":[]}
px=np.arange(W)[None,:].repeat(H,0).reshape(-1); py=np.arange(H)[:,None].repeat(W,1).reshape(-1)
pkt=(py//8)*(W//8)+(px//8)
for k in rng.choice(K, 200, replace=False):
    xmt = rp - pos[k]
    r0 = (xmt @ rot[k]) * sc[k]      # y_j = sum_i R[i][j]*xmt_i * s_j
    dd = (rd @ rot[k]) * sc[k]
    with np.errstate(all="ignore"):
        t0 = (-1-r0)/dd; t1=(1-r0)/dd
    tn = np.minimum(t0,t1).max(1); tf=np.maximum(t0,t1).min(1)
    hit = tn<=tf
    ta=np.maximum(tn,tm[:,0]); tb=np.minimum(tf,tm[:,1]+1e-5)
    lo=np.ceil((ta-tm[:,0])/dt-0.02); hi=np.floor((tb-tm[:,0])/dt+0.02)
    ln=np.where(hit & (lo<=hi), hi-lo+1, 0).astype(int)
    # packet-level: all rays in packets with any hit are examined; queued = ln>0
    L=np.sort(ln[ln>0])[::-1]
    if len(L)==0: continue
    chunks=[L[i:i+64] for i in range(0,len(L),64)]
    cur=sum(c.max() for c in chunks)
    w0=sum(c.max() for c in chunks[0::2]); w1=sum(c.max() for c in chunks[1::2])
    res["cur"].append(cur); res["cur2w"].append(max(w0,w1)); res["ideal"].append(L.sum()/64); res["nq"].append(len(L)); res["samples"].append(L.sum())
    for S in (3,4,5,6):
        items=[]
        for l in L:
            while l> S: items.append(S); l-=S
            items.append(l)
        it=np.sort(np.array(items))[::-1]
        res["split%d"%S].append(sum(it[i:i+64].max() for i in range(0,len(it),64)) )
for k,v in res.items(): print(k, "mean %.2f"%np.mean(v))
print("utilisation cur %.3f" % (np.sum(res["ideal"])/np.sum(res["cur"])), {("split%d"%S): round(np.sum(res["ideal"])/np.sum(res["split%d"%S]),3) for S in (3,4,5,6)})
# band schedule
print("---- band schedule")
import collections
tot=collections.defaultdict(float); ch=collections.defaultdict(float)
rng=np.random.default_rng(0)
for k in rng.choice(K, 200, replace=False):
    xmt = rp - pos[k]; r0 = (xmt @ rot[k]) * sc[k]; dd = (rd @ rot[k]) * sc[k]
    with np.errstate(all="ignore"):
        t0 = (-1-r0)/dd; t1=(1-r0)/dd
    tn = np.minimum(t0,t1).max(1); tf=np.maximum(t0,t1).min(1)
    hit = tn<=tf
    ta=np.maximum(tn,tm[:,0]); tb=np.minimum(tf,tm[:,1]+1e-5)
    lo=np.ceil((ta-tm[:,0])/dt-0.02); hi=np.floor((tb-tm[:,0])/dt+0.02)
    ln=np.where(hit & (lo<=hi), hi-lo+1, 0).astype(int)
    L=np.sort(ln[ln>0])[::-1]
    if len(L)==0: continue
    for S in (3,4,5,6,8):
        b=0; c=0; nchunks=0
        while True:
            rem=L-b*S; m=(rem>0).sum()
            if m==0: break
            for i in range(0,m,64):
                c+=min(S, rem[i:i+64].max()); nchunks+=1
            b+=1
        tot[S]+=c; ch[S]+=nchunks
    tot["ideal"]+=L.sum()/64; ch["cur"]+=(len(L)+63)//64
print({k: round(v/200,2) for k,v in tot.items()}, "chunks", {k: round(v/200,2) for k,v in ch.items()})
