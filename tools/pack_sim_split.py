import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from ava256_amd.scene import make_scene
from oracle.mvp_oracle import Oracle
from helpers import scene_rays
o = Oracle("f64")
cfg = sys.argv[1] if len(sys.argv)>1 else "C2"
N,H,W,K = {"C2":(1,512,512,4096),"C3":(1,512,512,16384),"C4":(1,1024,1024,8192)}[cfg]
s = make_scene(N,H,W,K,device="cpu",seed=1112)
rp, rd, tm = scene_rays(o, s)
rp=rp[0].reshape(-1,3); rd=rd[0].reshape(-1,3); tm=tm[0].reshape(-1,2)
pos=s["primpos"][0].numpy().astype(np.float64); rot=s["primrot"][0].numpy().astype(np.float64); sc=s["primscale"][0].numpy().astype(np.float64)
dt=float(s["stepsize"])
rng=np.random.default_rng(0)
STEP, CHUNK = 262.0, 150.0
res={}
lens=[]
def cost(items):
    it=np.sort(np.array(items))[::-1]
    ch=[it[i:i+64] for i in range(0,len(it),64)]
    return sum(c.max() for c in ch)*STEP + len(ch)*CHUNK, sum(c.max() for c in ch), len(ch)
for k in rng.choice(K, 150, replace=False):
    xmt = rp - pos[k]; r0 = (xmt @ rot[k]) * sc[k]; dd = (rd @ rot[k]) * sc[k]
    with np.errstate(all="ignore"):
        t0 = (-1-r0)/dd; t1=(1-r0)/dd
    tn = np.minimum(t0,t1).max(1); tf=np.maximum(t0,t1).min(1)
    hit = tn<=tf
    ta=np.maximum(tn,tm[:,0]); tb=np.minimum(tf,tm[:,1]+1e-5)
    lo=np.ceil((ta-tm[:,0])/dt-0.02); hi=np.floor((tb-tm[:,0])/dt+0.02)
    ln=np.where(hit & (lo<=hi), hi-lo+1, 0).astype(int)
    L=ln[ln>0]
    if len(L)==0: continue
    lens.append(L)
    res.setdefault("cur",[]).append(cost(L))
    res.setdefault("ideal",[]).append((L.sum()/64*STEP + np.ceil(len(L)/64)*CHUNK, L.sum()/64, np.ceil(len(L)/64)))
    for T in (3,4,5,6,8,10,12):
        items=[]
        for l in L:
            n=int(np.ceil(l/T)); q,r=divmod(l,n)
            items += [q+1]*r + [q]*(n-r)
        res.setdefault("even<=%d"%T,[]).append(cost(items))
    # adaptive: threshold = factor x median length of this primitive
    for f in (1.0,1.25,1.5,2.0):
        T=max(2,int(np.ceil(f*np.median(L))))
        items=[]
        for l in L:
            n=int(np.ceil(l/T)); q,r=divmod(l,n)
            items += [q+1]*r + [q]*(n-r)
        res.setdefault("med x%.2f"%f,[]).append(cost(items))
allL=np.concatenate(lens)
print(cfg, "rays/prim %.1f"%np.mean([len(l) for l in lens]), "len mean %.2f median %d p90 %d p99 %d max %d"%(allL.mean(), np.median(allL), np.percentile(allL,90), np.percentile(allL,99), allL.max()))
base=np.sum([c[0] for c in res["cur"]])
for k,v in res.items():
    print("%-10s VALU/prim %8.0f (%.3f)  wave-steps %.2f chunks %.2f"%(k, np.mean([c[0] for c in v]), np.sum([c[0] for c in v])/base, np.mean([c[1] for c in v]), np.mean([c[2] for c in v])))
# (round 6, second policy) split in TWO halves only the rays longer than T, at most `room` of them per primitive (the queue has
# kQueueCap - 64 * entries free slots): what a cheap in-kernel form could do
print("---- halves of rays longer than T (at most `room` splits per primitive)")
for T in (5, 6, 7, 8, 9):
    for room in (32, 64, 128, 10**6):
        tot = 0.0
        for L in lens:
            Ls = np.sort(L)[::-1]
            items = []
            nsplit = 0
            for l in Ls:
                if l > T and nsplit < room:
                    items += [int(np.ceil(l / 2)), int(l // 2)]
                    nsplit += 1
                else:
                    items.append(int(l))
            tot += cost(items)[0]
        print("T=%d room=%-7s VALU ratio %.3f" % (T, room if room < 10**6 else "inf", tot / base))
