# CPU simulation (numpy; no GPU): which fraction of (random plane, cell) pairs survives the box test of level 1, by the
# ORDER behind the 512-record cells -- Morton runs, Morton runs re-partitioned as k-d trees (what cells.h: k_refine_runs
# builds for runs of 8192 records) for several run lengths, the exact k-d partition, and a k-d top from SAMPLED medians
# (16-way per stage) with the refinement behind it.  DESIGN 3.1a quotes the numbers.
#   python3 tools/kd_order_sim.py [points]
import sys, time, numpy as np
sys.path.insert(0,'.')
from lsqrrecipes_amd import synth
def sampled_kd(pts, stages=3, sample=8192, levels=4, stop=8192):
    order = np.arange(len(pts))
    segs = [(0, len(pts))]
    for _ in range(stages):
        nxt = []
        for a, b in segs:
            n = b - a
            if n <= stop:
                nxt.append((a, b))
                continue
            seg = order[a:b]
            step = max(1, n // sample)
            samp = seg[::step][:sample]
            ids_s = np.zeros(len(samp), np.int64)
            ids = np.zeros(n, np.int64)
            P, S = pts[seg], pts[samp]
            for lv in range(levels):
                new_s, new = ids_s * 2, ids * 2
                for node in range(1 << lv):
                    ms = ids_s == node
                    if not ms.any():
                        continue
                    c = S[ms]
                    ax = int(np.argmax(c.max(0) - c.min(0)))
                    thr = np.median(c[:, ax])
                    new_s[ms] += c[:, ax] > thr
                    m = ids == node
                    new[m] += P[m, ax] > thr
                ids_s, ids = new_s, new
            o = np.argsort(ids, kind='stable')
            order[a:b] = seg[o]
            cnt = np.bincount(ids, minlength=1 << levels)
            off = a + np.concatenate([[0], np.cumsum(cnt)])
            nxt += [(int(off[i]), int(off[i + 1])) for i in range(1 << levels) if cnt[i]]
        segs = nxt
    return order
N=int(sys.argv[1]) if len(sys.argv)>1 else 2_000_000; CP=512
data=synth.plane(N,0.5)[0]; pts=data[:,:3].copy()
def kd_refine_chunks(order, chunk=8192, cp=CP):
    out=order.copy()
    for a in range(0,len(order),chunk):
        seg=out[a:a+chunk]
        stack=[(0,len(seg))]
        while stack:
            lo,hi=stack.pop()
            if hi-lo<=cp: continue
            c=pts[seg[lo:hi]]; ax=int(np.argmax(c.max(0)-c.min(0)))
            cells=-(-(hi-lo)//cp); m=(cells//2)*cp
            o=np.argpartition(c[:,ax],m) if m<hi-lo else np.arange(hi-lo)
            seg[lo:hi]=seg[lo:hi][o]
            stack.append((lo,lo+m)); stack.append((lo+m,hi))
        out[a:a+chunk]=seg
    return out
def kd_exact(order):
    return kd_refine_chunks(order, chunk=len(order))
def morton(pts,bits=10):
    lo=pts.min(0); sc=((1<<bits)-1)/(pts.max(0)-lo)
    q=((pts-lo)*sc).astype(np.uint64)
    def spread(v):
        r=np.zeros_like(v)
        for b in range(bits): r|=((v>>np.uint64(b))&np.uint64(1))<<np.uint64(3*b)
        return r
    key=spread(q[:,0])|(spread(q[:,1])<<np.uint64(1))|(spread(q[:,2])<<np.uint64(2))
    return np.argsort(key,kind='stable')
def survive(order, nplanes=400, delta=0.5, seed=1):
    g=np.random.default_rng(seed)
    nc=len(order)//CP; P=pts[order[:nc*CP]].reshape(nc,CP,3); lo=P.min(1); hi=P.max(1); ctr=(lo+hi)/2; half=(hi-lo)/2
    tot=0
    for _ in range(nplanes):
        i=g.choice(N,3,replace=False); a,b,c=pts[i]; n=np.cross(b-a,c-a); n/=np.linalg.norm(n); d=n@a
        dist=np.abs(ctr@n-d); r=half@np.abs(n)
        tot+=np.count_nonzero(dist<r+delta)
    return tot/(nplanes*nc)
np_=200 if N>4_000_000 else 400
t=time.time(); om=morton(pts); print("morton runs                          %.4f"%survive(om,np_),flush=True)
for ch in (8192,32768,65536,262144,1048576):
    if ch<N: print("morton + k-d inside runs of %-8d %.4f"%(ch,survive(kd_refine_chunks(om,ch),np_)),flush=True)
print("exact k-d partition                  %.4f"%survive(kd_exact(np.arange(N)),np_),flush=True)
o4=sampled_kd(pts,stages=3)
print("sampled k-d top (3 x 16-way)         %.4f"%survive(o4,np_),flush=True)
print("  ... + k-d inside runs of 8192      %.4f"%survive(kd_refine_chunks(o4),np_),flush=True)
