"""False-positive rates of the reference filter at configs[4] (k = 21, one chunk of 1878 x 150 bp reads), simulated in numpy:
per-plane hit rates of random windows (plane D: 0.68, not 0.11), four-lane hits, and how many chunks per read pass the
candidate rules tried for search_wide_kernel (slice_search.hpp).  python tools/fp_sim.py"""
import numpy as np
rng=np.random.default_rng(1)
k=21;L=150
def keys(reads):
    # reads: (n,L) codes 0..3 ; A=0,C=1,G=2,T=3 ; hi = code>>1 , lo = code&1
    n=reads.shape[0]
    hi=(reads>>1).astype(np.uint64); lo=(reads&1).astype(np.uint64)
    ka=np.zeros((n,L-k+1),dtype=np.uint64); kb=np.zeros_like(ka)
    for j in range(k):
        ka=(ka<<np.uint64(1))|hi[:,j:L-k+1+j]
        kb=(kb<<np.uint64(1))|lo[:,j:L-k+1+j]
    return ka,kb
idx=rng.integers(0,4,size=(1878,L),dtype=np.uint8)
ka,kb=keys(idx)
A=np.zeros(1<<k,bool);B=A.copy();C=A.copy();D=A.copy()
A[ka.ravel()]=1;B[kb.ravel()]=1;C[(ka^kb).ravel()]=1;D[(ka|kb).ravel()]=1
print("density",A.mean(),B.mean(),C.mean(),D.mean())
q=rng.integers(0,4,size=(200000,L),dtype=np.uint8)
qa,qb=keys(q)
ha=A[qa];hb=B[qb];hc=C[qa^qb];hd=D[qa|qb]
print("hit rates",ha.mean(),hb.mean(),hc.mean(),hd.mean())
full=ha&hb&hc&hd
print("full",full.mean(), "a&b",(ha&hb).mean(),"abc",(ha&hb&hc).mean())
# candidate: >=2 full hits in windows [0..66] of a read (forward only here), first in [0..45]
f=full[:,:67]
once=f[:,:46]
# cand if exists i<j, i<46, f[i]&f[j]
cnt=f.sum(1)
first=np.where(once.any(1), once.argmax(1), 999)
cand=np.array([ (f[r,first[r]+1:].any() if first[r]<999 else False) for r in range(len(f))])
print("cand per read-strand-chunk",cand.mean(), " x 10421 x 2 =",cand.mean()*10421*2)
# alternatives
W=full.shape[1]  # 130 windows, index w = q-(k-1)
def blocks_with_hit(f, nwin):
    nb=(nwin+k-1)//k
    out=np.zeros((f.shape[0],nb),bool)
    for b in range(nb):
        out[:,b]=f[:,b*k:min((b+1)*k,nwin)].any(1)
    return out
b2=blocks_with_hit(full,67).sum(1)>=2
print("(b) >=2 blocks by lim2:",b2.mean(), b2.mean()*20842)
b3=blocks_with_hit(full,88).sum(1)>=3
print("(d) >=3 blocks by lim3:",b3.mean(), b3.mean()*20842)
def greedy_seen(f,nwin):
    seen=np.zeros(f.shape[0],int); nxt=np.zeros(f.shape[0],int)
    for w in range(nwin):
        take=f[:,w]&(w>=nxt)
        seen+=take; nxt=np.where(take,w+k,nxt)
    return seen
g2=greedy_seen(full,67)>=2
print("(c) greedy>=2 by lim2:",g2.mean(), g2.mean()*20842)
g3=greedy_seen(full,88)>=3
print("greedy>=3 by lim3:",g3.mean(), g3.mean()*20842)
# first-hit restricted: first hit must be within first 46 windows
h1=full[:,:46].any(1)
print("once (first-hit range):",h1.mean(), h1.mean()*20842)
def blocks_with_hit_off(f, nwin, off):
    # block id = (w+off)//k
    ids=(np.arange(nwin)+off)//k
    nb=ids.max()+1
    out=np.zeros((f.shape[0],nb),bool)
    for b in range(nb):
        out[:,b]=f[:,:nwin][:,ids==b].any(1)
    return out
for nwin,J in ((67,2),(88,3)):
    a=blocks_with_hit_off(full,nwin,0).sum(1)>=J
    b=blocks_with_hit_off(full,nwin,k//2).sum(1)>=J
    c=blocks_with_hit_off(full,nwin,k//3).sum(1)>=J
    d=blocks_with_hit_off(full,nwin,2*k//3).sum(1)>=J
    print(nwin,J,"two alignments:",(a&b).mean()*20842," three:",(a&c&d).mean()*20842)
abc=ha&hb&hc
for nwin,J in ((67,2),(88,3)):
    a=blocks_with_hit_off(abc,nwin,0).sum(1)>=J
    b=blocks_with_hit_off(abc,nwin,k//2).sum(1)>=J
    print("ABC only",nwin,J,"one alignment:",a.mean()*20842,"two:",(a&b).mean()*20842)
