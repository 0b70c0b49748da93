"""CPU simulation (numpy/scipy, no GPU): how many 32-/16-point tiles must a wave of 64 consecutive queries open (box-to-lane distance
<= that lane's NN distance + 0.1 mm) when both clouds are ordered by the Hilbert curve of their hash cells vs by a balanced k-d split?
cfg4-like view pair at two misalignments.  Output: profiles/r02_kd_order_sim.txt.  This is what motivated grid_curve = 2."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'mv-lm-icp_amd'))
from mvicp import synth
from scipy.spatial import cKDTree

def spread(v):
    v = v.astype(np.uint64) & np.uint64(0x1fffff)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v
def hilbert3(x,y,z,bits):
    X=[x.astype(np.uint32).copy(),y.astype(np.uint32).copy(),z.astype(np.uint32).copy()]
    M=np.uint32(1<<(bits-1))
    Q=int(M)
    while Q>1:
        P=np.uint32(Q-1)
        for i in range(3):
            m=(X[i]&np.uint32(Q))!=0
            X[0]=np.where(m, X[0]^P, X[0])
            t=(X[0]^X[i])&P
            X[0]=np.where(m, X[0], X[0]^t)
            X[i]=np.where(m, X[i], X[i]^t)
        Q>>=1
    for i in range(1,3): X[i]^=X[i-1]
    t=np.zeros_like(X[0])
    Q=int(M)
    while Q>1:
        t=np.where((X[2]&np.uint32(Q))!=0, t^np.uint32(Q-1), t)
        Q>>=1
    for i in range(3): X[i]^=t
    return spread(X[2])|(spread(X[1])<<np.uint64(1))|(spread(X[0])<<np.uint64(2))
N=200000; K=32
pd_,_=synth.make_view(3,K,N); ps_,_=synth.make_view(4,K,N)
gt=[synth.gt_pose(k,K) for k in range(K)]
rng=np.random.default_rng(0)
def order_hilbert(p,h=0.0016):
    lo=p.min(0)-0.01*h; c=np.floor((p-lo)/h).astype(np.int64)
    bits=int(np.ceil(np.log2(c.max()+1)))
    key=hilbert3(c[:,0],c[:,1],c[:,2],bits)
    return np.lexsort((np.arange(len(p)),key))
def order_kd(p,leaf=32):
    idx=np.arange(len(p)); out=[]
    stack=[idx]
    res=[]
    def rec(ix):
        if len(ix)<=leaf: res.append(ix); return
        ext=p[ix].max(0)-p[ix].min(0); a=int(np.argmax(ext))
        m=len(ix)//2
        # split at a multiple of leaf so that leaves are full
        m=(m//leaf)*leaf if (m//leaf)*leaf>0 else m
        part=np.argpartition(p[ix,a],m)
        rec(ix[part[:m]]); rec(ix[part[m:]])
    sys.setrecursionlimit(10000)
    rec(idx)
    return np.concatenate(res)
# query transform: src local -> dst local via GT poses plus a small perturbation (round-3-like: ~0.4 mm median NN distance)
def rel(Ps,Pd): 
    R=Pd[:3,:3].T@Ps[:3,:3]; t=Pd[:3,:3].T@(Ps[:3,3]-Pd[:3,3]); return R,t
R,t=rel(gt[4],gt[3])
for pert in (0.0003,0.001):
    dR=synth.so3_exp(rng.normal(0,pert/0.1,3)); dt=rng.normal(0,pert,3)
    q_all=(ps_@R.T+t)@dR.T+dt
    tree=cKDTree(pd_); dnn,_=tree.query(q_all)
    print('pert',pert,'median NN dist mm',np.median(dnn)*1e3)
    for name,fo in (('hilbert',order_hilbert),('kd',order_kd)):
        od=fo(pd_); osrc=fo(ps_)
        P=pd_[od]; Q=q_all[osrc]; rad=dnn[osrc]*1.0+1e-4   # per-lane radius = NN distance + 0.1 mm
        for leaf in (32,16):
            nt=len(P)//leaf
            T=P[:nt*leaf].reshape(nt,leaf,3); tlo=T.min(1); thi=T.max(1)
            tl=cKDTree((tlo+thi)/2)
            half=(thi-tlo)/2; maxhalf=np.linalg.norm(half,axis=1).max()
            nw=len(Q)//64
            opened=[]; 
            for w in range(0,nw,37):
                q=Q[w*64:(w+1)*64]; r=rad[w*64:(w+1)*64]
                cand=tl.query_ball_point(q.mean(0), np.linalg.norm(q-q.mean(0),axis=1).max()+r.max()+maxhalf)
                cand=np.array(cand)
                # exact box-lane test
                d=np.maximum(np.maximum(tlo[cand][None,:,:]-q[:,None,:], q[:,None,:]-thi[cand][None,:,:]),0)
                dist=np.linalg.norm(d,axis=2)  # (64, ncand)
                need=(dist<=r[:,None]).any(0)
                opened.append(need.sum())
            opened=np.array(opened)
            print(f'  order {name:8s} leaf {leaf}: tiles opened per wave mean {opened.mean():.1f} median {np.median(opened):.0f}  -> candidates/query {opened.mean()*leaf:.0f}; tile box diag mean {np.linalg.norm(thi-tlo,axis=1).mean()*1e3:.2f} mm')
