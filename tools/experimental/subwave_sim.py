"""CPU simulation (numpy/scipy, no GPU): 32-point target tiles that a GROUP of 64 / 32 / 16 / 8 / 1 consecutive queries must open (box-to-query
distance <= that query's NN distance + 0.1 mm), both clouds in the k-d order of nn_grid.hip.  The tile kernel works on groups of 64 (a wave);
the smaller groups bound what sub-wave traversal could save.  Output: profiles/r02_subwave_sim.txt."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'mv-lm-icp_amd'))
from mvicp import synth
from scipy.spatial import cKDTree
N=200000; K=32
pd_,_=synth.make_view(3,K,N); ps_,_=synth.make_view(4,K,N)
gt=[synth.gt_pose(k,K) for k in range(K)]
rng=np.random.default_rng(0)
def order_kd(p,leaf=32):
    # mirror of kd_order (nn_grid.hip): whole tiles left (largest power of two below the tile count), widest axis
    idx=np.arange(len(p)); res=[]
    sys.setrecursionlimit(100000)
    def rec(ix):
        m=len(ix)
        if m<=leaf: res.append(ix); return
        ext=p[ix].max(0)-p[ix].min(0); a=int(np.argmax(ext))
        tiles=(m+leaf-1)//leaf; left=1
        while left*2<tiles: left*=2
        mid=left*leaf
        part=np.argpartition(p[ix,a],mid)
        rec(ix[part[:mid]]); rec(ix[part[mid:]])
    rec(idx); return np.concatenate(res)
def rel(Ps,Pd):
    R=Pd[:3,:3].T@Ps[:3,:3]; t=Pd[:3,:3].T@(Ps[:3,3]-Pd[:3,3]); return R,t
R,t=rel(gt[4],gt[3])
od=order_kd(pd_); osrc=order_kd(ps_)
P=pd_[od]
for pert in (0.0003,0.001):
    dR=synth.so3_exp(rng.normal(0,pert/0.1,3)); dt=rng.normal(0,pert,3)
    q_all=(ps_@R.T+t)@dR.T+dt
    tree=cKDTree(pd_); dnn,_=tree.query(q_all)
    Q=q_all[osrc]; rad=dnn[osrc]+1e-4
    print('pert',pert,'median NN dist mm',np.median(dnn)*1e3)
    leaf=32
    nt=len(P)//leaf
    T=P[:nt*leaf].reshape(nt,leaf,3); tlo=T.min(1); thi=T.max(1)
    tl=cKDTree((tlo+thi)/2); maxhalf=np.linalg.norm((thi-tlo)/2,axis=1).max()
    for G in (64,32,16,8,1):
        ng=len(Q)//G; opened=[]
        for w in range(0,ng,max(1,ng//400)):
            q=Q[w*G:(w+1)*G]; r=rad[w*G:(w+1)*G]
            c=q.mean(0)
            cand=np.array(tl.query_ball_point(c, np.linalg.norm(q-c,axis=1).max()+r.max()+maxhalf))
            d=np.maximum(np.maximum(tlo[cand][None]-q[:,None], q[:,None]-thi[cand][None]),0)
            need=(np.linalg.norm(d,axis=2)<=r[:,None]).any(0)
            opened.append(need.sum())
        opened=np.array(opened)
        print(f'  group of {G:2d} queries: tiles opened per group {opened.mean():.2f} -> candidates/query {opened.mean()*leaf:.0f}')
