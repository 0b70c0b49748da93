"""CPU simulation (numpy/scipy, no GPU): level-0 children (32-point target tiles) that survive the tile kernel's wave-level cull
(tile box vs the AABB of the wave's 64 queries, radius = the largest per-lane search radius) compared with the tiles some lane really needs
(per-lane box test), and what 2 / 4 / 8 sub-boxes of the query patch (consecutive 32 / 16 / 8 queries, each with its own radius) would let
through.  Both clouds in the k-d order of nn_grid.hip; per-lane radius = NN distance + 0.1 mm (seeded, nearly converged round).
Output: profiles/r02_cull_sim.txt."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'mv-lm-icp_amd'))
from mvicp import synth
from scipy.spatial import cKDTree
N = 200000; K = 32
pd_, _ = synth.make_view(3, K, N); ps_, _ = synth.make_view(4, K, N)
gt = [synth.gt_pose(k, K) for k in range(K)]
rng = np.random.default_rng(0)

def order_kd(p, leaf=32):
    idx = np.arange(len(p)); res = []
    sys.setrecursionlimit(100000)
    def rec(ix):
        m = len(ix)
        if m <= leaf: res.append(ix); return
        ext = p[ix].max(0) - p[ix].min(0); a = int(np.argmax(ext))
        tiles = (m + leaf - 1) // leaf; left = 1
        while left * 2 < tiles: left *= 2
        mid = left * leaf
        part = np.argpartition(p[ix, a], mid)
        rec(ix[part[:mid]]); rec(ix[part[mid:]])
    rec(idx); return np.concatenate(res)

def box_dist(lo, hi, blo, bhi):   # distance between AABB [lo,hi] and boxes [blo,bhi] (n,3)
    d = np.maximum(np.maximum(blo - hi, lo - bhi), 0)
    return np.linalg.norm(d, axis=1)

R = gt[3][:3, :3].T @ gt[4][:3, :3]; t = gt[3][:3, :3].T @ (gt[4][:3, 3] - gt[3][:3, 3])
od = order_kd(pd_); osrc = order_kd(ps_)
P = pd_[od]
leaf = 32
nt = len(P) // leaf
T = P[:nt * leaf].reshape(nt, leaf, 3); tlo = T.min(1); thi = T.max(1)
tl = cKDTree((tlo + thi) / 2); maxhalf = np.linalg.norm((thi - tlo) / 2, axis=1).max()
for pert in (0.0003, 0.001):
    dR = synth.so3_exp(rng.normal(0, pert / 0.1, 3)); dt = rng.normal(0, pert, 3)
    q_all = (ps_ @ R.T + t) @ dR.T + dt
    dnn, _ = cKDTree(pd_).query(q_all)
    Q = q_all[osrc]; rad = dnn[osrc] + 1e-4
    print('pert', pert, 'median NN dist mm', np.median(dnn) * 1e3)
    nw = len(Q) // 64
    acc = {1: [], 2: [], 4: [], 8: [], 'need': []}
    for w in range(0, nw, max(1, nw // 400)):
        q = Q[w * 64:(w + 1) * 64]; r = rad[w * 64:(w + 1) * 64]
        c = q.mean(0)
        cand = np.array(tl.query_ball_point(c, np.linalg.norm(q - c, axis=1).max() + r.max() + maxhalf))
        blo, bhi = tlo[cand], thi[cand]
        d = np.maximum(np.maximum(blo[None] - q[:, None], q[:, None] - bhi[None]), 0)
        need = (np.linalg.norm(d, axis=2) <= r[:, None]).any(0)
        acc['need'].append(need.sum())
        for parts in (1, 2, 4, 8):
            g = 64 // parts; ok = np.zeros(len(cand), bool)
            for s in range(parts):
                qq = q[s * g:(s + 1) * g]; rr = r[s * g:(s + 1) * g].max()
                ok |= box_dist(qq.min(0), qq.max(0), blo, bhi) <= rr
            acc[parts].append(ok.sum())
    need = np.mean(acc['need'])
    print(f'  tiles some lane needs (opened): {need:.2f} per wave')
    for parts in (1, 2, 4, 8):
        print(f'  wave-level cull with {parts} sub-box(es): {np.mean(acc[parts]):.2f} tiles reach the per-lane test ({np.mean(acc[parts]) - need:.2f} of them for nothing)')
