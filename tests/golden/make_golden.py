"""Generates tests/golden/*.npz from the reference's own sample data and the REAL vendored nanoflann
(oracle/_ref/libref_nanoflann.so, built by oracle/Makefile from /root/reference/include/nanoflann.hpp).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The outputs are data (inputs + expected outputs); no reference source text is stored.

bunny_nn.npz
  src, src_nor  : samples/Bunny_RealData/cloudXYZ_2.xyz, every 4th row      (queries' cloud)
  dst, dst_nor  : samples/Bunny_RealData/cloudXYZ_0.xyz, every 4th row      (target cloud)
  pose_src/dst  : poses_2.txt / poses_0.txt (ground truth, 4x4 row-major)
  pose_src_noisy: pose_src perturbed (fixed numbers below) — an "early ICP round" query set
  q_gt, q_noisy : the src points expressed in the dst frame (oracle query transform), the exact query vectors
                  handed to nanoflann
  idx_*, d2_*   : nanoflann 1-NN index and squared distance per query (frame.cpp:195-205 semantics)
  ties_*        : number of queries whose best distance is attained by >1 target (must be 0: tie rules coincide)
bunny_knn.npz
  knn_idx, knn_d2: nanoflann knnSearch(k=10) around each dst point (Frame::getNeighbours, frame.cpp:208-231)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "mv-lm-icp_amd"))
import orclib  # noqa: E402
from mvicp import io as mio  # noqa: E402

REF = "/root/reference/samples/Bunny_RealData"


def main():
    orc = orclib.load()
    ref = orclib.load_ref()
    assert ref is not None, "oracle/_ref not built (needs /root/reference)"
    sp, sn = mio.load_xyz(os.path.join(REF, "cloudXYZ_2.xyz"))
    dp, dn = mio.load_xyz(os.path.join(REF, "cloudXYZ_0.xyz"))
    sp, sn, dp, dn = sp[::4].copy(), sn[::4].copy(), dp[::4].copy(), dn[::4].copy()
    Ps = mio.load_matrix4(os.path.join(REF, "poses_2.txt"))
    Pd = mio.load_matrix4(os.path.join(REF, "poses_0.txt"))
    # deterministic perturbation ~ addNoise(sigma=0.02, sigmat=0.01)
    w = np.array([0.013, -0.021, 0.008])
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    Rn = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * (W @ W)
    Pn = Ps.copy()
    Pn[:3, :3] = Ps[:3, :3] @ Rn
    Pn[:3, 3] += np.array([0.006, -0.004, 0.009])
    out = {"src": sp, "src_nor": sn, "dst": dp, "dst_nor": dn, "pose_src": Ps, "pose_dst": Pd, "pose_src_noisy": Pn}
    for tag, P in (("gt", Ps), ("noisy", Pn)):
        q = orc.query_transform(P, Pd, sp)
        idx, d2 = ref.query(dp, q)
        # tie census by exhaustive scan of exact equality
        bi, bd = orc.nn_brute(dp, q)
        ties = 0
        for k in range(0, len(q), 1):
            d = ((q[k] - dp) ** 2)
            dd = (d[:, 0] + d[:, 1]) + d[:, 2]
            ties += int((dd == dd.min()).sum() > 1)
        assert np.array_equal(bi, idx) and np.array_equal(bd, d2), "oracle brute force disagrees with nanoflann"
        out["q_" + tag] = q; out["idx_" + tag] = idx; out["d2_" + tag] = d2; out["ties_" + tag] = np.int64(ties)
        print(tag, "queries", len(q), "ties", ties, "max d", np.sqrt(d2.max()))
    np.savez_compressed(os.path.join(HERE, "bunny_nn.npz"), **out)
    ki, kd = ref.knn_self(dp, 10)
    np.savez_compressed(os.path.join(HERE, "bunny_knn.npz"), pts=dp, knn_idx=ki, knn_d2=kd)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
