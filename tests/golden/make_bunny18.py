"""Generates tests/golden/bunny18.npz: the reference's DEFAULT multiview workload as data, plus the trajectory of the
reference-equivalent CPU path on it.  Run in the build container only (needs /root/reference):   python tests/golden/make_bunny18.py

Workload = main_multiview's defaults (src/main_multiview.cpp:30-51,63,74-86): dir samples/Bunny_RealData, limit 40 step 2 -> the 18 views
cloudXYZ_{0,2,...,34}.xyz (file order: common.h:150-156, by length then name = numeric), recomputeNormals on, frame 0 keeps its
ground-truth pose, the others addNoise(gt, 0.02, 0.01) from ONE default-seeded std::mt19937 (common.h:36-67) in frame order, knn 2
(frame.cpp:67-89), cutoff 0.05, point-to-plane, SophusSE3, robust, 20 rounds.  cfg1 (BASELINE.json configs[0]) = the same with --limit 2
--nopointToPlane: views 0 and 2, point-to-point; stored beside it.

What is stored (data only — inputs and expected outputs; no reference source text):
  xyz_e8, row_off : every row of the 18 clouds as int32 multiples of 1e-8 m (the files hold at most 8 decimals, and int / 1e8 is the
                    correctly rounded value of the decimal text: the same double the reference parses), concatenated; frame i = rows row_off[i] : row_off[i + 1]   (ragged sizes kept).  The phantom trailing
                    element of the reference's loadXYZ (common.h:233-238: an exact duplicate of the last row, pinned by
                    profiles/r05_lm_pin_sweep.txt) is NOT stored: the loaders append it (the default), and every recorded
                    output below is for the clouds WITH it — row_off[i + 1] - row_off[i] + 1 points per view
  gt              : poses_{0,2,...,34}.txt (18 x 4 x 4)
  init            : the noisy initial poses (default-seeded stream, compiled C++: oracle orc_add_noise)
  src, dst        : the pose graph INCLUDING frame 0's own (never searched) edges, as the driver builds it
  counts, weights : [20, E] per round: correspondences kept per edge and the float edge weight (1.5 x upper median, frame.cpp:166-176)
  poses           : [20, 18, 4, 4] poses after each round's solve
  lm_iters        : [20]
  nor_mean        : [18, 3] mean of the CPU-side PCA normals per view (real nanoflann 10-NN + numpy eigh; a coarse check only: the
                    normals themselves are recomputed by the path under test, as the reference does)
  cfg1_*          : the same fields for the 2-view point-to-point run
CPU path = tests/cpupath.py: REAL nanoflann (oracle/_ref) for every search, the oracle's Jet / LM restatement of Ceres for the solves
(pinned on the reference's published pairwise vector, README.md:141-146: tests/test_oracle_lm.py::test_readme_known_answer_reproduced)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "mv-lm-icp_amd"))
import cpupath  # noqa: E402
import orclib  # noqa: E402
from mvicp import io as mio  # noqa: E402
from mvicp import synth  # noqa: E402

REF = "/root/reference/samples/Bunny_RealData"


def pca_normals(pts, knn_idx):
    """common.h:331-346 pointSetPCA on Frame::getNeighbours(i, 10) (frame.cpp:208-231,244-255): eigenvector of the smallest eigenvalue of
    the (unnormalised) covariance of the 10 neighbours, flipped so that n_z <= 0."""
    nb = pts[knn_idx]                                  # n x 10 x 3
    c = nb - nb.mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", c, c)
    w, v = np.linalg.eigh(cov)
    nor = v[:, :, 0].copy()
    nor[nor[:, 2] > 0] *= -1.0
    return nor


def run(orc, pts, nor, gt, plane, rounds=20):
    K = len(pts)
    init = [gt[0].copy()]
    for i in range(1, K):
        init.append(orc.add_noise(gt[i], 0.02, 0.01, reset=(i == 1)))   # ONE generator, first use = frame 1 (main_multiview.cpp:83)
    init = np.array(init)
    src, dst = synth.pose_graph_knn(init, 2, skip_fixed0=False)
    fixed = np.zeros(K, dtype=np.uint8); fixed[0] = 1
    cp = cpupath.CpuPath(pts, nor, src, dst, fixed, 2, plane)
    P = init.copy()
    counts, weights, poses, iters = [], [], [], []
    for r in range(rounds):
        P, sm = cp.round(P)
        counts.append(cp.last["counts"].astype(np.int32)); weights.append(cp.last["weights"].astype(np.float32)); poses.append(P.copy()); iters.append(sm["iterations"])
        print("round", r + 1, "corr", int(cp.last["counts"].sum()), "lm", sm["iterations"], flush=True)
    cp.close()
    return {"init": init, "src": src, "dst": dst, "counts": np.array(counts), "weights": np.array(weights), "poses": np.array(poses), "lm_iters": np.array(iters, dtype=np.int32)}


def main():
    orc = orclib.load()
    ref = orclib.load_ref()
    assert ref is not None, "oracle/_ref not built (needs /root/reference)"
    ids = list(range(0, 36, 2))
    pts, gt, um = [], [], []
    for i in ids:
        p, _ = mio.load_xyz(os.path.join(REF, f"cloudXYZ_{i}.xyz"), phantom_row=False)
        u = np.rint(p * 1e8).astype(np.int32)
        assert np.array_equal(u / 1e8, p), "the cloud does not sit on the 1e-8 lattice"
        pts.append(np.vstack([p, p[-1:]])); um.append(u)   # the reference's cloud = the file's rows + the duplicated last row
        gt.append(mio.load_matrix4(os.path.join(REF, f"poses_{i}.txt")))
    nor = []
    for p in pts:
        ki, _ = ref.knn_self(p, 10)
        nor.append(pca_normals(p, ki))
    out = {"xyz_e8": np.concatenate(um), "row_off": np.cumsum([0] + [len(u) for u in um]).astype(np.int64), "gt": np.array(gt),
           "nor_mean": np.array([n.mean(axis=0) for n in nor])}
    print("cfg default: 18 views, rows", [len(p) for p in pts])
    out.update(run(orc, pts, nor, np.array(gt), 1))
    print("cfg1: views 0 / 2, point-to-point")
    r1 = run(orc, pts[:2], nor[:2], np.array(gt[:2]), 0)
    out.update({"cfg1_" + k: v for k, v in r1.items()})
    np.savez_compressed(os.path.join(HERE, "bunny18.npz"), **out)
    print("wrote bunny18.npz", os.path.getsize(os.path.join(HERE, "bunny18.npz")), "bytes")


if __name__ == "__main__":
    main()
