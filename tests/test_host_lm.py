"""Host logic of the product (mv-lm-icp_amd/host/lm.cpp + se3.h) on CPU: the LM solve over per-edge canonical
blocks, with the oracle standing in for the GPU evaluator (tests only), must reproduce the oracle's own
Jet-based Ceres restatement for every parameterization — this checks the canonical->local chain rule (M maps),
the (+) operators, the pose<->parameter conversions and the trust-region loop."""
import numpy as np
import pytest

import orclib
from mvicp import lib as L
from mvicp import synth

PARAMS = [orclib.PARAM_QUAT, orclib.PARAM_ANGLEAXIS, orclib.PARAM_SOPHUS]


def build(orc, plane, robust, K=4, N=500, seed=5):
    rng = np.random.default_rng(seed)
    pb = synth.make_problem(K, N)
    src, dst = pb["src"], pb["dst"]
    corr, w = [], []
    for s, d in zip(src, dst):
        # correspondences from the oracle's own search so the problem is a real ICP round
        f, sec, dist, wt, _, _ = orc.correspond_edge(pb["pts"][s], pb["init"][s], pb["pts"][d], pb["init"][d], 0.05)
        corr.append((f, sec)); w.append(float(wt))
    return pb, corr, w


@pytest.mark.parametrize("param", PARAMS)
@pytest.mark.parametrize("plane,robust", [(1, 1), (0, 1), (1, 0), (0, 0)])
def test_lm_matches_oracle(orc, param, plane, robust):
    pb, corr, w = build(orc, plane, robust)
    K = len(pb["pts"])
    prob = orc.make_problem(pb["pts"], pb["nor"], pb["fixed"], pb["src"], pb["dst"], corr, w, param, plane, robust)
    P_ref, sm_ref = orc.optimize(prob, pb["init"], 50)

    def evaluator(poses):
        return orc.edge_blocks(pb["pts"], pb["nor"], pb["src"], pb["dst"], corr, w, poses, plane, robust)

    P, sm = L.lm_solve_host(K, pb["src"], pb["dst"], pb["init"], pb["fixed"], param, evaluator, 50)
    assert sm["termination"] == sm_ref["termination"], (sm, sm_ref)
    assert sm["iterations"] == sm_ref["iterations"], (sm, sm_ref)
    assert abs(sm["final_cost"] - sm_ref["final_cost"]) <= 1e-9 * sm_ref["final_cost"]
    for k in range(K):
        dt, dr = synth.pose_diff(P[k], P_ref[k])
        assert dt < 1e-9 and dr < 1e-9, (k, dt, dr)
    assert sm["final_cost"] < sm["initial_cost"] and sm["successful_steps"] >= 1


def test_fixed_everything_is_a_noop(orc):
    pb, corr, w = build(orc, 1, 1, K=2, N=200)
    ev = lambda poses: orc.edge_blocks(pb["pts"], pb["nor"], pb["src"], pb["dst"], corr, w, poses, 1, 1)
    P, sm = L.lm_solve_host(2, pb["src"], pb["dst"], pb["init"], [1, 1], orclib.PARAM_SOPHUS, ev, 50)
    assert sm["iterations"] == 0 and np.allclose(P, pb["init"], atol=1e-15)


def test_evaluator_error_propagates():
    def bad(poses):
        raise ValueError("boom")
    with pytest.raises(ValueError):
        L.lm_solve_host(2, [1], [0], np.array([np.eye(4)] * 2), [1, 0], 2, bad, 5)


@pytest.mark.parametrize("param", PARAMS)
def test_edges_from_a_fixed_source_are_excluded(orc, param):
    """icp-ceres.cpp:255,351,426 `if(srcCloud.fixed) continue;`: a graph that lists edges out of frame 0 (and out of a second
    fixed frame) must optimise exactly the objective without them — whatever their blocks contain."""
    rng = np.random.default_rng(8)
    pb = synth.make_problem(4, 400)
    src, dst = synth.pose_graph_knn(pb["init"], 2, skip_fixed0=False)   # frame 0 is a source here
    fixed = np.array([0, 0, 1, 0], dtype=np.uint8)                      # fixed[0] is forced by the solver; frame 2 is fixed by the caller
    corr, w = [], []
    for s, d in zip(src, dst):
        f, sec, dist, wt, _, _ = orc.correspond_edge(pb["pts"][s], pb["init"][s], pb["pts"][d], pb["init"][d], 0.05)
        corr.append((f, sec)); w.append(float(wt))
    keep = [e for e, s in enumerate(src) if s not in (0, 2)]
    prob_all = orc.make_problem(pb["pts"], pb["nor"], [1, 0, 1, 0], src, dst, corr, w, param, 1, 1)
    prob_kept = orc.make_problem(pb["pts"], pb["nor"], [1, 0, 1, 0], src[keep], dst[keep], [corr[e] for e in keep], [w[e] for e in keep], param, 1, 1)
    P_all, sm_all = orc.optimize(prob_all, pb["init"], 50)
    P_kept, sm_kept = orc.optimize(prob_kept, pb["init"], 50)
    assert np.array_equal(P_all, P_kept) and sm_all == sm_kept   # the oracle itself follows the reference rule

    def evaluator(poses):
        blk = orc.edge_blocks(pb["pts"], pb["nor"], src, dst, corr, w, poses, 1, 1)
        for e, s in enumerate(src):
            if s in (0, 2):
                blk[e] = rng.normal(0, 1e3, 91)   # must never be read
        return blk

    P, sm = L.lm_solve_host(4, src, dst, pb["init"], fixed, param, evaluator, 50)
    assert sm["iterations"] == sm_kept["iterations"] and sm["termination"] == sm_kept["termination"]
    for k in range(4):
        dt, dr = synth.pose_diff(P[k], P_kept[k])
        assert dt < 1e-9 and dr < 1e-9, (k, dt, dr)


@pytest.mark.parametrize("seed", range(12))
def test_random_pose_graphs(orc, seed):
    """Arbitrary directed pose graphs (not the ring the drivers build: several edges into one view, edges out of fixed views, views nobody
    constrains), random fixed masks with view 0 fixed as in the reference, every parameterization / cost / loss: the host solve (general
    sparsity path of the factorisation) against the oracle's."""
    rng = np.random.default_rng(300 + seed)
    K = int(rng.integers(2, 7))
    pb = synth.make_problem(K, int(rng.integers(150, 500)), pose_seed=int(4000 + seed))
    pairs = [(s, d) for s in range(K) for d in range(K) if s != d]
    pick = sorted(rng.choice(len(pairs), size=int(rng.integers(1, min(len(pairs), 8) + 1)), replace=False))
    src = np.array([pairs[i][0] for i in pick], dtype=np.int32); dst = np.array([pairs[i][1] for i in pick], dtype=np.int32)
    fixed = (rng.uniform(size=K) < 0.3).astype(np.int32); fixed[0] = 1
    param = PARAMS[seed % 3]
    plane, robust = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    corr, w = [], []
    for s, d in zip(src, dst):
        if fixed[s]:
            corr.append((np.zeros(0, np.int32), np.zeros(0, np.int32))); w.append(0.0)
            continue
        f, sec, dist, wt, _, _ = orc.correspond_edge(pb["pts"][s], pb["init"][s], pb["pts"][d], pb["init"][d], 0.05)
        corr.append((f, sec)); w.append(float(wt))
    prob = orc.make_problem(pb["pts"], pb["nor"], fixed, src, dst, corr, w, param, plane, robust)
    P_ref, sm_ref = orc.optimize(prob, pb["init"], 50)

    def evaluator(poses):
        return orc.edge_blocks(pb["pts"], pb["nor"], src, dst, corr, w, poses, plane, robust)

    P, sm = L.lm_solve_host(K, src, dst, pb["init"], fixed, param, evaluator, 50)
    assert sm["termination"] == sm_ref["termination"] and sm["iterations"] == sm_ref["iterations"], (seed, sm, sm_ref)
    for k in range(K):
        dt, dr = synth.pose_diff(P[k], P_ref[k])
        assert dt < 1e-8 and dr < 1e-8, (seed, k, dt, dr)


def test_product_lm_reproduces_the_readme_vector(orc):
    """The PRODUCT's host solve (host/lm.cpp: analytic chain rule over canonical blocks, its own factorisation) on the reference's published
    vector: README.md:141-146, real Ceres, angle-axis 7.76957e-11 / quaternion 6.31278e-11 (see test_oracle_lm.py::
    test_readme_known_answer_reproduced for the inputs).  The evaluator is the oracle's (no GPU here); the GPU evaluator goes through the
    same solve in tests/test_gpu_parity.py::test_pairwise_readme_vector_on_gpu."""
    import os
    K = np.load(os.path.join(os.path.dirname(__file__), "golden", "pairwise_kat.npz"))
    pts = np.vstack([K["pts"], K["pts"][-1:]])
    P = K["P_libcxx"]
    dstp = pts @ P[:3, :3].T + P[:3, 3]
    ids = np.arange(len(pts), dtype=np.int32)
    for param in (orclib.PARAM_ANGLEAXIS, orclib.PARAM_QUAT):
        ev = lambda poses: orc.edge_blocks([dstp, pts], None, [1], [0], [(ids, ids)], [0.0], poses, 0, 0)
        Pout, sm = L.lm_solve_host(2, [1], [0], np.array([np.eye(4), np.eye(4)]), [1, 0], param, ev, 50)
        dt = orc.pose_diff(P, Pout[1])[0]
        assert sm["termination"] == 2 and sm["iterations"] == 6, sm
        assert abs(dt / K["readme_dt"][param] - 1) < 1e-4, (param, dt)   # (1e-10-sized quantity out of 1e-16-relative arithmetic: 4+ digits)
