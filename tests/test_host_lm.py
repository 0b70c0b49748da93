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
