"""Wherever the REAL Ceres Solver exists, pin the part of the LM that the reference's only published vector (README.md:141-146: pairwise,
point-to-point, no loss, a solve that never rejects a step) does not reach: a robust (SoftLOneLoss(edge.weight), icp-ceres.cpp:374,449)
point-to-plane MULTIVIEW solve — several free poses, the function-tolerance stop of the multiview rounds, the SE(3) local parameterization
(sophus_se3.h) and the plain angle-axis blocks — through tools/ceres_oracle.cpp (own code over ceres::AutoDiffCostFunction with the reference's
options, icp-ceres.cpp:66-89) against the product's host LM (mv-lm-icp_amd/host/lm.cpp over the oracle's per-edge blocks) at 1e-8.

Ceres / Eigen are absent from this image and from the GPU box (tools/ceres_probe.py): the test then SKIPS — DESIGN.md §7 lists this as the
remaining unpinned part.  No GPU needed."""
import os
import struct
import subprocess

import numpy as np
import pytest

import orclib
from mvicp import lib as L
from mvicp import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "_build", "ceres_oracle")


@pytest.fixture(scope="module")
def ceres_oracle():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_ceres_oracle.sh")], capture_output=True, text=True)
    if r.returncode == 3:
        pytest.skip("Ceres Solver / Eigen3 not installed on this machine: " + r.stdout.strip())
    assert r.returncode == 0 and os.path.exists(TOOL), r.stdout + r.stderr
    return TOOL


def write_problem(path, pb, corr, weights, param, plane, robust):
    K, E = len(pb["pts"]), len(pb["src"])
    with open(path, "wb") as f:
        f.write(b"MVCERES1")
        f.write(struct.pack("<5i", K, E, param, plane, int(robust)))
        for k in range(K):
            f.write(struct.pack("<2i", len(pb["pts"][k]), int(pb["fixed"][k])))
            f.write(np.ascontiguousarray(pb["pts"][k], dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(pb["nor"][k], dtype="<f8").tobytes())
            f.write(np.ascontiguousarray(pb["init"][k].T, dtype="<f8").tobytes())          # column-major 4x4
        for e in range(E):
            fi, se = corr[e]
            f.write(struct.pack("<3i", int(pb["src"][e]), int(pb["dst"][e]), len(fi)))
            f.write(struct.pack("<d", float(np.float32(weights[e]))))                       # OutgoingEdge::weight is a float (frame.h:26)
            f.write(np.ascontiguousarray(fi, dtype="<i4").tobytes()); f.write(np.ascontiguousarray(se, dtype="<i4").tobytes())


def test_tool_source_is_present_and_cites_the_reference():
    """(runs everywhere) the optional checker is committed, own code, and names the reference lines it restates."""
    src = open(os.path.join(ROOT, "tools", "ceres_oracle.cpp")).read()
    for cite in ("icp-ceres.cpp:66-89", "icp-ceres.cpp:325-395", "sophus_se3.h", "SoftLOneLoss", "AutoDiffLocalParameterization", "SPARSE_NORMAL_CHOLESKY"):
        assert cite in src, cite
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "build_ceres_oracle.sh")], capture_output=True, text=True)
    assert r.returncode in (0, 3), r.stdout + r.stderr          # builds, or says cleanly that Ceres is not here — never a broken build


@pytest.mark.parametrize("param,oparam", [(2, orclib.PARAM_SOPHUS), (1, orclib.PARAM_ANGLEAXIS)])
def test_robust_point_to_plane_multiview_solve_against_real_ceres(ceres_oracle, orc, tmp_path, param, oparam):
    pb = synth.make_problem(8, 4000)                                  # cfg3's shape (8 views, ring graph, E = 14), reduced to 4 000 points per view
    corr, w = [], []
    for s, d in zip(pb["src"], pb["dst"]):
        f, sec, dist, wt, _, _ = orc.correspond_edge(pb["pts"][s], pb["init"][s], pb["pts"][d], pb["init"][d], 0.05)
        corr.append((f, sec)); w.append(float(wt))
    prob = tmp_path / "problem.bin"; out = tmp_path / "poses.bin"
    write_problem(str(prob), pb, corr, w, param, 1, True)
    subprocess.check_call([ceres_oracle, str(prob), str(out)])
    raw = open(str(out), "rb").read()
    K = len(pb["pts"])
    P_ceres = np.frombuffer(raw[:K * 128], dtype="<f8").reshape(K, 4, 4).transpose(0, 2, 1)
    iters, term = struct.unpack("<2i", raw[K * 128:K * 128 + 8])

    def evaluator(poses):
        return orc.edge_blocks(pb["pts"], pb["nor"], pb["src"], pb["dst"], corr, w, poses, 1, 1)

    P, sm = L.lm_solve_host(K, pb["src"], pb["dst"], pb["init"], pb["fixed"], oparam, evaluator, 50)
    assert sm["iterations"] == iters, (sm, iters, term)
    for k in range(K):
        dt, dr = synth.pose_diff(P[k], P_ceres[k])
        assert dt < 1e-8 and dr < 1e-8, (k, dt, dr)
