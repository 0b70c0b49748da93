"""-m gpu: the headless C++ drivers (mv-lm-icp_amd/bin/multiview, pairwise — host mirror of the reference's
main_multiview.cpp / main_pairwise.cpp over the C ABI) on data written in the reference's on-disk formats."""
import os
import re
import subprocess

import numpy as np
import pytest

import mvicp
from mvicp import lib as L
from mvicp import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mv-lm-icp_amd", "bin")


def write_dataset(d, pb):
    for i, (p, n) in enumerate(zip(pb["pts"], pb["nor"])):
        np.savetxt(os.path.join(d, f"cloud_{i}.xyz"), np.hstack([p, n]), fmt="%.17g")
        np.savetxt(os.path.join(d, f"pose_{i}.txt"), pb["init"][i], fmt="%.17g")
        np.savetxt(os.path.join(d, f"groundtruth_{i}.txt"), pb["gt"][i], fmt="%.17g")


@pytest.mark.parametrize("flags,param,plane", [([], L.PARAM_SOPHUS_SE3, 1), (["--nosophusSE3", "--angleAxis"], L.PARAM_ANGLE_AXIS, 1),
                                               (["--nosophusSE3", "--nopointToPlane"], L.PARAM_EIGEN_QUATERNION, 0)])
def test_multiview_driver_matches_engine_loop(tmp_path, flags, param, plane):
    pb = synth.make_problem(5, 3000)
    d = tmp_path / "data"; o = tmp_path / "out"
    d.mkdir(); o.mkdir()
    write_dataset(str(d), pb)
    cmd = [os.path.join(BIN, "multiview"), "--dir", str(d), "--out", str(o), "--step", "1", "--limit", "40", "--rounds", "4", "--quiet", "--norecomputeNormals"] + flags
    subprocess.check_call(cmd)
    got = np.array([np.loadtxt(os.path.join(str(o), f"pose_{i}.txt")) for i in range(5)])
    # same loop through the Python binding; the driver's graph includes the fixed frame's own (inactive) edges
    src, dst = synth.pose_graph_knn(pb["init"], 2, skip_fixed0=False)
    eng = mvicp.Engine(0)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(src, dst)
    poses = pb["init"].copy()
    for _ in range(4):
        eng.correspond(poses, pb["fixed"], 0.05)
        poses, sm = eng.optimize(poses, pb["fixed"], param, plane, True, 50)
    eng.close()
    assert np.allclose(got, poses, rtol=0, atol=1e-14), np.abs(got - poses).max()


def test_correspondence_copy_back_through_frame_api(tmp_path):
    """Frame::neighbours[].correspondances filled by the driver-side adaptor equal the engine's lists (ascending src index)."""
    pb = synth.make_problem(3, 2000)
    d = tmp_path / "data"; o = tmp_path / "out"
    d.mkdir(); o.mkdir()
    write_dataset(str(d), pb)
    subprocess.check_call([os.path.join(BIN, "multiview"), "--dir", str(d), "--out", str(o), "--step", "1", "--rounds", "1", "--quiet", "--copyback"])
    assert os.path.exists(os.path.join(str(o), "pose_2.txt"))


def test_multiview_driver_default_flags_recompute_normals(tmp_path):
    """Reference defaults (recomputeNormals on): the driver's PCA normals + loop equal the engine's."""
    pb = synth.make_problem(4, 3000)
    d = tmp_path / "data"; o = tmp_path / "out"
    d.mkdir(); o.mkdir()
    write_dataset(str(d), pb)
    subprocess.check_call([os.path.join(BIN, "multiview"), "--dir", str(d), "--out", str(o), "--step", "1", "--rounds", "3", "--quiet"])
    got = np.array([np.loadtxt(os.path.join(str(o), f"pose_{i}.txt")) for i in range(4)])
    src, dst = synth.pose_graph_knn(pb["init"], 2, skip_fixed0=False)
    eng = mvicp.Engine(0)
    eng.set_frames(pb["pts"], None)
    nor = [eng.recompute_normals(i, 10) for i in range(4)]
    eng.set_frames(pb["pts"], nor); eng.set_graph(src, dst)
    poses = pb["init"].copy()
    for _ in range(3):
        eng.correspond(poses, pb["fixed"], 0.05)
        poses, sm = eng.optimize(poses, pb["fixed"], L.PARAM_SOPHUS_SE3, 1, True, 50)
    eng.close()
    assert np.allclose(got, poses, rtol=0, atol=1e-13), np.abs(got - poses).max()


def test_pairwise_driver_recovers_known_transform(tmp_path):
    G = np.load(os.path.join(ROOT, "tests", "golden", "bunny_nn.npz"))
    cloud = tmp_path / "cloud.xyz"
    np.savetxt(str(cloud), np.hstack([G["dst"], G["dst_nor"]]), fmt="%.17g")
    for extra in ([], ["--pointToPlane"]):
        out = subprocess.check_output([os.path.join(BIN, "pairwise"), "--cloud", str(cloud)] + extra).decode()
        vals = re.findall(r"diff_tra:([0-9.e+-]+)\s+diff_rot_degrees:([0-9.e+-]+)", out)
        assert len(vals) == 3, out
        for t, r in vals:
            assert float(t) < 1e-8 and float(r) < 1e-5, out  # README.md:141-146: ~1e-10 m, 1.7e-6 deg (acos floor)
