"""-m gpu: the headless C++ drivers (mv-lm-icp_amd/bin/multiview, pairwise — host mirror of the reference's
main_multiview.cpp / main_pairwise.cpp over the C ABI) on data written in the reference's on-disk formats."""
import os
import re
import subprocess

import numpy as np
import pytest

import cpupath
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
    cmd = [os.path.join(BIN, "multiview"), "--dir", str(d), "--out", str(o), "--step", "1", "--limit", "40", "--rounds", "4", "--quiet", "--norecomputeNormals", "--drop_phantom_row"] + flags
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
    # ... and PARITY: the reference-equivalent CPU path (real nanoflann + oracle LM; tests/cpupath.py) on the driver's graph, from the same
    # files' poses, same 4 rounds — the driver's final poses within 1e-9 m / rad of it
    cpu = cpupath.CpuPath(pb["pts"], pb["nor"], src, dst, pb["fixed"], param, plane)
    Pc = pb["init"].copy()
    for _ in range(4):
        Pc, smc = cpu.round(Pc)
    cpu.close()
    assert smc["iterations"] == sm["iterations"], (smc, sm)
    for k in range(5):
        dt, dr = synth.pose_diff(got[k], Pc[k])
        assert dt < 1e-9 and dr < 1e-9, (k, dt, dr)


def test_correspondence_copy_back_through_frame_api(tmp_path):
    """Frame::neighbours[].correspondances / .weight filled by the driver-side adaptor (frame.cpp:110,158,176 semantics) equal the
    engine's lists at the same poses, triple for triple; and Frame::getClosestPoint (S1', through the bound session) returns each
    correspondence's neighbour."""
    pb = synth.make_problem(3, 2000)
    d = tmp_path / "data"; o = tmp_path / "out"
    d.mkdir(); o.mkdir()
    write_dataset(str(d), pb)
    out = subprocess.check_output([os.path.join(BIN, "multiview"), "--dir", str(d), "--out", str(o), "--step", "1", "--rounds", "1", "--quiet", "--copyback",
                                   "--norecomputeNormals", "--drop_phantom_row", "--dump_corr", str(o), "--check_nn", "300"]).decode()
    m = re.search(r"getClosestPoint check: (\d+) queries, (\d+) mismatches", out)
    assert m and int(m.group(1)) >= 1000 and int(m.group(2)) == 0, out
    src, dst = synth.pose_graph_knn(pb["init"], 2, skip_fixed0=False)
    eng = mvicp.Engine(0)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(src, dst)
    counts, weights = eng.correspond(pb["init"], pb["fixed"], 0.05)
    e = 0
    for i in range(3):
        for j in range(2):
            path = os.path.join(str(o), f"corr_{i}_{j}.txt")
            with open(path) as f:
                hdr = f.readline().split()
            assert int(hdr[0]) == dst[e]
            if i == 0:   # fixed frame: never searched (frame.cpp:93) -> list stays empty, weight stays the pose distance of the graph build
                assert int(hdr[2]) == 0 and counts[e] == 0
            else:
                rows = np.loadtxt(path, skiprows=1).reshape(-1, 3)
                gf, gs, gd = eng.get_correspondences(e)
                assert int(hdr[2]) == counts[e] == len(rows)
                assert np.array_equal(rows[:, 0].astype(np.int32), gf) and np.array_equal(rows[:, 1].astype(np.int32), gs) and np.array_equal(rows[:, 2], gd)
                assert np.float32(hdr[1]) == weights[e]
            e += 1
    eng.close()


def test_copy_back_skips_unchanged_lists_and_refills_changed_ones(tmp_path):
    """the Frame mirror refills `neighbours[j].correspondances` only when the library's list changed (mvicp_correspondence_epochs).
    Driver run: round 0 search + solve, rounds 1.. search only (--freeze_from 1: bit-identical poses from round 2 on), before round 3 frame 2 moves.
    Copy statistics per round, and after the last round EVERY list (refilled or kept) equals a fresh engine's at the driver's last poses."""
    pb = synth.make_problem(4, 2500)
    d = tmp_path / "data"; o = tmp_path / "out"
    d.mkdir(); o.mkdir()
    write_dataset(str(d), pb)
    out = subprocess.check_output([os.path.join(BIN, "multiview"), "--dir", str(d), "--out", str(o), "--step", "1", "--rounds", "4", "--quiet", "--copyback", "--copy_stats",
                                   "--norecomputeNormals", "--drop_phantom_row", "--freeze_from", "1", "--perturb_frame", "2", "--perturb_round", "3",
                                   "--dump_corr", str(o)]).decode()
    stats = {int(r): (int(c), int(s)) for r, c, s in re.findall(r"copyback: round (\d+) copied (\d+) skipped (\d+)", out)}
    src, dst = synth.pose_graph_knn(pb["init"], 2, skip_fixed0=False)
    searched = [e for e in range(len(src)) if src[e] != 0]          # frame 0 is fixed: never searched, never filled (frame.cpp:93)
    touched = [e for e in searched if src[e] == 2 or dst[e] == 2]
    assert stats[0] == (len(searched), 0) and stats[1] == (len(searched), 0), stats   # round 1: the solve moved the poses
    assert stats[2] == (0, len(searched)), stats                                        # identical poses: nothing is re-copied
    assert stats[3] == (len(touched), len(searched) - len(touched)) and 0 < len(touched) < len(searched), stats
    P = np.array([np.loadtxt(os.path.join(str(o), f"search_pose_{i}.txt")) for i in range(4)])
    eng = mvicp.Engine(0)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(src, dst)
    counts, weights = eng.correspond(P, pb["fixed"], 0.05)
    e = 0
    for i in range(4):
        for j in range(2):
            path = os.path.join(str(o), f"corr_{i}_{j}.txt")
            with open(path) as f:
                hdr = f.readline().split()
            if i != 0:
                rows = np.loadtxt(path, skiprows=1).reshape(-1, 3)
                gf, gs, gd = eng.get_correspondences(e)
                assert int(hdr[2]) == counts[e] == len(rows)
                assert np.array_equal(rows[:, 0].astype(np.int32), gf) and np.array_equal(rows[:, 1].astype(np.int32), gs) and np.array_equal(rows[:, 2], gd), (i, j)
                assert np.float32(hdr[1]) == weights[e]
            e += 1
    eng.close()


@pytest.mark.parametrize("k", [10, 5])
def test_frame_get_neighbours_equals_nanoflann_knnsearch(tmp_path, k):
    """Frame::getNeighbours(int queryIdx, size_t num_results) (include/frame.h:48, src/internal/frame.cpp:208-231) through the host
    mirror, asked point by point like frame.cpp:249, on EVERY row of the reference's cloudXYZ_0.xyz: the returned points equal
    pts[knnSearch indices] of the REAL nanoflann (golden tests/golden/bunny_knn_full.npz, generated from /root/reference/include/nanoflann.hpp) element for
    element — including the 1185 points whose 10th place is an exact distance tie on this lattice-like scan (tie order = the tree's visiting order).
    k = 5: a 5-list is the head of the 10-list wherever no exact tie spans the 5th place; those rows are compared."""
    K = np.load(os.path.join(ROOT, "tests", "golden", "pairwise_kat.npz"))
    G = np.load(os.path.join(ROOT, "tests", "golden", "bunny_knn_full.npz"))
    pts = K["pts"]
    d = tmp_path / "data"; d.mkdir()
    np.savetxt(str(d / "cloud_0.xyz"), np.hstack([pts, K["nor"]]), fmt="%.17g")
    np.savetxt(str(d / "pose_0.txt"), np.eye(4))
    out = tmp_path / "knn.bin"
    subprocess.check_call([os.path.join(BIN, "multiview"), "--dir", str(d), "--step", "1", "--norecomputeNormals", "--drop_phantom_row", "--quiet",
                           "--dump_knn", str(out), "--knn_k", str(k)])
    got = np.fromfile(str(out), dtype=np.float64).reshape(len(pts), k, 3)
    want = pts[G["knn_idx"][:, :k]]
    if k == 10:
        assert G["tie_at_k"].sum() > 1000
        assert np.array_equal(got, want)
    else:
        e = pts[:, None, :] - pts[G["knn_idx"]]
        d2 = e[..., 0] * e[..., 0] + e[..., 1] * e[..., 1] + e[..., 2] * e[..., 2]     # frame.h:70-76, left to right, no FMA
        clean = d2[:, k - 1] < d2[:, k]                      # no exact tie across the k-th place: the first k of the 10-list ARE the k-list
        assert clean.sum() > 0.8 * len(pts)
        assert np.array_equal(got[clean], want[clean])
        assert np.array_equal(got[:, 0], pts)                 # self first, always
    # out-of-range requests are errors, not garbage
    r = subprocess.run([os.path.join(BIN, "multiview"), "--dir", str(d), "--step", "1", "--norecomputeNormals", "--quiet", "--dump_knn", str(out), "--knn_k", "40"],
                       capture_output=True, text=True)
    assert r.returncode == 2 and "3 <= num_results <= 16" in r.stderr


def test_multiview_driver_default_flags_recompute_normals(tmp_path):
    """Reference defaults (recomputeNormals on): the driver's PCA normals + loop equal the engine's."""
    pb = synth.make_problem(4, 3000)
    d = tmp_path / "data"; o = tmp_path / "out"
    d.mkdir(); o.mkdir()
    write_dataset(str(d), pb)
    subprocess.check_call([os.path.join(BIN, "multiview"), "--dir", str(d), "--out", str(o), "--step", "1", "--rounds", "3", "--quiet", "--drop_phantom_row"])
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
    """bin/pairwise on the reference's own input (all rows of cloudXYZ_0 written back in its .xyz format): the transform it draws is
    the golden P (default-seeded std::mt19937, common.h:36-67) and every solver recovers it — README.md:141-150: Ceres variants
    6-8e-11 / 1.7e-6 deg, closed form 6.6e-15.  Bars: see test_pairwise_known_answer_on_gpu."""
    K = np.load(os.path.join(ROOT, "tests", "golden", "pairwise_kat.npz"))
    cloud = tmp_path / "cloud.xyz"
    np.savetxt(str(cloud), np.hstack([K["pts"], K["nor"]]), fmt="%.17g")
    for extra in ([], ["--pointToPlane"]):
        pfile = tmp_path / "P.txt"
        out = subprocess.check_output([os.path.join(BIN, "pairwise"), "--cloud", str(cloud), "--dump_P", str(pfile), "--precision", "12"] + extra).decode()
        assert np.allclose(np.loadtxt(str(pfile)), K["P"], rtol=0, atol=1e-15)
        vals = re.findall(r"(closed form|ceres \w+)\s+diff_tra:([0-9.e+-]+)\s+diff_rot_degrees:([0-9.e+-]+)", out)
        assert [v[0] for v in vals] == ["closed form", "ceres CeresAngleAxis", "ceres EigenQuaternion", "ceres SophusSE3"], out
        for name, t, r in vals:
            if name == "closed form":
                if not extra:
                    assert float(t) < 1e-13 and float(r) < 3e-6, out      # README.md:148: 6.6e-15, 2.4e-6 deg (acos floor)
                continue   # point-to-plane closed form = ONE linearised step from identity (icp-closedform.cpp:30-54): not exact for this P
            assert float(t) <= 1e-9 and float(r) <= 2e-6, out


def test_pairwise_driver_prints_the_readme_lines(tmp_path):
    """The literal drop-in on the reference's published vector: `pairwise` with the reference's defaults (loadXYZ's duplicated last row)
    and the README run's noise stream (libc++ variate order, --noise_stream libc++) on cloudXYZ_0.xyz prints the README's own lines
    (README.md:141-146, real Ceres): `ceres CeresAngleAxis diff_tra:7.76957e-11`, `ceres EigenQuaternion diff_tra:6.31278e-11` —
    NN-free path: GPU linearize + the product's host LM."""
    K = np.load(os.path.join(ROOT, "tests", "golden", "pairwise_kat.npz"))
    cloud = tmp_path / "cloud.xyz"
    np.savetxt(str(cloud), np.hstack([K["pts"], K["nor"]]), fmt="%.17g")
    pfile = tmp_path / "P.txt"
    out = subprocess.check_output([os.path.join(BIN, "pairwise"), "--cloud", str(cloud), "--noise_stream", "libc++", "--dump_P", str(pfile)]).decode()
    assert np.allclose(np.loadtxt(str(pfile)), K["P_libcxx"], rtol=0, atol=1e-15)
    vals = dict((m[0], m[1]) for m in re.findall(r"(ceres \w+)\s+diff_tra:([0-9.e+-]+)", out))
    assert vals["ceres CeresAngleAxis"] == "7.76957e-11" and vals["ceres EigenQuaternion"] == "6.31278e-11", out


# ---------------------------------------------------------------- the reference's DEFAULT workload (Bunny, 18 views) and cfg1, from committed data
GOLD = os.path.join(ROOT, "tests", "golden", "bunny18.npz")


def write_bunny(d, g, views):
    """tests/golden/bunny18.npz -> the reference's on-disk layout (cloudXYZ_<i>.xyz with `x y z nx ny nz` rows, poses_<i>.txt = ground truth).
    The coordinates are printed with 8 decimals: the same decimal strings (up to trailing zeros) as the reference's sample files, hence
    the same doubles.  The stored normals are zeros: the default flags recompute them (main_multiview.cpp:49,68-70)."""
    off = g["row_off"]
    for k in range(views):
        xyz = g["xyz_e8"][off[k]:off[k + 1]].astype(np.float64) / 1e8
        np.savetxt(os.path.join(d, f"cloudXYZ_{2 * k}.xyz"), np.hstack([xyz, np.zeros_like(xyz)]), fmt="%.8f %.8f %.8f %g %g %g")
        np.savetxt(os.path.join(d, f"poses_{2 * k}.txt"), g["gt"][k], fmt="%.17g")
        # the odd-numbered views exist in the reference's folder and are skipped by --step 2: stand-ins keep the same file indexing
        np.savetxt(os.path.join(d, f"cloudXYZ_{2 * k + 1}.xyz"), np.zeros((3, 6)), fmt="%g")
        np.savetxt(os.path.join(d, f"poses_{2 * k + 1}.txt"), np.eye(4), fmt="%g")


def read_trace(path, K, E, rounds):
    counts = np.zeros((rounds, E), dtype=np.int64); weights = np.zeros((rounds, E), dtype=np.uint32); poses = np.zeros((rounds, K, 4, 4))
    edge_of = {}
    for line in open(path):
        t = line.split()
        if t[0] == "C":
            r, i, j = int(t[1]), int(t[2]), int(t[3])
            e = edge_of.setdefault((i, j), len(edge_of))
            counts[r, e] = int(t[5]); weights[r, e] = int(t[6])
        elif t[0] == "P":
            poses[int(t[1]), int(t[2])] = np.array(t[3:19], dtype=np.float64).reshape(4, 4)
    assert len(edge_of) == E
    return counts, weights.view(np.float32), poses


@pytest.mark.parametrize("case", ["default18", "cfg1"])
def test_reference_default_workload_trajectory(tmp_path, case):
    """bin/multiview with the reference's own defaults (Bunny, limit 40 step 2 -> 18 views, recomputeNormals on, knn 2, cutoff 0.05,
    point-to-plane SophusSE3 robust, 20 rounds, default-seeded noise) — and BASELINE.json's cfg1 (--limit 2 --nopointToPlane: the Bunny pair,
    point-to-point, as a LOOP) — against the recorded trajectory of the reference-equivalent CPU path (real nanoflann + oracle LM,
    tests/golden/make_bunny18.py): per round the kept correspondences per edge and the float edge weights bit for bit, the poses within
    1e-7.  Real scan data: lattice coordinates, ragged cloud sizes (8.5 k - 16.9 k rows), a cutoff that rejects 0.2 - 1.3 % of the queries."""
    g = np.load(GOLD)
    pre = "" if case == "default18" else "cfg1_"
    views = 18 if case == "default18" else 2
    d = tmp_path / "data"; d.mkdir()
    write_bunny(str(d), g, views)
    tr = str(tmp_path / "trace.txt")
    flags = [] if case == "default18" else ["--limit", "2", "--nopointToPlane"]
    subprocess.check_call([os.path.join(BIN, "multiview"), "--dir", str(d), "--quiet", "--trace", tr] + flags, timeout=600)
    E = len(g[pre + "src"])
    counts, weights, poses = read_trace(tr, views, E, 20)
    exp_c, exp_w, exp_p = g[pre + "counts"], g[pre + "weights"], g[pre + "poses"]
    # the fixed frame's own edges are never searched (frame.cpp:93): the driver leaves their graph-build weight in place, the recorded
    # path stores 0 — compare the searched edges
    live = g[pre + "src"] != 0
    assert exp_c[:, ~live].sum() == 0 and counts[:, ~live].sum() == 0
    for r in range(20):
        assert np.array_equal(counts[r, live], exp_c[r, live]), (case, r, np.flatnonzero(counts[r, live] != exp_c[r, live]))
        assert weights[r, live].tobytes() == exp_w[r, live].tobytes(), (case, r)
        for k in range(views):
            dt, dr = synth.pose_diff(poses[r, k], exp_p[r, k])
            assert dt < 1e-7 and dr < 1e-7, (case, r, k, dt, dr)
    # the cutoff does reject queries on this data (the synthetic bench workloads accept every one)
    n_src = np.diff(g["row_off"])[g[pre + "src"][live]]
    if case == "default18":
        assert 0.5 < exp_c[0, live].sum() / n_src.sum() < 1.0
