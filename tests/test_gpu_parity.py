"""-m gpu: the HIP path (through the C ABI) against the oracle and the committed nanoflann goldens.
Bar: bit-exact for indices, squared distances, distances, counts and float weights; fp64 tolerance (stated per
test) for the normal equations and poses."""
import os

import numpy as np
import pytest

import mvicp
import orclib
from mvicp import lib as L
from mvicp import synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "bunny_nn.npz"))
TREE = 102  # NN_GRID with the hash fast path disabled: every query takes the exact AABB-tree descent
METHODS = [L.NN_BRUTE, L.NN_GRID, TREE, L.NN_TILE]


class _Eng(mvicp.Engine):
    """Engine whose nn_method accepts TREE (sets the nn_tree_only option around the call)."""

    def _m(self, method):
        self.set_option("nn_tree_only", 1 if method == TREE else 0)
        return L.NN_GRID if method == TREE else method

    def nn_query(self, frame, queries, nn_method=L.NN_AUTO):
        return super().nn_query(frame, queries, self._m(nn_method))

    def correspond(self, poses, fixed, thresh, nn_method=L.NN_AUTO):
        return super().correspond(poses, fixed, thresh, self._m(nn_method))


@pytest.fixture(scope="module")
def eng():
    e = _Eng(0)
    yield e
    e.close()


# ---------------------------------------------------------------- S1' / NN
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("tag", ["gt", "noisy"])
def test_nn_query_matches_nanoflann_golden(eng, method, tag):
    eng.set_frames([G["dst"]], [G["dst_nor"]])
    idx, d2 = eng.nn_query(0, G["q_" + tag], method)
    assert np.array_equal(idx, G["idx_" + tag])
    assert np.array_equal(d2, G["d2_" + tag])


@pytest.mark.parametrize("method", METHODS)
def test_nn_edge_cases(eng, orc, method):
    rng = np.random.default_rng(3)
    dst = rng.uniform(-0.1, 0.1, (777, 3))  # ragged size (not a tile multiple)
    eng.set_frames([dst], None)
    q = np.vstack([dst[:40], dst[:40] * 9.0, rng.uniform(-0.3, 0.3, (133, 3)), [[5.0, 5.0, 5.0]]])
    idx, d2 = eng.nn_query(0, q, method)
    oi, od = orc.nn_brute(dst, q)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)
    assert np.all(d2[:40] == 0)
    # single-point cloud, single query
    eng.set_frames([dst[:1]], None)
    idx, d2 = eng.nn_query(0, q[:5], method)
    assert np.all(idx == 0)
    # empty query batch is a no-op; empty cloud is an error (nanoflann throws: nanoflann.hpp:904)
    idx, d2 = eng.nn_query(0, np.zeros((0, 3)), method)
    assert len(idx) == 0
    eng.set_frames([np.zeros((0, 3))], None)
    with pytest.raises(mvicp.MvicpError):
        eng.nn_query(0, q[:5], method)


def test_duplicate_targets_follow_nanoflanns_visit_order(eng, orc, refnn):
    """Exact distance ties (duplicated target points): the reference keeps the target its tree VISITS first (nanoflann.hpp:1205-1212,
    1222-1233), which depends on the query — neither the lowest nor the highest index.  Every kernel must return what the REAL
    nanoflann (oracle/_ref) returns; with the rule switched off ("tie_rule" 0) they return the lowest index, like the oracle's scan."""
    assert refnn is not None, "oracle/_ref (real nanoflann) was not built"
    rng = np.random.default_rng(4)
    dst = rng.uniform(-0.1, 0.1, (600, 3))
    dst[300:] = dst[:300][rng.permutation(300)]  # every point exists twice, at unrelated indices
    eng.set_frames([dst], None)
    q = np.vstack([dst[:300] + 1e-4, dst[100:200], rng.uniform(-0.1, 0.1, (50, 3))])   # near a pair, exactly on a pair (d2 = 0 twice), generic
    ri, rd = refnn.query(dst, q)
    oi, od = orc.nn_brute(dst, q)
    assert np.array_equal(rd, od) and not np.array_equal(ri, oi), "the data must make the two tie rules disagree"
    for m in METHODS:
        idx, d2 = eng.nn_query(0, q, m)
        assert np.array_equal(d2, rd) and np.array_equal(idx, ri), (m, int((idx != ri).sum()))
    eng.set_option("tie_rule", 0)
    try:
        for m in METHODS:
            idx, d2 = eng.nn_query(0, q, m)
            assert np.array_equal(idx, oi) and np.array_equal(d2, od)
    finally:
        eng.set_option("tie_rule", 1)


@pytest.mark.parametrize("opts", [{}, {"tile_mfma": 0}, {"tile_mfma": 2, "tile_bounds": 2}, {"nn_cache": 0}, {"tie_lazy": 0}])
@pytest.mark.parametrize("method", [L.NN_BRUTE, L.NN_GRID, L.NN_TILE, L.NN_AUTO])
def test_duplicate_targets_through_the_correspondence_path(refnn, orc, method, opts):
    """The same on the edge path, round after round (seeded, bounds-leaving and cache-aware rounds included): the `second` index of every
    correspondence is the real nanoflann's, the lists / weights / poses follow the reference-equivalent CPU path.  By default (tie_lazy = 1, round 6)
    the reference-equivalent trees do not exist when the first search reports its ties: that search is repeated once after building them —
    {"tie_lazy": 0} is the eager build at mvicp_set_graph."""
    import cpupath
    assert refnn is not None
    pb = synth.make_problem(3, 3000)
    rng = np.random.default_rng(11)
    pts = [p.copy() for p in pb["pts"]]; nor = [n.copy() for n in pb["nor"]]
    for k in range(3):   # a third of every cloud exists twice
        pick = rng.choice(len(pts[k]), len(pts[k]) // 3, replace=False)
        pts[k] = np.vstack([pts[k], pts[k][pick]]); nor[k] = np.vstack([nor[k], nor[k][pick]])
    e = mvicp.Engine(0)
    for k, v in opts.items():
        e.set_option(k, v)
    e.set_frames(pts, nor); e.set_graph(pb["src"], pb["dst"])
    cpu = cpupath.CpuPath(pts, nor, pb["src"], pb["dst"], pb["fixed"], 2, 1, orc=orc, ref=refnn)
    poses = pb["init"].copy()
    disagreements = 0
    for r in range(7):
        c, w = e.correspond(poses, pb["fixed"], 0.05, method)
        corr = cpu.correspond(poses)
        for k in range(e.E):
            gf, gs, gd = e.get_correspondences(k)
            f, s2, d, wk = corr[k]
            assert np.array_equal(gf, f) and np.array_equal(gd, d) and (len(f) == 0 or w[k] == wk), (r, k)
            assert np.array_equal(gs, s2), (r, k, int((gs != s2).sum()))
            if len(f):
                lo = orc.correspond_edge(pts[pb["src"][k]], poses[pb["src"][k]], pts[pb["dst"][k]], poses[pb["dst"][k]], 0.05)[1]
                disagreements += int((lo != s2).sum())
        poses, sm = e.optimize(poses, pb["fixed"], L.PARAM_SOPHUS_SE3, 1, True, 50)
    assert disagreements > 0, "the data must make the two tie rules disagree somewhere"
    cpu.close(); e.close()


# ---------------------------------------------------------------- S1 / correspond
def check_correspond(eng, orc, pts, nor, poses, fixed, src, dst, thresh, method):
    eng.set_frames(pts, nor)
    eng.set_graph(src, dst)
    counts, weights = eng.correspond(poses, fixed, thresh, method)
    for e, (s, d) in enumerate(zip(src, dst)):
        if fixed[s]:
            assert counts[e] == 0
            continue
        f, sec, dist, w, _, _ = orc.correspond_edge(pts[s], poses[s], pts[d], poses[d], thresh)
        gf, gs, gd = eng.get_correspondences(e)
        assert counts[e] == len(f)
        assert np.array_equal(gf, f) and np.array_equal(gs, sec)
        assert np.array_equal(gd, dist)  # bit-exact distances
        if len(f):
            assert weights[e] == w  # bit-exact float weight
        else:
            assert weights[e] == 0
    return counts, weights


@pytest.mark.parametrize("method", METHODS)
def test_correspond_bunny_golden_pair(eng, orc, method):
    pts = [G["dst"], G["src"]]
    nor = [G["dst_nor"], G["src_nor"]]
    poses = np.array([G["pose_dst"], G["pose_src_noisy"]])
    counts, weights = check_correspond(eng, orc, pts, nor, poses, [1, 0], [1], [0], 0.05, method)
    # against the nanoflann golden directly
    f, s, d = eng.get_correspondences(0)
    keep = np.sqrt(G["d2_noisy"]) < float(np.float32(0.05))
    assert np.array_equal(f, np.nonzero(keep)[0]) and np.array_equal(s, G["idx_noisy"][keep])
    assert np.array_equal(d, np.sqrt(G["d2_noisy"][keep]))


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("thresh", [0.05, 0.004, 1e-7])
def test_correspond_synthetic_multiview(eng, orc, method, thresh):
    pb = synth.make_problem(4, 3000)
    check_correspond(eng, orc, pb["pts"], pb["nor"], pb["init"], pb["fixed"], pb["src"], pb["dst"], thresh, method)


def test_correspond_ragged_sizes_and_fixed_sources(eng, orc):
    rng = np.random.default_rng(9)
    pb = synth.make_problem(3, 2500)
    pts = [pb["pts"][0][:2500], pb["pts"][1][:1031], pb["pts"][2][:64]]
    nor = [pb["nor"][0][:2500], pb["nor"][1][:1031], pb["nor"][2][:64]]
    src = np.array([0, 1, 1, 2, 2]); dst = np.array([1, 0, 2, 1, 0])
    check_correspond(eng, orc, pts, nor, pb["init"], [1, 0, 0], src, dst, 0.05, L.NN_BRUTE)
    check_correspond(eng, orc, pts, nor, pb["init"], [1, 0, 1], src, dst, 0.05, L.NN_GRID)
    check_correspond(eng, orc, pts, nor, pb["init"], [1, 0, 0], src, dst, 0.05, TREE)
    check_correspond(eng, orc, pts, nor, pb["init"], [1, 0, 0], src, dst, 0.05, L.NN_TILE)


def test_grid_far_queries_and_clustered_clouds(eng, orc):
    """Queries far outside the cloud, inside holes, and clouds with wildly non-uniform density: the hash block
    test must hand over to the tree exactly when it cannot prove optimality."""
    rng = np.random.default_rng(12)
    a = rng.normal(0, 0.001, (3000, 3))                      # tight cluster
    b = rng.uniform(-0.5, 0.5, (500, 3))                     # sparse halo
    c = np.stack([np.linspace(-1, 1, 700), np.zeros(700), np.zeros(700)], 1)  # a line (degenerate bbox axes)
    dst = np.vstack([a, b, c])
    eng.set_frames([dst], None)
    q = np.vstack([rng.uniform(-2, 2, (2000, 3)), rng.normal(0, 0.002, (2000, 3)), dst[::7] + 1e-9, [[100.0, -50.0, 3.0]], [[0.0, 0.0, 1e6]]])
    oi, od = orc.nn_brute(dst, q)
    for m in (L.NN_GRID, TREE):
        idx, d2 = eng.nn_query(0, q, m)
        assert np.array_equal(idx, oi) and np.array_equal(d2, od), m
    # planar cloud (zero extent along z)
    flat = np.concatenate([rng.uniform(-1, 1, (4000, 2)), np.zeros((4000, 1))], 1)
    eng.set_frames([flat], None)
    q = rng.uniform(-1.2, 1.2, (3000, 3)) * [1, 1, 0.05]
    oi, od = orc.nn_brute(flat, q)
    for m in (L.NN_GRID, TREE):
        idx, d2 = eng.nn_query(0, q, m)
        assert np.array_equal(idx, oi) and np.array_equal(d2, od), m


# ---------------------------------------------------------------- normal equations
@pytest.mark.parametrize("plane", [1, 0])
@pytest.mark.parametrize("robust", [1, 0])
def test_linearize_matches_oracle_blocks(eng, orc, plane, robust):
    pb = synth.make_problem(4, 6000)
    eng.set_frames(pb["pts"], pb["nor"])
    eng.set_graph(pb["src"], pb["dst"])
    counts, weights = eng.correspond(pb["init"], pb["fixed"], 0.05)
    corr = [eng.get_correspondences(e)[:2] for e in range(eng.E)]
    for poses in (pb["init"], pb["gt"]):
        got = eng.linearize(poses, plane, robust)
        want = orc.edge_blocks(pb["pts"], pb["nor"], pb["src"], pb["dst"], corr, weights, poses, plane, robust)
        # fp64 sums of ~6000 terms in a different association order: relative 1e-11 of the block's scale
        for e in range(eng.E):
            scale = np.abs(want[e, :78]).max()
            assert np.allclose(got[e, :78], want[e, :78], rtol=0, atol=1e-11 * scale), (e, np.abs(got[e, :78] - want[e, :78]).max() / scale)
            gs = np.abs(want[e, 78:90]).max() + 1e-300
            assert np.allclose(got[e, 78:90], want[e, 78:90], rtol=0, atol=1e-10 * gs)
            assert abs(got[e, 90] - want[e, 90]) <= 1e-12 * abs(want[e, 90])


def test_linearize_explicit_correspondences_and_chunk_boundaries(eng, orc):
    # counts straddling the 4096-correspondence workgroup chunk and odd counts (the 16-B pair loads)
    rng = np.random.default_rng(5)
    N = 9001
    pb = synth.make_problem(2, N)
    eng.set_frames(pb["pts"], pb["nor"])
    eng.set_graph([1], [0])
    for n in (1, 2, 4095, 4096, 4097, 8193, 9001):
        f = np.sort(rng.choice(N, n, replace=False)).astype(np.int32)
        s = rng.integers(0, N, n).astype(np.int32)
        eng.set_correspondences(0, f, s, 0.013)
        got = eng.linearize(pb["init"], 1, 1)
        want = orc.edge_blocks(pb["pts"], pb["nor"], [1], [0], [(f, s)], [np.float32(0.013)], pb["init"], 1, 1)
        scale = np.abs(want[0, :78]).max()
        assert np.allclose(got[0], want[0], rtol=1e-10, atol=1e-11 * scale), n
    eng.set_correspondences(0, np.zeros(0, np.int32), np.zeros(0, np.int32), 0.0)
    assert np.all(eng.linearize(pb["init"], 1, 1) == 0)


# ---------------------------------------------------------------- S2 and the whole loop
@pytest.mark.parametrize("param", [L.PARAM_SOPHUS_SE3, L.PARAM_ANGLE_AXIS, L.PARAM_EIGEN_QUATERNION])
@pytest.mark.parametrize("plane", [1, 0])
def test_optimize_matches_oracle(eng, orc, param, plane):
    pb = synth.make_problem(4, 5000)
    eng.set_frames(pb["pts"], pb["nor"])
    eng.set_graph(pb["src"], pb["dst"])
    counts, weights = eng.correspond(pb["init"], pb["fixed"], 0.05)
    corr = [eng.get_correspondences(e)[:2] for e in range(eng.E)]
    P, sm = eng.optimize(pb["init"], pb["fixed"], param, plane, True, 50)
    prob = orc.make_problem(pb["pts"], pb["nor"], pb["fixed"], pb["src"], pb["dst"], corr, weights, param, plane, 1)
    P_ref, sm_ref = orc.optimize(prob, pb["init"], 50)
    assert sm["iterations"] == sm_ref["iterations"] and sm["termination"] == sm_ref["termination"], (sm, sm_ref)
    for k in range(4):
        dt, dr = synth.pose_diff(P[k], P_ref[k])
        assert dt < 1e-9 and dr < 1e-9, (k, dt, dr)  # north-star tolerance is 1e-5 m / 1e-5 rad


def run_icp_gpu(eng, pb, rounds, param, plane, method=L.NN_AUTO):
    poses = pb["init"].copy()
    for _ in range(rounds):
        eng.correspond(poses, pb["fixed"], 0.05, method)
        poses, sm = eng.optimize(poses, pb["fixed"], param, plane, True, 50)
    return poses


def run_icp_oracle(orc, pb, rounds, param, plane):
    poses = pb["init"].copy()
    for _ in range(rounds):
        corr, w = [], []
        for s, d in zip(pb["src"], pb["dst"]):
            f, sec, dist, wt, _, _ = orc.correspond_edge(pb["pts"][s], poses[s], pb["pts"][d], poses[d], 0.05)
            corr.append((f, sec)); w.append(wt)
        prob = orc.make_problem(pb["pts"], pb["nor"], pb["fixed"], pb["src"], pb["dst"], corr, w, param, plane, 1)
        poses, sm = orc.optimize(prob, poses, 50)
    return poses


@pytest.mark.parametrize("param,plane", [(L.PARAM_SOPHUS_SE3, 1), (L.PARAM_ANGLE_AXIS, 1), (L.PARAM_EIGEN_QUATERNION, 0)])
def test_full_icp_loop_matches_oracle(eng, orc, param, plane):
    """The loop body of main_multiview.cpp:150-169, 6 rounds, GPU vs the CPU restatement on the same inputs."""
    pb = synth.make_problem(4, 4000)
    eng.set_frames(pb["pts"], pb["nor"])
    eng.set_graph(pb["src"], pb["dst"])
    Pg = run_icp_gpu(eng, pb, 6, param, plane)
    Po = run_icp_oracle(orc, pb, 6, param, plane)
    for k in range(4):
        dt, dr = synth.pose_diff(Pg[k], Po[k])
        assert dt < 1e-8 and dr < 1e-8, (k, dt, dr)
    if plane:  # (point-to-point on this near-spherical scene slides along the surface: parity holds, accuracy does not)
        e0 = max(synth.pose_diff(pb["init"][k], pb["gt"][k])[0] for k in range(4))
        e1 = max(synth.pose_diff(Pg[k], pb["gt"][k])[0] for k in range(4))
        assert e1 < 0.6 * e0, (e0, e1)


KAT = np.load(os.path.join(os.path.dirname(__file__), "golden", "pairwise_kat.npz"))


@pytest.mark.parametrize("plane", [0, 1])
def test_pairwise_known_answer_on_gpu(eng, orc, plane):
    """The reference's only known-answer test on its own inputs (main_pairwise.cpp:34-61,117-133: all rows of cloudXYZ_0, P from the
    default-seeded std::mt19937) through the device path (S3 = a 2-frame graph with index-aligned pairs).  README.md:141-146 quotes
    diff_tra 6-8e-11 / diff_rot 1.7e-6 deg; the solve ends on the parameter tolerance without taking the last step, so the error is
    the size of that step (< 1e-8 |x|): bar = one decade above the README sample, same termination, and the GPU trajectory equal
    to the oracle's (same iteration count, poses within 1e-11)."""
    pts, nrm, P = KAT["pts"], KAT["nor"], KAT["P"]
    dstp = pts @ P[:3, :3].T + P[:3, 3]
    dstn = nrm @ P[:3, :3].T
    n = len(pts)
    ids = np.arange(n, dtype=np.int32)
    eng.set_frames([dstp, pts], [dstn, nrm])
    eng.set_graph([1], [0])
    eng.set_correspondences(0, ids, ids, 0.0)
    for param in (0, 1, 2):
        Pout, sm = eng.optimize(np.array([np.eye(4), np.eye(4)]), [1, 0], param, plane, False, 50)
        dt, dr_deg = orc.pose_diff(P, Pout[1])            # the reference's poseDiff (acos form, degrees)
        assert sm["termination"] == 2, sm
        assert dt <= 1e-9 and dr_deg <= 2e-6, (param, plane, dt, dr_deg, sm)
        assert dt <= 1.25 * KAT["reached_dt"][plane, param] + 1e-13, (param, plane, dt)   # the recorded value, not only the bar
        prob = orc.make_problem([dstp, pts], [dstn, nrm], [1, 0], [1], [0], [(ids, ids)], [0.0], param, plane, 0)
        Pref, smr = orc.optimize(prob, np.array([np.eye(4), np.eye(4)]), 50)
        assert sm["iterations"] == smr["iterations"], (sm, smr)
        dt, dr = synth.pose_diff(Pout[1], Pref[1])
        assert dt < 1e-11 and dr < 1e-11, (param, plane, dt, dr)


def test_pairwise_readme_vector_on_gpu(eng, orc):
    """The reference's PUBLISHED numbers through the device path: README.md:141-146 (real Ceres, point-to-point) angle-axis diff_tra
    7.76957e-11, quaternion 6.31278e-11, on the reference's own inputs (cloudXYZ_0 with loadXYZ's duplicated last row, P from the
    libc++ noise stream; tests/test_oracle_lm.py::test_readme_known_answer_reproduced).  GPU linearize + host/lm.cpp."""
    pts = np.vstack([KAT["pts"], KAT["pts"][-1:]])
    P = KAT["P_libcxx"]
    dstp = pts @ P[:3, :3].T + P[:3, 3]
    ids = np.arange(len(pts), dtype=np.int32)
    eng.set_frames([dstp, pts], None)
    eng.set_graph([1], [0])
    eng.set_correspondences(0, ids, ids, 0.0)
    for param in (L.PARAM_ANGLE_AXIS, L.PARAM_EIGEN_QUATERNION):
        Pout, sm = eng.optimize(np.array([np.eye(4), np.eye(4)]), [1, 0], param, 0, False, 50)
        dt = orc.pose_diff(P, Pout[1])[0]
        assert sm["termination"] == 2 and sm["iterations"] == 6, sm
        assert abs(dt / KAT["readme_dt"][param] - 1) < 1e-4, (param, dt, KAT["readme_dt"][param])


def test_point_to_point_block_against_kabsch(eng):
    """Row f4: the closed-form point-to-point solution (icp-closedform.cpp:9-26, here SVD in numpy AND the library's host solver)
    is the exact minimiser of sum |R p + t - q|^2, so it must be a stationary point of the GPU's point-to-point normal equations
    (non-robust): g = 0 there, and the LM solve from identity must end next to it."""
    rng = np.random.default_rng(21)
    pts, P = KAT["pts"][::2], KAT["P"]
    dstp = pts @ P[:3, :3].T + P[:3, 3] + rng.normal(0, 1e-3, pts.shape)
    pm, qm = pts.mean(0), dstp.mean(0)
    U, S, Vt = np.linalg.svd((dstp - qm).T @ (pts - pm))
    R = U @ Vt
    assert np.linalg.det(R) > 0
    Tk = np.eye(4); Tk[:3, :3] = R; Tk[:3, 3] = qm - R @ pm
    Th = L.closedform_point_to_point(pts, dstp)
    assert np.allclose(Th, Tk, atol=1e-12)
    ids = np.arange(len(pts), dtype=np.int32)
    eng.set_frames([dstp, pts], None); eng.set_graph([1], [0])
    eng.set_correspondences(0, ids, ids, 0.0)
    H, g, cost = L.unpack_block(eng.linearize(np.array([np.eye(4), Tk]), 0, 0)[0])
    assert np.isclose(cost, 0.5 * np.sum((pts @ R.T + Tk[:3, 3] - dstp) ** 2), rtol=1e-12)
    gscale = np.sqrt(np.diag(H)[:6]) * np.sqrt(2 * cost)           # |J_col| |r|: the size each gradient entry would have off-optimum
    assert np.all(np.abs(g[:6]) < 1e-9 * gscale), (g[:6], gscale)
    for param in (0, 1, 2):
        Pout, sm = eng.optimize(np.array([np.eye(4), np.eye(4)]), [1, 0], param, 0, False, 50)
        dt, dr = synth.pose_diff(Pout[1], Tk)
        assert dt < 1e-5 and dr < 1e-5, (param, dt, dr, sm)       # function tolerance 1e-6 on a non-zero-residual problem


def test_recompute_normals_after_correspond_regathers(eng):
    """normals recomputed AFTER the search must reach the next evaluation (the operand stream bakes n and n.q in)."""
    pb = synth.make_problem(3, 3000)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    eng.correspond(pb["init"], pb["fixed"], 0.05)
    nor = [eng.recompute_normals(i, 10) for i in range(3)]       # overwrites the analytic normals on the device
    got = eng.linearize(pb["init"], 1, 1)
    fresh = mvicp.Engine(0)
    fresh.set_frames(pb["pts"], nor); fresh.set_graph(pb["src"], pb["dst"])
    fresh.correspond(pb["init"], pb["fixed"], 0.05)
    want = fresh.linearize(pb["init"], 1, 1)
    fresh.close()
    assert np.array_equal(got, want)
    # and the next round (lists unchanged, so they would be reused) still sees the new normals
    eng.correspond(pb["init"], pb["fixed"], 0.05)
    assert np.array_equal(eng.linearize(pb["init"], 1, 1), want)


def test_lists_survive_recompute_normals_and_epochs_track_changes(eng, orc):
    """correspond -> recompute_normals -> get / map_correspondences used to fail ('export of edge holds 0 triples') because the export
    keyed on list_valid, which recompute_normals clears.  The lists of the last search are still on the device: they must come back, equal to
    the oracle's.  And mvicp_correspondence_epochs: an edge keeps its epoch exactly when its list is provably the same (bit-identical poses,
    same cutoff); the mapped buffer then stays valid without device work, and still holds the right triples."""
    pb = synth.make_problem(4, 3000)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    E = eng.E

    def check_lists(P, cutoff=0.05):
        t, off = eng.map_correspondences()
        for e, (s_, d_) in enumerate(zip(pb["src"], pb["dst"])):
            f, sec, dist, w, _, _ = orc.correspond_edge(pb["pts"][s_], P[s_], pb["pts"][d_], P[d_], cutoff)
            seg = t[off[e]:off[e + 1]]
            assert np.array_equal(seg["first"], f) and np.array_equal(seg["second"], sec) and np.array_equal(seg["dist"], dist), e
            gf, gs, gd = eng.get_correspondences(e)
            assert np.array_equal(gf, f) and np.array_equal(gs, sec) and np.array_equal(gd, dist), e

    P = pb["init"].copy()
    eng.correspond(P, pb["fixed"], 0.05)
    ep0 = eng.correspondence_epochs()
    assert len(set(ep0.tolist())) == E and (ep0 > 0).all()
    for i in range(4):
        eng.recompute_normals(i, 10)
    check_lists(P)                                            # (export BEFORE any get: the failing sequence of ADVICE r5)
    assert np.array_equal(eng.correspondence_epochs(), ep0)   # normals do not touch the triples
    # identical poses: every list is provably the same -> epochs kept, lists still right (whatever kernel AUTO picks; three rounds cross its regimes)
    for _ in range(3):
        eng.correspond(P, pb["fixed"], 0.05)
        assert np.array_equal(eng.correspondence_epochs(), ep0)
        check_lists(P)
    # one frame moves: exactly the edges that touch it get new epochs
    P2 = P.copy(); P2[3, 0, 3] += 1e-4
    eng.correspond(P2, pb["fixed"], 0.05)
    ep1 = eng.correspondence_epochs()
    touched = np.array([(s_ == 3 or d_ == 3) for s_, d_ in zip(pb["src"], pb["dst"])])
    assert touched.any() and not touched.all()
    assert (ep1[touched] > ep0.max()).all() and np.array_equal(ep1[~touched], ep0[~touched])
    check_lists(P2)
    # a different cutoff changes every list's inputs; so does reset_history and an explicit list
    eng.correspond(P2, pb["fixed"], 0.04)
    ep2 = eng.correspondence_epochs()
    assert (ep2 > ep1.max()).all()
    check_lists(P2, 0.04)
    eng.reset_history()
    eng.correspond(P2, pb["fixed"], 0.04)
    ep3 = eng.correspondence_epochs()
    assert (ep3 > ep2.max()).all()
    check_lists(P2, 0.04)
    f0, s0, _ = eng.get_correspondences(0)
    eng.set_correspondences(0, f0[:10], s0[:10], 0.01)
    ep4 = eng.correspondence_epochs()
    assert ep4[0] > ep3.max() and np.array_equal(ep4[1:], ep3[1:])
    eng.correspond(P2, pb["fixed"], 0.04)                      # the search replaces the explicit list: a new epoch again, the others stay
    ep5 = eng.correspondence_epochs()
    assert ep5[0] > ep4[0] and np.array_equal(ep5[1:], ep3[1:])
    check_lists(P2, 0.04)


def test_failed_structure_build_is_sticky_and_never_a_silent_brute_force(eng, orc):
    """a failed background build used to be reported once and forgotten — a retried mvicp_set_graph succeeded and every edge touching
    that frame silently fell back to the O(N^2) brute-force kernel.  Now the frame stays failed (with its message) until it is uploaded again."""
    pb = synth.make_problem(3, 2000)
    for async_build in (1, 0):
        eng.set_option("async_build", async_build)
        eng.set_option("fault_inject_build", 2)                    # the second build from now fails
        if async_build:
            eng.set_frames(pb["pts"], pb["nor"])                   # (the upload returns; the failure surfaces at the first call that needs the structures)
        else:
            with pytest.raises(mvicp.MvicpError, match="injected structure-build failure"):
                eng.set_frames(pb["pts"], pb["nor"])
            eng.npts = [len(p) for p in pb["pts"]]
            eng.set_frame(2, pb["pts"][2], pb["nor"][2])           # (the loop stopped at frame 1: finish the upload)
        for _ in range(2):                                         # sticky: the retry fails the same way
            with pytest.raises(mvicp.MvicpError, match="frame 1: injected structure-build failure"):
                eng.set_graph(pb["src"], pb["dst"])
        with pytest.raises(mvicp.MvicpError, match="frame 1"):
            eng.nn_query(0, pb["pts"][0][:8])
        eng.set_frame(1, pb["pts"][1], pb["nor"][1])               # a new upload of that slot clears it
        eng.set_graph(pb["src"], pb["dst"])
        counts, weights = eng.correspond(pb["init"], pb["fixed"], 0.05)
        for e, (s_, d_) in enumerate(zip(pb["src"], pb["dst"])):
            f, sec, dist, w, _, _ = orc.correspond_edge(pb["pts"][s_], pb["init"][s_], pb["pts"][d_], pb["init"][d_], 0.05)
            gf, gs, gd = eng.get_correspondences(e)
            assert np.array_equal(gf, f) and np.array_equal(gs, sec) and np.array_equal(gd, dist) and weights[e] == w
    eng.set_option("async_build", 1)


def test_fixed_source_edges_are_excluded_from_the_solve(eng, orc):
    """ADVICE r1 / icp-ceres.cpp:255,351,426: correspond() with NO fixed mask fills the edges out of frame 0 too; the solve
    (which forces fixed[0]) must ignore them like the reference does."""
    pb = synth.make_problem(4, 3000)
    src, dst = synth.pose_graph_knn(pb["init"], 2, skip_fixed0=False)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(src, dst)
    counts, weights = eng.correspond(pb["init"], np.zeros(4, np.uint8), 0.05)
    assert all(counts[e] > 0 for e in range(len(src)))            # edges out of frame 0 were searched
    corr = [eng.get_correspondences(e)[:2] for e in range(eng.E)]
    P, sm = eng.optimize(pb["init"], np.zeros(4, np.uint8), L.PARAM_SOPHUS_SE3, 1, True, 50)
    keep = [e for e in range(len(src)) if src[e] != 0]
    prob = orc.make_problem(pb["pts"], pb["nor"], pb["fixed"], src[keep], dst[keep], [corr[e] for e in keep], weights[keep], orclib.PARAM_SOPHUS, 1, 1)
    P_ref, sm_ref = orc.optimize(prob, pb["init"], 50)
    assert sm["iterations"] == sm_ref["iterations"]
    for k in range(4):
        dt, dr = synth.pose_diff(P[k], P_ref[k])
        assert dt < 1e-9 and dr < 1e-9, (k, dt, dr)


# ---------------------------------------------------------------- size-independent properties at full size
def test_full_size_properties_cfg2(eng):
    """BASELINE config 2 size (2 x 100k, point-to-plane): NN of a cloud against itself is the identity with d2 = 0;
    brute force and grid agree bit-for-bit; linearize is invariant to a common rigid motion of both poses."""
    pb = synth.make_problem(2, 100_000)
    eng.set_frames(pb["pts"], pb["nor"])
    eng.set_graph([1], [0])
    for m in (L.NN_GRID, TREE):
        idx, d2 = eng.nn_query(1, pb["pts"][1][:20000], m)
        assert np.array_equal(idx, np.arange(20000)) and np.all(d2 == 0)
    c1, w1 = eng.correspond(pb["init"], pb["fixed"], 0.05, L.NN_BRUTE)
    a = eng.get_correspondences(0)
    for m in (L.NN_GRID, TREE, L.NN_TILE):
        c2, w2 = eng.correspond(pb["init"], pb["fixed"], 0.05, m)
        b = eng.get_correspondences(0)
        assert c1[0] == c2[0] and w1[0] == w2[0] and all(np.array_equal(x, y) for x, y in zip(a, b)), m
    assert np.all(np.diff(a[0]) > 0)  # ascending source index
    blk = eng.linearize(pb["init"], 1, 1)
    T = np.eye(4); T[:3, :3] = synth.so3_exp(np.array([0.3, -0.2, 0.5])); T[:3, 3] = [0.2, 0.1, -0.3]
    moved = np.array([T @ P for P in pb["init"]])
    blk2 = eng.linearize(moved, 1, 1)
    scale = np.abs(blk[0, :78]).max()
    assert np.allclose(blk, blk2, rtol=1e-9, atol=1e-12 * scale)


# ---------------------------------------------------------------- multi-GPU plumbing that a 1-GPU box can exercise
def test_rccl_communicator_single_rank(orc):
    """RCCL path with world = 1: the all-reduce of the per-edge blocks and of the counts/medians must be an exact
    identity (the real N > 1 exchange is covered by tests/test_gloo_shard.py on CPU and by the driver's 8-GPU run)."""
    import torch
    rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    rccl = rccl if os.path.exists(rccl) else None
    pb = synth.make_problem(3, 3000)
    a = mvicp.Engine(0)
    a.set_frames(pb["pts"], pb["nor"]); a.set_graph(pb["src"], pb["dst"])
    ca, wa = a.correspond(pb["init"], pb["fixed"], 0.05)
    ba = a.linearize(pb["init"], 1, 1)
    b = mvicp.Engine(0, rank=0, world=1)
    b.set_frames(pb["pts"], pb["nor"]); b.set_graph(pb["src"], pb["dst"])
    b.comm_init(mvicp.Engine.comm_unique_id(rccl), rccl)
    cb, wb = b.correspond(pb["init"], pb["fixed"], 0.05)
    bb = b.linearize(pb["init"], 1, 1)
    assert np.array_equal(ca, cb) and np.array_equal(wa, wb) and np.array_equal(ba, bb)
    Pa, _ = a.optimize(pb["init"], pb["fixed"]); Pb, _ = b.optimize(pb["init"], pb["fixed"])
    assert np.array_equal(Pa, Pb)
    # later rounds: ONE collective per search — [queued first evaluation's blocks | counts | medians | armed | scales] — and one per
    # further LM evaluation (device buffer -> all-reduce -> pinned copy); through the moving rounds, the bracket select and the fixed point
    b.profile(True)
    rounds = 12
    for r in range(rounds):
        b.profile_reset()
        ca, wa = a.correspond(Pa, pb["fixed"], 0.05); cb, wb = b.correspond(Pb, pb["fixed"], 0.05)
        assert np.array_equal(ca, cb) and wa.tobytes() == wb.tobytes(), r
        assert b.profile_get("comm")[1] == 1, (r, b.profile_get("comm"))                 # the search: exactly one exchange
        Pa, sa = a.optimize(Pa, pb["fixed"]); Pb, sb = b.optimize(Pb, pb["fixed"])
        assert np.array_equal(Pa, Pb) and sa == sb, r
        assert b.profile_get("spec.hit")[1] == 1, r
        assert b.profile_get("comm")[1] == 1 + sb["evaluations"] - 1, (r, sb)            # + one per evaluation the queued launch did not serve
    assert sb["successful_steps"] == 0                                                   # the loop reached the fixed point
    # a rank whose history is gone (new registration) does not arm: same collective, nothing queued, same results
    a.reset_history(); b.reset_history()
    ca, wa = a.correspond(pb["init"], pb["fixed"], 0.05); cb, wb = b.correspond(pb["init"], pb["fixed"], 0.05)
    assert np.array_equal(ca, cb) and wa.tobytes() == wb.tobytes()
    Pa, sa = a.optimize(pb["init"], pb["fixed"]); Pb, sb = b.optimize(pb["init"], pb["fixed"])
    assert np.array_equal(Pa, Pb) and sa == sb
    a.close(); b.close()


def test_reset_history_replays_the_registration_bit_for_bit():
    """mvicp_reset_history = a fresh run of the reference program on the same clouds and graph: a second registration from the same
    initial poses walks through exactly the same poses, counts, weights and LM summaries as the first (no cross-registration state)."""
    pb = synth.make_problem(4, 6000)
    eng = mvicp.Engine(0)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    runs = []
    for reg in range(2):
        if reg:
            eng.reset_history()
        P = pb["init"].copy()
        log = []
        for _ in range(10):
            c, w = eng.correspond(P, pb["fixed"], 0.05)
            P, sm = eng.optimize(P, pb["fixed"])
            log.append((c.copy(), w.copy(), P.copy(), sm))
        runs.append(log)
    for x, y in zip(*runs):
        assert np.array_equal(x[0], y[0]) and x[1].tobytes() == y[1].tobytes() and np.array_equal(x[2], y[2]) and x[3] == y[3]
    eng.close()


def test_sharded_contexts_sum_to_the_unsharded_blocks():
    """Two contexts on one GPU playing rank 0 and rank 1 of a 2-way shard: each fills only its owned edge slots
    (zeros elsewhere); their sum equals the single-context blocks BIT FOR BIT — the property that makes the RCCL
    sum exact and the poses independent of the GPU count."""
    pb = synth.make_problem(5, 5000)
    full = mvicp.Engine(0)
    full.set_frames(pb["pts"], pb["nor"]); full.set_graph(pb["src"], pb["dst"])
    cf, wf = full.correspond(pb["init"], pb["fixed"], 0.05)
    bf = full.linearize(pb["init"], 1, 1)
    parts, counts = [], []
    for r in range(2):
        e = mvicp.Engine(0, rank=r, world=2)
        e.set_frames(pb["pts"], pb["nor"]); e.set_graph(pb["src"], pb["dst"])
        c, w = e.correspond(pb["init"], pb["fixed"], 0.05)
        parts.append(e.linearize(pb["init"], 1, 1)); counts.append(c)
        e.close()
    own = L.edge_owner([len(pb["pts"][s]) for s in pb["src"]], 2)
    for e in range(len(own)):
        assert np.all(parts[1 - own[e]][e] == 0)
    assert np.array_equal(parts[0] + parts[1], bf)
    assert np.array_equal(counts[0] + counts[1], cf)
    full.close()


# ---------------------------------------------------------------- f1: Frame::recomputeNormals
def _pca_normals(pts, knn_idx):
    nb = pts[knn_idx]                                   # (n, k, 3) in nanoflann's result order
    c = nb - nb.mean(axis=1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", c, c)
    w, v = np.linalg.eigh(cov)
    ref = v[:, :, 0]
    return np.where(ref[:, 2:3] > 0, -ref, ref), w


def _check_normals(nrm, ref, w):
    well = (w[:, 1] - w[:, 0]) > 1e-6 * w[:, 2]        # skip (near-)degenerate smallest eigenvalues
    dots = np.abs(np.sum(nrm * ref, axis=1))
    assert np.all(dots[well] > 1 - 1e-9), dots[well].min()
    assert np.all(nrm[:, 2] <= 0) and np.allclose(np.linalg.norm(nrm, axis=1), 1, atol=1e-12)
    sure = well & (np.abs(ref[:, 2]) > 1e-6)
    assert np.all(np.sum(nrm * ref, axis=1)[sure] > 0)


def test_recompute_normals_matches_nanoflann_knn_and_pca(eng):
    """frame.cpp:244-255 + common.h:331-346: 10-NN (self included) from the REAL nanoflann (golden bunny_knn.npz: every 4th row of
    cloudXYZ_0), PCA in numpy.  The k-NN LISTS must be nanoflann's element for element — equal distances in the order its tree visits
    them (csrc/kdvisit.h) — and the normals agree to 1e-9 (iterative eigen-solvers differ in rounding only)."""
    K = np.load(os.path.join(os.path.dirname(__file__), "golden", "bunny_knn.npz"))
    pts, gi, gd = K["pts"], K["knn_idx"], K["knn_d2"]
    eng.set_frames([pts], None)
    nrm, knn = eng.recompute_normals(0, 10, want_knn=True)
    e = pts[:, None, :] - pts[knn]
    myd = (e[:, :, 0] * e[:, :, 0] + e[:, :, 1] * e[:, :, 1]) + e[:, :, 2] * e[:, :, 2]
    assert np.array_equal(myd, gd)                     # distances bit-equal, already in ascending order
    assert np.all(knn[:, 0] == np.arange(len(pts)))    # self first (distance 0)
    assert np.array_equal(knn, gi)
    ref, w = _pca_normals(pts, gi)
    _check_normals(nrm, ref, w)


def test_recompute_normals_full_cloud_matches_nanoflann_lists(eng):
    """Row f1 on the reference's own lattice data: EVERY row of samples/Bunny_RealData/cloudXYZ_0.xyz (16 264 points; z is quantised to
    1 mm, so 1185 points have an exact tie at the 10th place and a lowest-index rule would pick another neighbour set for 3.6 % of the
    points — normals up to 20 degrees apart, final multiview poses 4.5e-5 apart).  The device lists equal nanoflann's knnSearch lists
    (golden bunny_knn_full.npz) element for element: 0 differing sets, 0 differing orders."""
    pts = KAT["pts"]
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "bunny_knn_full.npz"))
    eng.set_frames([pts], None)
    nrm, knn = eng.recompute_normals(0, 10, want_knn=True)
    assert int(G["tie_at_k"].sum()) == 1185
    differing_sets = int((~np.all(np.sort(knn, axis=1) == np.sort(G["knn_idx"], axis=1), axis=1)).sum())
    differing_lists = int((~np.all(knn == G["knn_idx"], axis=1)).sum())
    assert differing_sets == 0 and differing_lists == 0, (differing_sets, differing_lists)
    ref, w = _pca_normals(pts, G["knn_idx"])
    _check_normals(nrm, ref, w)
    ang = np.degrees(np.arccos(np.clip(np.abs(np.sum(nrm * ref, axis=1)), 0, 1)))
    print(f"f1 full cloud: differing 10-NN sets 0 / {len(pts)}, normal angle vs numpy-eigh PCA of nanoflann's sets: max {ang.max():.2e} deg")


def test_recomputed_normals_feed_point_to_plane(eng, orc):
    """The ICP path uses the device normals written by mvicp_recompute_normals (reference default, main_multiview.cpp:49)."""
    pb = synth.make_problem(3, 4000)
    eng.set_frames(pb["pts"], None)
    nor = [eng.recompute_normals(i, 10) for i in range(3)]
    eng.set_graph(pb["src"], pb["dst"])
    counts, weights = eng.correspond(pb["init"], pb["fixed"], 0.05)
    corr = [eng.get_correspondences(e)[:2] for e in range(eng.E)]
    got = eng.linearize(pb["init"], 1, 1)
    want = orc.edge_blocks(pb["pts"], nor, pb["src"], pb["dst"], corr, weights, pb["init"], 1, 1)
    scale = np.abs(want[:, :78]).max()
    assert np.allclose(got, want, rtol=1e-9, atol=1e-11 * scale)
    # synthetic analytic normals vs PCA normals: same surface, so they agree up to sign / sampling noise
    ang = np.abs(np.sum(nor[1] * pb["nor"][1], axis=1))
    assert np.median(ang) > 0.99


def test_linearised_point_to_plane_matches_closed_form(eng):
    """icp-closedform.cpp:30-54 builds the 6x6 linearised point-to-plane normal equations C x = d at identity; they are exactly
    the source block of K5's output at identity poses without the robust loss: H_ss = sum [n; p x n][n; p x n]^T."""
    rng = np.random.default_rng(2)
    n = 5000
    src = rng.normal(0, 0.1, (n, 3)); dst = src + rng.normal(0, 0.002, (n, 3))
    nor = rng.normal(0, 1, (n, 3)); nor /= np.linalg.norm(nor, axis=1, keepdims=True)
    eng.set_frames([dst, src], [nor, nor]); eng.set_graph([1], [0])
    ids = np.arange(n, dtype=np.int32)
    eng.set_correspondences(0, ids, ids, 0.0)
    H, g, cost = L.unpack_block(eng.linearize(np.array([np.eye(4), np.eye(4)]), 1, 0)[0])
    u = np.hstack([nor, np.cross(src, nor)])
    r = np.sum((src - dst) * nor, axis=1)
    assert np.allclose(H[:6, :6], u.T @ u, rtol=1e-11) and np.allclose(g[:6], u.T @ r, rtol=1e-9, atol=1e-15)
    assert np.isclose(cost, 0.5 * np.sum(r * r), rtol=1e-12)


# ---------------------------------------------------------------- temporal NN cache
def test_temporal_cache_is_bit_identical_to_full_search(orc):
    """From the second grid search on, a query whose previous neighbour is provably still nearest (re-evaluated distance
    below the stored lower bound on all other targets minus the pose-induced displacement) skips the search.  Every round
    must give exactly the lists a fresh full search gives, and the oracle's."""
    pb = synth.make_problem(4, 6000)
    engs = []
    for cache in (1, 0):
        e = mvicp.Engine(0)
        e.set_option("nn_cache", cache)
        e.set_option("list_reuse", cache)  # reference engine: search everything, re-compact and re-gather every round
        e.set_frames(pb["pts"], pb["nor"]); e.set_graph(pb["src"], pb["dst"])
        e.profile(True); e.set_option("nn_census", 1)
        engs.append(e)
    poses = pb["init"].copy()
    hits = []
    for r in range(9):
        res = []
        for e in engs:
            e.profile_reset()
            c, w = e.correspond(poses, pb["fixed"], 0.05, L.NN_GRID)
            res.append((c, w, [e.get_correspondences(k) for k in range(e.E)], e.nn_census()))
        (c1, w1, l1, s1), (c0, w0, l0, s0) = res
        assert np.array_equal(c1, c0) and np.array_equal(w1, w0), r
        for a, b in zip(l1, l0):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), r
        hits.append(1.0 - s1["hits"] / s1["queries"])
        if r in (0, 8):
            for k, (s, d) in enumerate(zip(pb["src"], pb["dst"])):
                f, sec, dist, wt, _, _ = orc.correspond_edge(pb["pts"][s], poses[s], pb["pts"][d], poses[d], 0.05)
                assert np.array_equal(l1[k][0], f) and np.array_equal(l1[k][1], sec) and np.array_equal(l1[k][2], dist) and w1[k] == wt
        blk1 = engs[0].linearize(poses, 1, 1); blk0 = engs[1].linearize(poses, 1, 1)
        assert np.array_equal(blk1, blk0), r   # reused lists / operand streams are the very same bytes
        poses, sm = engs[0].optimize(poses, pb["fixed"], L.PARAM_SOPHUS_SE3, 1, True, 50)
    # the cache must actually engage once the poses settle (fraction of queries still searched collapses)
    assert hits[0] == 1.0 and hits[-1] < 0.2, hits
    # a changed cutoff or a perturbed pose must fall back to searching and stay exact
    poses2 = poses.copy(); poses2[2][:3, 3] += [0.004, -0.003, 0.002]
    for e in engs:
        e.correspond(poses2, pb["fixed"], 0.05, L.NN_GRID)
    for k in range(engs[0].E):
        assert all(np.array_equal(x, y) for x, y in zip(engs[0].get_correspondences(k), engs[1].get_correspondences(k)))
    for e in engs:
        e.correspond(poses2, pb["fixed"], 0.01, L.NN_GRID)
    for k in range(engs[0].E):
        assert all(np.array_equal(x, y) for x, y in zip(engs[0].get_correspondences(k), engs[1].get_correspondences(k)))
    for e in engs:
        e.close()


@pytest.mark.parametrize("mu", [0.05, 0.4, 0.003])
def test_tile_kernel_lower_bounds_feed_the_temporal_cache(orc, mu):
    """The tile kernel's BND build (the round in which AUTO hands over to the grid kernel) leaves, per query, a lower bound on the
    distance to every target other than the answer; the next grid search trusts it (temporal cache).  Engine A: tile search WITH
    bounds at poses P, then a grid search at P moved by a random rigid motion of a given size; engine B does the same with the cache
    off (full searches).  For motions from 1e-7 m (nearly every query is a cache hit) to millimetres (hardly any) the lists, counts,
    weights and normal-equation blocks must be identical, the first and last also to the oracle's; and the cache must really engage
    for the small motions — otherwise the test proves nothing."""
    pb = synth.make_problem(4, 6000)
    engs = []
    for cache in (1, 0):
        e = mvicp.Engine(0)
        e.set_option("nn_cache", cache); e.set_option("list_reuse", cache)
        e.set_option("tile_bounds", 2); e.set_option("tile_mu", mu)
        e.set_frames(pb["pts"], pb["nor"]); e.set_graph(pb["src"], pb["dst"])
        e.profile(True); e.set_option("nn_census", 1)
        engs.append(e)
    rng = np.random.default_rng(5)
    poses = pb["init"].copy()
    for r in range(3):   # get near the fixed point first: the hand-over happens in a nearly converged registration
        for e in engs:
            e.correspond(poses, pb["fixed"], 0.05, L.NN_TILE)
        poses, _ = engs[0].optimize(poses, pb["fixed"], L.PARAM_SOPHUS_SE3, 1, True, 50)
    mags = [1e-7, 1e-6, 1e-5, 1e-4, 5e-4, 3e-3, 0.0]
    hit_frac = []
    for it, mag in enumerate(mags):
        for e in engs:   # tile search (A: leaves bounds) at the current poses
            e.correspond(poses, pb["fixed"], 0.05, L.NN_TILE)
        moved = poses.copy()
        for k in range(1, len(moved)):
            T = np.eye(4); T[:3, :3] = synth.so3_exp(rng.normal(0, mag / 0.4, 3)); T[:3, 3] = rng.normal(0, mag, 3)
            moved[k] = moved[k] @ T
        res = []
        for e in engs:
            e.profile_reset()
            c, w = e.correspond(moved, pb["fixed"], 0.05, L.NN_GRID)
            res.append((c, w, [e.get_correspondences(k) for k in range(e.E)], e.nn_census()))
        (c1, w1, l1, s1), (c0, w0, l0, s0) = res
        assert np.array_equal(c1, c0) and w1.tobytes() == w0.tobytes(), (mu, mag)
        for a, b in zip(l1, l0):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (mu, mag)
        assert np.array_equal(engs[0].linearize(moved, 1, 1), engs[1].linearize(moved, 1, 1)), (mu, mag)
        hit_frac.append(s1["hits"] / s1["queries"])
        assert s0["hits"] == 0
        if it in (0, len(mags) - 2):
            for k, (s_, d_) in enumerate(zip(pb["src"], pb["dst"])):
                f, sec, dist, wt, _, _ = orc.correspond_edge(pb["pts"][s_], moved[s_], pb["pts"][d_], moved[d_], 0.05)
                assert np.array_equal(l1[k][0], f) and np.array_equal(l1[k][1], sec) and np.array_equal(l1[k][2], dist) and w1[k] == wt
        # a second cached grid round on top (bounds now partly the tile kernel's, partly the grid kernel's own)
        for e in engs:
            e.correspond(poses, pb["fixed"], 0.05, L.NN_GRID)
        for k in range(engs[0].E):
            assert all(np.array_equal(x, y) for x, y in zip(engs[0].get_correspondences(k), engs[1].get_correspondences(k))), (mu, mag)
    # the bounds are useful: tiny motions are answered from the cache, millimetre motions are not (0.0 = identical poses: all hits
    # except exact ties / queries without a neighbour)
    assert hit_frac[0] > 0.5 and hit_frac[-1] > 0.5 and hit_frac[-2] < hit_frac[0], hit_frac
    for e in engs:
        e.close()


@pytest.mark.parametrize("opts", [
    {"tile_seed": 0}, {"tile_waves": 4}, {"tile_waves": 8}, {"prune_rho": 0.0}, {"prune_rho": 0.6},
    {"auto_settle": 0.05}, {"auto_settle": 5.0}, {"nn_cache": 0, "list_reuse": 0}, {"spin_wait": 1}, {"sel_bracket": 0},
    {"grid_curve": 0}, {"grid_curve": 1}, {"grid_target": 2.5}, {"nn_cell": 1}, {"nn_cell": 1, "auto_switch": 0.2}, {"nn_cell": 1, "auto_switch": 50.0}, {"nn_cell": 1, "prune_rho": 0.0},
    {"prune_rho": 3.0}, {"spec_eval": 0}, {"lin_share_p": 0}, {"tile_bounds": 0}, {"tile_bounds": 2}, {"tile_bounds": 2, "tile_mu": 0.5}, {"tile_mu": 0.005},
    {"tile_cache": 0}, {"tile_cache": 0, "list_reuse": 0}, {"tile_cache": 1, "list_reuse": 0}, {"tile_cache": 1, "tile_mu": 0.5}, {"tile_cache": 1, "auto_settle": 5.0},
    # the matrix-pipe tile kernel (nn_mfma.hip) against the VALU one (nn_tile.hip), in every role, and its own tunables
    {"tile_mfma": 0}, {"tile_mfma": 2}, {"tile_mfma": 2, "tile_bounds": 2}, {"tile_mfma": 0, "tile_bounds": 2}, {"tile_mfma": 2, "tile_cache": 1, "tile_mu": 0.5},
    {"mfma_trig": 0}, {"mfma_trig": 33}, {"mfma_kacc": 4}, {"mfma_kacc": 512}, {"tile_mfma": 2, "tile_seed": 0}, {"mfma_lbt": 0}, {"mfma_lbt": 0, "tile_seed": 0}, {"tile_mfma": 2, "tile_waves": 6}, {"tile_mfma": 2, "tile_waves": 4},
    # round 6: the miss_block path of the cache-aware rounds (off / every wave), the early cache prologue, the seed-block entry of the seeded launches
    {"tile_miss": 0}, {"tile_miss": 64}, {"tile_miss": 64, "tile_mu": 0.5}, {"tile_miss": 2, "auto_settle": 5.0}, {"tile_bounds": 2, "tile_cache": 2}, {"tile_bounds": 2, "tile_cache": 2, "tile_mfma": 2, "tile_miss": 64},
    {"mfma_entry": 1}, {"mfma_entry": 1, "tile_bounds": 2}, {"mfma_entry": 1, "tile_seed": 0},
    {"spec2_eval": 0}, {"lin_interleave": 0}, {"lin_interleave": 0, "lin_share_p": 0}, {"reject_cache": 0}, {"reject_cache": 0, "tile_mfma": 2}, {"tile_mfma": 2, "auto_settle": 5.0}, {"cache_mfma_ratio": 0}, {"cache_mfma_ratio": 0.01}, {"cache_mfma_ratio": 0.01, "auto_settle": 5.0},
])
def test_tuning_options_never_change_results(opts):
    """Every speed knob (kernel variants, seeding, cell pruning, AUTO hand-over policy, caches) must leave the whole ICP
    trajectory bit-identical: counts, weights, correspondence lists every round, and the final poses.  The two knobs that
    change the sorted order of the clouds (k-d / curve order, cell size) change the summation order of the normal equations, so for
    them the correspondences are bit-identical at equal poses (round 0) and the trajectory agrees to rounding."""
    reorders = "grid_curve" in opts or "grid_target" in opts
    base_opts = {}
    if "nn_cell" in opts:   # the cell-staging kernel needs the brick map, which exists for the cell-curve orders only: same order on both sides
        base_opts = {"grid_curve": 1}
        opts = dict(opts, grid_curve=1)
    pb = synth.make_problem(5, 5000)

    def run(options):
        e = mvicp.Engine(0)
        for k, v in options.items():
            e.set_option(k, v)
        e.set_frames(pb["pts"], pb["nor"]); e.set_graph(pb["src"], pb["dst"])
        poses = pb["init"].copy()
        trace = []
        for r in range(7):
            c, w = e.correspond(poses, pb["fixed"], 0.05, L.NN_AUTO)
            lists = [e.get_correspondences(k) for k in range(e.E)] if r in (0, 3, 6) else None
            trace.append((c.copy(), w.copy(), lists))
            poses, sm = e.optimize(poses, pb["fixed"], L.PARAM_SOPHUS_SE3, 1, True, 50)
        e.close()
        return trace, poses

    base_trace, base_poses = run(base_opts)
    trace, poses = run(opts)
    for r, ((c0, w0, l0), (c1, w1, l1)) in enumerate(zip(base_trace, trace)):
        if reorders and r > 0:
            assert np.array_equal(c0, c1) and np.allclose(w0, w1, rtol=1e-6, atol=0), (opts, r)
            continue
        assert np.array_equal(c0, c1) and w0.tobytes() == w1.tobytes(), (opts, r)
        if l0 is not None:
            for a, b in zip(l0, l1):
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (opts, r)
    if reorders:
        assert np.allclose(base_poses, poses, rtol=0, atol=1e-12), opts
    else:
        assert np.array_equal(base_poses, poses), opts


# ---------------------------------------------------------------- error behaviour of the ABI
def test_abi_error_paths(eng):
    pb = synth.make_problem(2, 500)
    e = mvicp.Engine(0)
    with pytest.raises(mvicp.MvicpError):          # graph before frames
        e.set_graph([1], [0])
    e.set_frames(pb["pts"], None)                  # no normals
    with pytest.raises(mvicp.MvicpError):          # self edge / out of range
        e.set_graph([1], [1])
    with pytest.raises(mvicp.MvicpError):
        e.set_graph([2], [0])
    e.set_graph([1], [0])
    with pytest.raises(mvicp.MvicpError):          # linearize before any correspondences
        e.linearize(pb["init"], 0, 0)
    e.correspond(pb["init"], pb["fixed"], 0.05)
    with pytest.raises(mvicp.MvicpError):          # point-to-plane without normals on the dst frame
        e.linearize(pb["init"], 1, 1)
    blk = e.linearize(pb["init"], 0, 1)            # point-to-point is fine
    assert np.isfinite(blk).all()
    with pytest.raises(mvicp.MvicpError):          # bad explicit correspondences
        e.set_correspondences(0, [0, 1], [0, 10_000], 0.0)
    with pytest.raises(mvicp.MvicpError):          # set_frame after the graph is frozen
        e.set_frames(pb["pts"], None) or e.lib.mvicp_set_frame(e.h, 0, None, None, 5) and (_ for _ in ()).throw(mvicp.MvicpError("x"))
    e.close()


def test_empty_edges_and_tiny_cutoff(eng, orc):
    """An edge with no correspondence inside the cutoff: count 0, weight 0 (the reference dereferences end() there,
    frame.cpp:166-168), and the LM simply leaves the poses where they are."""
    pb = synth.make_problem(3, 800)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    counts, weights = eng.correspond(pb["init"], pb["fixed"], 1e-9)
    assert counts.sum() == 0 and np.all(weights == 0)
    P, sm = eng.optimize(pb["init"], pb["fixed"])
    # (poses round-trip through the parameterization like the reference's, icp-ceres.cpp:405-420,472-474: equal to rounding)
    assert np.allclose(P, pb["init"], rtol=0, atol=1e-15) and sm["final_cost"] == 0.0 and sm["iterations"] == 0


def _weight_ref(d2, thresh):
    """frame.cpp:156-176 on the host: keep sqrt(d2) < (double)thresh, upper median, x1.5, to float."""
    d = np.sqrt(d2)
    d = d[d < float(np.float32(thresh))]
    if d.size == 0:
        return 0, np.float32(0)
    return d.size, np.float32(np.partition(d, d.size // 2)[d.size // 2] * 1.5)


@pytest.mark.parametrize("case", ["all_zero", "half_zero", "near_equal", "one_binade", "wide", "single"])
def test_median_select_degenerate_key_sets(eng, case):
    """The radix select (exponent digit -> two compacted refinements -> in-LDS finish) on key sets that stress each stage:
    all keys equal (compact buffers as long as the list), zeros mixed with normal keys, keys that differ only in their lowest
    mantissa bits, keys spanning many binades, a single key.  Weight and count must equal the reference rule bit for bit."""
    rng = np.random.default_rng(11)
    n = 9001 if case != "single" else 1
    dst = rng.uniform(-1.0, 1.0, (n, 3))
    if case == "all_zero":
        off = np.zeros((n, 3))
    elif case == "half_zero":
        off = np.where(rng.random((n, 1)) < 0.5, 0.0, 1.0) * rng.normal(0, 1e-5, (n, 3))
    elif case == "near_equal":
        off = np.tile([3e-5, 0.0, 0.0], (n, 1))           # d2 ~ 9e-10 for every pair, low bits differ with the rounding of x + 3e-5
    elif case == "one_binade":
        off = rng.uniform(1.0, 1.4, (n, 1)) * np.array([[2e-5, 0.0, 0.0]])
    elif case == "wide":
        off = (10.0 ** rng.uniform(-12, -3.5, (n, 1))) * np.array([[1.0, 0.0, 0.0]])
    else:
        off = np.array([[1e-4, 0.0, 0.0]])
    src = dst + off
    I = np.eye(4)
    poses = np.stack([I, I])
    fixed = np.array([1, 0], dtype=np.uint8)
    eng.set_frames([dst, src], [None, None]); eng.set_graph([1], [0])
    idx, d2 = eng.nn_query(0, src, L.NN_BRUTE)
    want_n, want_w = _weight_ref(d2, 0.05)
    for method in (L.NN_GRID, L.NN_TILE):
        counts, weights = eng.correspond(poses, fixed, 0.05, method)
        assert counts[0] == want_n
        assert weights.dtype == np.float32 and weights[0].tobytes() == want_w.tobytes(), (case, method, weights[0], want_w)


def test_full_size_properties_cfg4():
    """BASELINE config 4 size (32 views x 200k points, 62 edges): size-independent properties of the whole round —
    (1) a second search at the same poses (temporal cache + list reuse engaged) returns exactly the first result;
    (2) rank-0-of-2 + rank-1-of-2 shards sum bit-exactly to the unsharded per-edge blocks and counts;
    (3) moving every pose by one common rigid transform changes neither the correspondences nor the blocks beyond rounding;
    (4) lists are strictly ascending in the source index and the counts checksum matches."""
    pb = synth.make_problem(32, 200_000)
    full = mvicp.Engine(0)
    full.set_frames(pb["pts"], pb["nor"]); full.set_graph(pb["src"], pb["dst"])
    c1, w1 = full.correspond(pb["gt"], pb["fixed"], 0.05)
    b1 = full.linearize(pb["gt"], 1, 1)
    l1 = [full.get_correspondences(e) for e in (0, 17, 61)]
    c2, w2 = full.correspond(pb["gt"], pb["fixed"], 0.05)          # cache hits everywhere, lists reused
    b2 = full.linearize(pb["gt"], 1, 1)
    l2 = [full.get_correspondences(e) for e in (0, 17, 61)]
    assert np.array_equal(c1, c2) and np.array_equal(w1, w2) and np.array_equal(b1, b2)
    for a, b in zip(l1, l2):
        assert all(np.array_equal(x, y) for x, y in zip(a, b)) and np.all(np.diff(a[0]) > 0)
    assert int(c1.sum()) == sum(len(pb["pts"][s]) for s in pb["src"])  # cutoff 5 cm: every query finds a partner on this scene
    T = np.eye(4); T[:3, :3] = synth.so3_exp(np.array([-0.4, 0.1, 0.25])); T[:3, 3] = [0.05, -0.2, 0.3]
    moved = np.array([T @ P for P in pb["gt"]])
    c3, w3 = full.correspond(moved, pb["fixed"], 0.05)
    same = [np.array_equal(full.get_correspondences(e)[1], l1[k][1]) for k, e in enumerate((0, 17, 61))]
    assert np.array_equal(c3, c1) and all(same)  # (the query map changes by rounding only: neighbours are identical here)
    b3 = full.linearize(moved, 1, 1)
    assert np.allclose(b3, b1, rtol=1e-8, atol=1e-12 * np.abs(b1[:, :78]).max())
    full.close()
    parts, counts = [], []
    for r in range(2):
        e = mvicp.Engine(0, rank=r, world=2)
        e.set_frames(pb["pts"], pb["nor"]); e.set_graph(pb["src"], pb["dst"])
        c, w = e.correspond(pb["gt"], pb["fixed"], 0.05)
        parts.append(e.linearize(pb["gt"], 1, 1)); counts.append(c)
        e.close()
    assert np.array_equal(parts[0] + parts[1], b1) and np.array_equal(counts[0] + counts[1], c1)


def test_speculative_first_evaluation_is_used_and_exact():
    """mvicp_correspond queues the first linearization of the following solve (same poses after the parameterization round trip,
    previous solve's flags).  From the second round on every solve must be served by it (one wait per round instead of two), a change
    of flags must fall back to a fresh evaluation, and the trajectory must be bit-identical to spec_eval = 0."""
    pb = synth.make_problem(4, 4000)
    res = {}
    for spec in (1, 0):
        e = mvicp.Engine(0)
        e.set_option("spec_eval", spec)
        e.set_frames(pb["pts"], pb["nor"]); e.set_graph(pb["src"], pb["dst"])
        e.profile(True)
        poses = pb["init"].copy()
        hits = []
        for r in range(6):
            e.profile_reset()
            e.correspond(poses, pb["fixed"], 0.05)
            param, plane = (L.PARAM_ANGLE_AXIS, 0) if r == 3 else (L.PARAM_SOPHUS_SE3, 1)   # round 3 changes the flags: the queued launch must be ignored
            poses, sm = e.optimize(poses, pb["fixed"], param, plane, True, 50)
            hits.append(e.profile_get("spec.hit")[1])
        res[spec] = (poses, hits)
        e.close()
    assert np.array_equal(res[1][0], res[0][0])
    assert res[0][1] == [0] * 6
    assert res[1][1] == [0, 1, 1, 0, 0, 1], res[1][1]   # round 0: no flags yet; round 3: flags differ; round 4: flags differ from round 3's


def test_second_queued_evaluation_at_the_fixed_point_is_used_and_exact():
    """Round 6 (spec2_eval): once a registration has converged, a round's poses are last round's bit for bit and its solve is `first evaluation -> one LM iteration ->
    candidate evaluation -> function-tolerance stop`, with last round's candidate poses.  mvicp_correspond then queues BOTH evaluations behind its own kernels, so the
    round waits once instead of twice.  From the second fixed-point round on every solve's two evaluations are served by the queued launches; a round whose poses moved
    gets none; a change of flags voids both; the whole trajectory (poses, counts, weights, LM summaries) is bit-identical to spec2_eval = 0."""
    pb = synth.make_problem(4, 4000)
    res = {}
    for spec2 in (1, 0):
        e = mvicp.Engine(0)
        e.set_option("spec2_eval", spec2)
        e.set_frames(pb["pts"], pb["nor"]); e.set_graph(pb["src"], pb["dst"])
        e.profile(True)
        poses = pb["init"].copy()
        log = []
        for r in range(16):
            e.profile_reset()
            before = poses.copy()
            c, w = e.correspond(poses, pb["fixed"], 0.05)
            param, plane = (L.PARAM_ANGLE_AXIS, 0) if r == 14 else (L.PARAM_SOPHUS_SE3, 1)   # round 14 changes the flags at the fixed point
            poses, sm = e.optimize(poses, pb["fixed"], param, plane, True, 50)
            log.append((c.copy(), w.tobytes(), poses.copy(), sm["iterations"], sm["evaluations"], sm["termination"], e.profile_get("spec.hit")[1], e.profile_get("spec2.hit")[1],
                        bool(np.array_equal(before, poses))))
        res[spec2] = log
        e.close()
    for a, b in zip(res[1], res[0]):
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2]) and a[3:6] == b[3:6]
    assert all(l[7] == 0 for l in res[0])
    still = [l[8] for l in res[1]]                       # rounds whose solve left the poses untouched
    assert sum(still) >= 6 and not still[0], still
    h2 = [l[7] for l in res[1]]
    for r in range(2, 14):
        # two consecutive untouched rounds before r: the search of round r saw last search's poses, and last solve ended on a candidate evaluation
        if still[r - 2] and still[r - 1] and still[r]:
            assert h2[r] == 1 and res[1][r][4] == 2, (r, h2, still)      # both evaluations of the solve were queued ones
        if not still[r - 1]:
            assert h2[r] == 0, (r, h2, still)                            # the poses moved since the last search: nothing queued
    assert sum(h2[:14]) >= 4, (h2, still)
    assert h2[14] == 0 and res[1][14][6] == 0                            # flags changed: both queued evaluations are ignored


def test_changed_fixed_mask_at_identical_poses_is_not_mistaken_for_a_fixed_point():
    """The converged-registration shortcuts (bit-identical transforms: nothing rewritten, compaction not launched) must not survive
    a change of the fixed mask at the very same poses: edges out of a newly fixed frame lose their lists (frame.cpp:93), edges out of a
    released frame get theirs."""
    pb = synth.make_problem(4, 3000)
    src, dst = synth.pose_graph_knn(pb["init"], 2, skip_fixed0=False)
    eng = mvicp.Engine(0)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(src, dst)
    masks = [np.array(m, dtype=np.uint8) for m in ([1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 1, 0], [1, 0, 1, 0], [0, 0, 0, 0], [1, 0, 0, 0])]
    for m in masks:
        for _ in range(2):   # twice: the second call runs with bit-identical transforms AND an unchanged mask
            c, w = eng.correspond(pb["gt"], m, 0.05, L.NN_GRID)
            blk = eng.linearize(pb["gt"], 1, 1)
            fresh = mvicp.Engine(0)
            fresh.set_frames(pb["pts"], pb["nor"]); fresh.set_graph(src, dst)
            cf, wf = fresh.correspond(pb["gt"], m, 0.05, L.NN_GRID)
            bf = fresh.linearize(pb["gt"], 1, 1)
            fresh.close()
            assert np.array_equal(c, cf) and w.tobytes() == wf.tobytes() and np.array_equal(blk, bf), m
            assert all(c[e] == 0 for e in range(len(src)) if m[src[e]])
    eng.close()


# ---------------------------------------------------------------- randomised shapes (round 3)
def _random_cloud(rng, n, kind, scale):
    if kind == "blob":
        p = rng.normal(0.0, 0.05, (n, 3))
    elif kind == "line":
        p = np.outer(rng.uniform(-1, 1, n), [0.3, -0.2, 0.1]) + rng.normal(0.0, 1e-6, (n, 3))
    elif kind == "plane":
        p = np.column_stack([rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n), np.zeros(n)])
    elif kind == "lattice":   # exact ties between targets everywhere (a range-image grid, like the Bunny scans)
        side = int(np.ceil(n ** 0.5))
        g = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2)[:n].astype(np.float64)
        p = np.column_stack([g * 0.0078125, np.full(n, 0.25)])
    else:                      # two well-separated clusters (queries far from most of the cloud)
        p = np.vstack([rng.normal(-0.5, 0.01, (n // 2, 3)), rng.normal(0.5, 0.01, (n - n // 2, 3))])
    return np.ascontiguousarray(p * scale)


@pytest.mark.parametrize("seed", range(40))
def test_random_shapes_every_kernel_and_the_cached_rounds(eng, orc, seed):
    """Irregular inputs the synthetic surfaces never produce: 1..900 points per cloud, degenerate shapes, exact ties, coordinates scaled
    over five decades.  (1) every NN kernel against the oracle's brute force; (2) three AUTO rounds with slightly moving poses, so that the
    seeded tile rounds, the hand-over, the temporal cache and the in-place lists all run on them, each round against the oracle."""
    rng = np.random.default_rng(100 + seed)
    kinds = ["blob", "line", "plane", "lattice", "clusters"]
    scale = float(10.0 ** rng.integers(-3, 3))
    K = 3
    top = 6000 if seed % 4 == 3 else 900
    pts = [_random_cloud(rng, int(rng.integers(1, top)), kinds[int(rng.integers(0, 5))], scale) for _ in range(K)]
    # (1) single queries
    eng.set_frames(pts, None)
    q = np.vstack([pts[1][: min(len(pts[1]), 64)], _random_cloud(rng, 97, "blob", scale)])
    for m in METHODS:
        idx, d2 = eng.nn_query(0, q, m)
        oi, od = orc.nn_brute(pts[0], q)
        assert np.array_equal(idx, oi) and np.array_equal(d2, od), (seed, m)
    # (2) rounds
    src = np.array([1, 2, 2], dtype=np.int32); dst = np.array([0, 0, 1], dtype=np.int32)
    fixed = np.array([1, 0, 0], dtype=np.int32)
    eng2 = mvicp.Engine(0)
    try:
        eng2.set_frames(pts, None); eng2.set_graph(src, dst)
        thresh = np.float32(0.3 * scale)
        poses = np.stack([np.eye(4)] * K)
        for r in range(4):
            for k in range(1, K):   # a small rigid motion per round, shrinking: the later rounds are cache hits
                a = rng.normal(0.0, 0.02 / 4 ** r)
                Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
                poses[k, :3, :3] = Rz @ poses[k, :3, :3]
                poses[k, :3, 3] += rng.normal(0.0, 0.003 * scale / 4 ** r, 3)
            counts, weights = eng2.correspond(poses, fixed, thresh, L.NN_AUTO)
            for e, (s, d) in enumerate(zip(src, dst)):
                f, sec, dist, w, _, _ = orc.correspond_edge(pts[s], poses[s], pts[d], poses[d], thresh)
                gf, gs, gd = eng2.get_correspondences(e)
                assert counts[e] == len(f), (seed, r, e)
                assert np.array_equal(gf, f) and np.array_equal(gs, sec) and np.array_equal(gd, dist), (seed, r, e)
                assert weights[e] == (w if len(f) else 0), (seed, r, e)
        counts2, weights2 = eng2.correspond(poses, fixed, thresh, L.NN_AUTO)   # same poses again
        assert np.array_equal(counts2, counts) and weights2.tobytes() == weights.tobytes()
    finally:
        eng2.close()


@pytest.mark.parametrize("seed", range(30))
def test_random_graphs_costs_and_parameterizations(eng, orc, seed):
    """Random pose graphs (2..6 views, arbitrary directed edges incl. several into one target and edges out of fixed frames), random fixed
    masks (view 0 always fixed, as in the reference), every parameterization / cost / loss combination: per-edge blocks and the whole solve against the oracle."""
    rng = np.random.default_rng(500 + seed)
    K = int(rng.integers(2, 7))
    pb = synth.make_problem(K, int(rng.integers(600, 2500)), pose_seed=int(7000 + seed))
    pairs = [(s, d) for s in range(K) for d in range(K) if s != d]
    pick = rng.choice(len(pairs), size=int(rng.integers(1, min(len(pairs), 8) + 1)), replace=False)
    src = np.array([pairs[i][0] for i in sorted(pick)], dtype=np.int32)
    dst = np.array([pairs[i][1] for i in sorted(pick)], dtype=np.int32)
    fixed = (rng.uniform(size=K) < 0.3).astype(np.int32)
    fixed[0] = 1                                 # the reference always fixes view 0 (icp-ceres.cpp:244,341,417; mvicp_optimize forces it)
    param = [L.PARAM_SOPHUS_SE3, L.PARAM_ANGLE_AXIS, L.PARAM_EIGEN_QUATERNION][seed % 3]
    plane, robust = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    eng.set_frames(pb["pts"], pb["nor"])
    eng.set_graph(src, dst)
    counts, weights = eng.correspond(pb["init"], fixed, 0.05)
    corr = [eng.get_correspondences(e)[:2] for e in range(eng.E)]
    got = eng.linearize(pb["init"], plane, robust)
    want = orc.edge_blocks(pb["pts"], pb["nor"], src, dst, corr, weights, pb["init"], plane, robust)
    for e in range(eng.E):
        if counts[e] == 0:
            assert np.all(got[e] == 0)
            continue
        scale = np.abs(want[e, :78]).max()
        assert np.allclose(got[e, :78], want[e, :78], rtol=0, atol=1e-11 * scale), (seed, e)
        assert np.allclose(got[e, 78:90], want[e, 78:90], rtol=0, atol=1e-10 * (np.abs(want[e, 78:90]).max() + 1e-300)), (seed, e)
        assert abs(got[e, 90] - want[e, 90]) <= 1e-12 * abs(want[e, 90]) + 1e-300, (seed, e)
    P, sm = eng.optimize(pb["init"], fixed, param, plane, bool(robust), 50)
    prob = orc.make_problem(pb["pts"], pb["nor"], fixed, src, dst, corr, weights, param, plane, robust)
    P_ref, sm_ref = orc.optimize(prob, pb["init"], 50)
    assert sm["iterations"] == sm_ref["iterations"] and sm["termination"] == sm_ref["termination"], (seed, sm, sm_ref)
    for k in range(K):
        dt, dr = synth.pose_diff(P[k], P_ref[k])
        assert dt < 1e-8 and dr < 1e-8, (seed, k, dt, dr)
        if fixed[k]:
            assert synth.pose_diff(P[k], pb["init"][k])[0] < 1e-12


@pytest.mark.parametrize("seed", range(30))
def test_random_shapes_knn_lists_equal_nanoflann(eng, refnn, seed):
    """Row f1 beyond the Bunny lattice: k-NN lists of random clouds — blobs, jittered and exact lattices (ties at every place), planes,
    clusters, duplicated points; 12..3000 points, k = 3 / 10 / 16 — element for element against the real nanoflann's knnSearch
    (tie order = the order its tree visits the leaves, csrc/kdvisit.h)."""
    if refnn is None:
        pytest.skip("oracle/_ref was not built")
    rng = np.random.default_rng(900 + seed)
    kind = ["blob", "lattice", "plane", "clusters", "lattice"][seed % 5]
    n = int(rng.integers(12, 3000))
    pts = _random_cloud(rng, n, kind, float(10.0 ** rng.integers(-2, 2)))
    if seed % 4 == 1:
        pts[n // 2:] = pts[: n - n // 2]      # exact duplicates
    if seed % 5 == 1:
        pts[:, 2] += np.round(rng.normal(0, 2, n)) * 0.0078125 * 0.5   # a quantised third coordinate, like a range image
    k = [3, 10, 16][seed % 3]
    eng.set_frames([pts], None)
    nrm, knn = eng.recompute_normals(0, k, want_knn=True)
    gi, gd = refnn.knn_self(pts, k)
    e = pts[:, None, :] - pts[knn]
    myd = (e[:, :, 0] * e[:, :, 0] + e[:, :, 1] * e[:, :, 1]) + e[:, :, 2] * e[:, :, 2]
    assert np.array_equal(myd, gd), seed
    assert np.array_equal(knn, gi), (seed, int((~np.all(knn == gi, axis=1)).sum()))


@pytest.mark.parametrize("factor", [0.0, 1.0, 4.0])
@pytest.mark.parametrize("mfma", [0, 2])
def test_far_queries_and_unbounded_search(orc, factor, mfma):
    """Queries far outside the target, in units of the target's own blocks (the matrix-pipe kernel's f16 operands cover 4096 scaled units
    around a block: beyond that, and while no finite threshold exists, it must fall back to confirming exhaustively), with the search
    radius unbounded / equal to the cutoff / the default: the kept correspondences are the oracle's in every case, round after round."""
    rng = np.random.default_rng(77)
    small = rng.uniform(-1e-3, 1e-3, (3000, 3))                       # a 2 mm target ...
    far = rng.uniform(-1e-3, 1e-3, (2500, 3)) + np.array([0.9, -0.4, 0.2])   # ... asked from a metre away (|alpha| ~ 1e5 scaled units)
    mixed = np.vstack([rng.uniform(-0.2, 0.2, (1500, 3)), small[:700] + 1e-5])   # some near, some far
    # (the near ones are IN RANGE of the 2 mm block with a threshold of metres: -T is beyond the f16 pieces and the lane must admit everything)
    pts = [small, far, mixed]
    src = np.array([1, 2, 2], dtype=np.int32); dst = np.array([0, 0, 1], dtype=np.int32)
    fixed = np.array([1, 0, 0], dtype=np.int32)
    e = mvicp.Engine(0)
    try:
        e.set_option("tile_mfma", mfma); e.set_option("nn_search_factor", factor)
        e.set_frames(pts, None); e.set_graph(src, dst)
        poses = np.stack([np.eye(4)] * 3)
        for thresh in (np.float32(2.0), np.float32(0.05), np.float32(2.0)):
            for r in range(3):
                poses[1, :3, 3] += rng.normal(0, 1e-4, 3); poses[2, :3, 3] += rng.normal(0, 1e-4, 3)
                counts, weights = e.correspond(poses, fixed, thresh, L.NN_TILE)
                for k, (s, d) in enumerate(zip(src, dst)):
                    f, sec, dist, w, _, _ = orc.correspond_edge(pts[s], poses[s], pts[d], poses[d], thresh)
                    gf, gs, gd = e.get_correspondences(k)
                    assert counts[k] == len(f), (factor, mfma, float(thresh), r, k)
                    assert np.array_equal(gf, f) and np.array_equal(gs, sec) and np.array_equal(gd, dist), (factor, mfma, float(thresh), r, k)
    finally:
        e.close()


# ---------------------------------------------------------------- the lists in the reference's layout, all edges at once (export.hip)
@pytest.mark.parametrize("thresh", [0.05, 0.006])
def test_map_correspondences_is_the_reference_layout(eng, orc, thresh):
    """mvicp_map_correspondences = Frame::neighbours[j].correspondances of every edge (frame.cpp:129,156-160; frame.h:18-22): triples
    {first, second, dist} in ascending first — device un-sort + one copy — against the oracle's own loop, bit for bit (dist = the IEEE
    sqrt of d2, frame.cpp:139), over a registration: moving rounds (lists re-compacted), cache-aware rounds (entries patched in place)
    and fixed-point rounds (nothing rewritten); with a loose cutoff (every query kept) and a tight one (a third rejected, membership changes)."""
    pb = synth.make_problem(5, 6000, cone_deg=40.0 if thresh < 0.01 else 100.0, sigma=0.004 if thresh < 0.01 else 0.02, sigmat=0.002 if thresh < 0.01 else 0.01)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    eng.reset_history()
    poses = pb["init"].copy()
    for rnd in range(9):
        counts, weights = eng.correspond(poses, pb["fixed"], thresh)
        trip, off = eng.map_correspondences()
        assert off[0] == 0 and np.array_equal(np.diff(off), counts)
        if rnd in (0, 1, 4, 8):
            for e, (s, d) in enumerate(zip(pb["src"], pb["dst"])):
                f, sec, dist, w, _, _ = orc.correspond_edge(pb["pts"][s], poses[s], pb["pts"][d], poses[d], thresh)
                t = trip[off[e]:off[e + 1]]
                assert np.array_equal(t["first"], f) and np.array_equal(t["second"], sec), (rnd, e)
                assert t["dist"].tobytes() == dist.tobytes(), (rnd, e)
                gf, gs, gd = eng.get_correspondences(e)     # the per-edge call slices the same export
                assert np.array_equal(gf, f) and np.array_equal(gs, sec) and gd.tobytes() == dist.tobytes()
            if thresh < 0.01 and rnd == 0:
                assert counts.sum() < 0.95 * sum(len(pb["pts"][s]) for s in pb["src"])   # the cutoff does reject queries here
        poses, sm = eng.optimize(poses, pb["fixed"], L.PARAM_SOPHUS_SE3, 1, True, 50)
    # a second map call without a search in between is the same buffer (no work), also as a zero-copy view
    v, off2 = eng.map_correspondences(copy=False)
    assert np.array_equal(off, off2) and np.array_equal(v, trip)


def test_rejected_queries_become_cache_hits_without_changing_any_list(orc):
    """Round 6, `reject_cache`: in a cache-aware round a query whose old neighbour AND every other target are provably beyond the cutoff after the pose update is a
    temporal-cache hit (it stays rejected, frame.cpp:156).  Partial-overlap problem with a tight cutoff (a third of the queries rejected): the hit fraction of the
    cache-aware rounds is higher with the option than without, and with BOTH settings every round's lists, counts and float weights are the oracle's bit for bit."""
    thresh = 0.006
    pb = synth.make_problem(5, 6000, cone_deg=40.0, sigma=0.004, sigmat=0.002)
    hits = {}
    for opt in (0, 1):
        e = mvicp.Engine(0)
        e.set_option("reject_cache", opt)
        e.set_option("auto_settle", 0.05)         # hand over to the cache-aware rounds early, while the poses still move
        e.set_frames(pb["pts"], pb["nor"]); e.set_graph(pb["src"], pb["dst"])
        e.profile(True); e.set_option("nn_census", 1)
        poses = pb["init"].copy()
        hits[opt] = []
        for rnd in range(8):
            e.profile_reset()
            counts, weights = e.correspond(poses, pb["fixed"], thresh)
            cs = e.nn_census()
            hits[opt].append(cs["hits"] / max(cs["queries"], 1.0))
            assert counts.sum() < 0.95 * sum(len(pb["pts"][s]) for s in pb["src"])
            for k, (s_, d_) in enumerate(zip(pb["src"], pb["dst"])):
                f, sec, dist, w, _, _ = orc.correspond_edge(pb["pts"][s_], poses[s_], pb["pts"][d_], poses[d_], thresh)
                gf, gs, gd = e.get_correspondences(k)
                assert np.array_equal(gf, f) and np.array_equal(gs, sec) and gd.tobytes() == dist.tobytes() and weights[k] == w, (opt, rnd, k)
            poses, sm = e.optimize(poses, pb["fixed"], L.PARAM_SOPHUS_SE3, 1, True, 50)
        e.close()
    moving = [r for r in range(8) if 0.0 < hits[0][r] < 0.999 or 0.0 < hits[1][r] < 0.999]
    assert moving, (hits,)                                                        # some rounds really were cache-aware with the poses still moving
    assert all(hits[1][r] >= hits[0][r] for r in range(8)) and any(hits[1][r] > hits[0][r] for r in moving), hits   # (small problem, fast convergence: the gain is a fraction of a per cent here, 20 points on cfg4_partial)


def test_async_map_and_per_edge_wait(eng, orc):
    """mvicp_map_correspondences_async + mvicp_wait_correspondences (round 6): the export travels in chunks (one per source frame); after wait(e) the bytes of edge e
    are the oracle's list, whatever order the edges are waited for in; a wait without an export and an edge out of range are errors, not garbage."""
    pb = synth.make_problem(5, 5000)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    eng.reset_history()
    with pytest.raises(mvicp.MvicpError):
        eng.wait_correspondences(0)                       # nothing searched, nothing exported
    poses = pb["init"].copy()
    for rnd in range(3):
        counts, weights = eng.correspond(poses, pb["fixed"], 0.05)
        trip, off = eng.map_correspondences_async()       # views of the pinned buffer: defined edge by edge
        assert np.array_equal(np.diff(off), counts)
        for e in list(range(eng.E))[::-1] if rnd % 2 else range(eng.E):
            eng.wait_correspondences(e)
            s_, d_ = pb["src"][e], pb["dst"][e]
            f, sec, dist, w, _, _ = orc.correspond_edge(pb["pts"][s_], poses[s_], pb["pts"][d_], poses[d_], 0.05)
            t = trip[off[e]:off[e + 1]]
            assert np.array_equal(t["first"], f) and np.array_equal(t["second"], sec) and t["dist"].tobytes() == dist.tobytes(), (rnd, e)
        with pytest.raises(mvicp.MvicpError):
            eng.wait_correspondences(eng.E)
        t2, off2 = eng.map_correspondences()              # the blocking form afterwards: the same buffer
        assert np.array_equal(off2, off) and np.array_equal(t2, trip)
        poses, _ = eng.optimize(poses, pb["fixed"], L.PARAM_SOPHUS_SE3, 1, True, 50)


def test_map_correspondences_skips_explicit_and_fixed_edges(eng, orc):
    """Edges whose source is fixed are never searched (frame.cpp:93) and an edge that holds an explicit list (mvicp_set_correspondences: any
    order, repeats) has no per-query positions: both have zero width in the map; mvicp_get_correspondences still returns the explicit list
    in ascending-first order; the next search puts the edge back."""
    pb = synth.make_problem(3, 3000)
    src, dst = synth.pose_graph_knn(pb["init"], 2, skip_fixed0=False)   # frame 0's own edges included
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(src, dst)
    counts, _ = eng.correspond(pb["init"], pb["fixed"], 0.05)
    trip, off = eng.map_correspondences()
    assert np.array_equal(np.diff(off), counts) and np.all(counts[src == 0] == 0) and np.all(counts[src != 0] > 0)
    e = int(np.flatnonzero(src != 0)[0])
    rng = np.random.default_rng(5)
    fi = rng.integers(0, len(pb["pts"][src[e]]), 500).astype(np.int32); se = rng.integers(0, len(pb["pts"][dst[e]]), 500).astype(np.int32)
    eng.set_correspondences(e, fi, se, 0.01)
    trip2, off2 = eng.map_correspondences()
    assert off2[e + 1] == off2[e]
    others = [k for k in range(len(src)) if k != e]
    for k in others:
        assert np.array_equal(trip2[off2[k]:off2[k + 1]], trip[off[k]:off[k + 1]])
    gf, gs, gd = eng.get_correspondences(e)
    order = np.argsort(fi, kind="stable")
    assert np.array_equal(gf, fi[order]) and np.array_equal(gs, se[order]) and np.all(gd == 0)
    counts3, _ = eng.correspond(pb["init"], pb["fixed"], 0.05)
    trip3, off3 = eng.map_correspondences()
    assert np.array_equal(counts3, counts) and np.array_equal(trip3, trip) and np.array_equal(off3, off)


def test_device_sqrt_is_the_ieee_sqrt(eng):
    """export.hip hands back dist = __dsqrt_rn(d2): must equal the host's correctly rounded sqrt (frame.cpp:139) for every d2 — checked through
    the product path on a cloud whose query distances cover twelve decades (explicit targets at random distances from the queries)."""
    rng = np.random.default_rng(11)
    n = 200000
    dst = rng.uniform(-1, 1, (n, 3))
    r = 10.0 ** rng.uniform(-9, 0, n)
    u = rng.normal(0, 1, (n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    srcp = dst + u * r[:, None] * 1e-3
    eng.set_frames([dst, srcp], None); eng.set_graph([1], [0])
    P = np.array([np.eye(4), np.eye(4)])
    counts, _ = eng.correspond(P, [1, 0], 0.05)
    trip, off = eng.map_correspondences()
    idx, d2 = eng.nn_query(0, srcp[trip["first"]])
    assert counts[0] > 0.99 * n and np.array_equal(idx, trip["second"])
    assert np.sqrt(d2).tobytes() == trip["dist"].tobytes()
    assert trip["dist"].min() < 1e-11 and trip["dist"].max() > 1e-4


# ---------------------------------------------------------------- regime transitions of mvicp_correspond
@pytest.mark.parametrize("seed", range(8))
def test_correspond_regime_transitions_match_a_fresh_context(orc, seed):
    """mvicp_correspond carries a dozen cross-round flags (temporal cache, seeds, reusable lists, settled medians, bracket select, AUTO state,
    tie / far skips, the queued evaluation).  Whatever happened before, a search is a pure function of (clouds, graph, poses, fixed mask,
    cutoff): after EVERY round of a randomly perturbed registration — cutoff changes, fixed-mask changes, mvicp_reset_history, forced kernel
    methods, option flips, rounds without a solve (bit-identical poses), an explicit list installed on an edge — the persistent context's
    counts, float weights and reference-order triples must equal those of a FRESH context asked once at the same poses (and the oracle's
    on a sampled edge)."""
    rng = np.random.default_rng(9000 + seed)
    K = int(rng.integers(3, 6))
    pb = synth.make_problem(K, int(rng.integers(2500, 5000)), cone_deg=float(rng.choice([100.0, 45.0])), pose_seed=int(700 + seed))
    src, dst = pb["src"], pb["dst"]
    A = mvicp.Engine(0)
    A.set_frames(pb["pts"], pb["nor"]); A.set_graph(src, dst)
    poses = pb["init"].copy()
    fixed = pb["fixed"].copy()
    cutoff = 0.05
    method = L.NN_AUTO
    events = []
    for rnd in range(14):
        ev = rng.choice(["none", "none", "cutoff", "fixed", "reset", "method", "option", "hold", "explicit"]) if rnd > 0 else "none"
        events.append(ev)
        if ev == "cutoff":
            cutoff = float(rng.choice([0.05, 0.02, 0.008, 0.004]))
        elif ev == "fixed":
            k = int(rng.integers(1, K)); fixed[k] = 1 - fixed[k]
        elif ev == "reset":
            A.reset_history()
        elif ev == "method":
            method = int(rng.choice([L.NN_AUTO, L.NN_AUTO, L.NN_BRUTE, L.NN_GRID, L.NN_TILE]))
        elif ev == "option":
            name = str(rng.choice(["list_reuse", "nn_cache", "sel_bracket", "spec_eval", "tile_cache", "tile_seed", "tile_miss", "mfma_entry", "reject_cache"]))
            A.set_option(name, float(rng.integers(0, 2)) * (8.0 if name == "tile_miss" else 1.0))
            if rng.random() < 0.5:
                A.set_option("tile_mfma", float(rng.integers(0, 3)))
        elif ev == "explicit":
            e = int(rng.integers(0, len(src)))
            n = int(rng.integers(1, 200))
            A.set_correspondences(e, rng.integers(0, len(pb["pts"][src[e]]), n).astype(np.int32), rng.integers(0, len(pb["pts"][dst[e]]), n).astype(np.int32), 0.01)
        counts, weights = A.correspond(poses, fixed, cutoff, method)
        trip, off = A.map_correspondences()
        B = mvicp.Engine(0)
        B.set_frames(pb["pts"], pb["nor"]); B.set_graph(src, dst)
        cb, wb = B.correspond(poses, fixed, cutoff, L.NN_BRUTE if rnd % 3 == 0 else L.NN_AUTO)
        tb, ob = B.map_correspondences()
        B.close()
        assert np.array_equal(counts, cb) and weights.tobytes() == wb.tobytes(), (seed, rnd, events)
        assert np.array_equal(off, ob) and np.array_equal(trip, tb), (seed, rnd, events)
        e = int(rng.integers(0, len(src)))
        if not fixed[src[e]]:
            f, sec, dist, w, _, _ = orc.correspond_edge(pb["pts"][src[e]], poses[src[e]], pb["pts"][dst[e]], poses[dst[e]], cutoff)
            t = trip[off[e]:off[e + 1]]
            assert np.array_equal(t["first"], f) and np.array_equal(t["second"], sec) and t["dist"].tobytes() == dist.tobytes() and weights[e] == w, (seed, rnd, e, events)
        if ev != "hold" and counts.sum() > 0:
            poses, sm = A.optimize(poses, fixed, int(rng.integers(0, 3)), int(rng.integers(0, 2)), bool(rng.integers(0, 2)), 50)
    A.close()
