"""Anchors the LM half of the oracle (oracle/oracle.cpp part (b): Jets, functors, local parameterizations, LM): finite differences,
cross-parameterization agreement, and the reference's only known-answer test (src/main_pairwise.cpp:44-61,117-133) — whose published
result for real Ceres (README.md:141-146) is reproduced to six digits (test_readme_known_answer_reproduced: the pin of this half)."""
import numpy as np
import pytest

import orclib
from mvicp import synth

PARAMS = [orclib.PARAM_QUAT, orclib.PARAM_ANGLEAXIS, orclib.PARAM_SOPHUS]


def small_problem(orc, param, plane, robust, K=3, N=300, seed=3):
    rng = np.random.default_rng(seed)
    pb = synth.make_problem(K, N)
    src, dst = pb["src"], pb["dst"]
    corr, w = [], []
    for s, d in zip(src, dst):
        n = 120
        corr.append((rng.integers(0, N, n).astype(np.int32), rng.integers(0, N, n).astype(np.int32)))
        w.append(0.01 + 0.01 * rng.random())
    prob = orc.make_problem(pb["pts"], pb["nor"], pb["fixed"], src, dst, corr, w, param, plane, robust)
    return pb, prob


@pytest.mark.parametrize("param", PARAMS)
@pytest.mark.parametrize("plane", [0, 1])
@pytest.mark.parametrize("robust", [0, 1])
def test_gradient_matches_finite_differences(orc, param, plane, robust):
    pb, prob = small_problem(orc, param, plane, robust)
    K = len(pb["pts"])
    A = orc.ambient(param)
    x = np.concatenate([orc.pose_to_param(param, P) for P in pb["init"]])
    cost, H, g = orc.evaluate_x(prob, x)
    free = [i for i in range(K) if not pb["fixed"][i]]
    h = 1e-6
    for bi, f in enumerate(free):
        for l in range(6):
            d = np.zeros(6); d[l] = h
            xp = x.copy(); xm = x.copy()
            xp[f * A:(f + 1) * A] = orc.local_plus(param, x[f * A:(f + 1) * A], d)
            xm[f * A:(f + 1) * A] = orc.local_plus(param, x[f * A:(f + 1) * A], -d)
            cp, _, _ = orc.evaluate_x(prob, xp, jac=False)
            cm, _, _ = orc.evaluate_x(prob, xm, jac=False)
            fd = (cp - cm) / (2 * h)
            assert abs(fd - g[bi * 6 + l]) <= 1e-6 * max(1.0, abs(fd)) + 1e-9, (param, plane, robust, bi, l, fd, g[bi * 6 + l])
    assert np.allclose(H, H.T, rtol=1e-12, atol=1e-18)


@pytest.mark.parametrize("plane", [0, 1])
def test_gauss_newton_hessian_matches_fd_of_residuals_nonrobust(orc, plane):
    # non-robust: H = J^T J exactly; check H v against finite differences of the gradient for small residual problems
    pb, prob = small_problem(orc, orclib.PARAM_SOPHUS, plane, 0)
    poses = pb["gt"].copy()  # at GT with random correspondences residuals are large; GN != full Hessian, so check J^T J PSD + symmetric only
    cost, H, g = orc.evaluate(prob, poses)
    ev = np.linalg.eigvalsh(H)
    assert ev.min() > -1e-9 * ev.max()


def random_pose(rng, rot=0.5, tra=0.1):
    T = np.eye(4)
    T[:3, :3] = synth.so3_exp(rng.normal(0, rot, 3))
    T[:3, 3] = rng.normal(0, tra, 3)
    return T


@pytest.mark.parametrize("plane", [0, 1])
def test_three_parameterizations_agree_multiview(orc, plane):
    # exact correspondences through known transforms -> every parameterization must land on the same poses
    rng = np.random.default_rng(11)
    K, N = 4, 400
    world = rng.normal(0, 0.1, (N, 3))
    wn = rng.normal(0, 1, (N, 3)); wn /= np.linalg.norm(wn, axis=1, keepdims=True)
    gt = [np.eye(4)] + [random_pose(rng) for _ in range(K - 1)]
    pts = [(world - T[:3, 3]) @ T[:3, :3] for T in gt]
    nor = [wn @ T[:3, :3] for T in gt]
    init = [gt[0]] + [T @ np.block([[synth.so3_exp(rng.normal(0, 0.05, 3)), rng.normal(0, 0.02, (3, 1))], [np.zeros((1, 3)), np.ones((1, 1))]]) for T in gt[1:]]
    src = np.array([1, 1, 2, 2, 3, 3]); dst = np.array([0, 2, 1, 3, 2, 0])
    corr = [(np.arange(N, dtype=np.int32), np.arange(N, dtype=np.int32)) for _ in src]
    fixed = np.array([1, 0, 0, 0], dtype=np.uint8)
    res = []
    for param in PARAMS:
        prob = orc.make_problem(pts, nor, fixed, src, dst, corr, [0.01] * len(src), param, plane, 0)
        P, sm = orc.optimize(prob, np.array(init), 50)
        assert sm["final_cost"] < 1e-12 * max(1.0, sm["initial_cost"]) + 1e-20, sm
        res.append(P)
        for k in range(K):
            dt, dr = orc.pose_diff(P[k], gt[k])
            assert dt < 1e-7 and dr < 1e-5, (param, k, dt, dr, sm)
    for P in res[1:]:
        assert np.allclose(P, res[0], atol=1e-7)


@pytest.mark.parametrize("plane", [0, 1])
def test_pairwise_known_answer(orc, plane):
    """main_pairwise.cpp:44-61: recover P = addNoise(Translation(.01,-.01,-.005) Rx(pi/4) Ry(1) Rz(-.2), .1, .1) from
    index-aligned pairs (src = cloud, dst = P * cloud); README.md:141-146 reports diff_tra ~ 6-8e-11 and
    diff_rot ~ 1.7e-6 deg for all three Ceres variants (accuracy floor of poseDiff's acos)."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "bunny_nn.npz"))
    pts, nrm = G["dst"], G["dst_nor"]  # cloudXYZ_0 rows (every 4th)
    c, s = np.cos, np.sin
    Rx = np.array([[1, 0, 0], [0, c(np.pi / 4), -s(np.pi / 4)], [0, s(np.pi / 4), c(np.pi / 4)]])
    Ry = np.array([[c(1), 0, s(1)], [0, 1, 0], [-s(1), 0, c(1)]])
    Rz = np.array([[c(-.2), -s(-.2), 0], [s(-.2), c(-.2), 0], [0, 0, 1]])
    Pclean = np.eye(4); Pclean[:3, :3] = Rx @ Ry @ Rz; Pclean[:3, 3] = [.01, -.01, -.005]
    P = synth.add_noise(Pclean, 0.1, 0.1, np.random.default_rng(5489))
    dstp = pts @ P[:3, :3].T + P[:3, 3]
    dstn = nrm @ P[:3, :3].T
    N = len(pts)
    corr = [(np.arange(N, dtype=np.int32), np.arange(N, dtype=np.int32))]
    for param in PARAMS:
        # frame 0 = dst (fixed, identity), frame 1 = src (free, starts at identity): icp-ceres.cpp:137-218,525-565
        prob = orc.make_problem([dstp, pts], [dstn, nrm], [1, 0], [1], [0], corr, [0.0], param, plane, 0)
        Pout, sm = orc.optimize(prob, np.array([np.eye(4), np.eye(4)]), 50)
        dt, dr = orc.pose_diff(P, Pout[1])
        assert dt < 1e-9 and dr < 1e-5, (param, dt, dr, sm)


# ---------------------------------------------------------------- the reference's known-answer test on its exact inputs
KAT = None


def kat():
    global KAT
    if KAT is None:
        import os
        KAT = np.load(os.path.join(os.path.dirname(__file__), "golden", "pairwise_kat.npz"))
    return KAT


def test_kat_inputs_match_the_readme_listing(orc):
    """README.md:105-114 prints the first rows of cloudXYZ_0.xyz as loaded by main_pairwise.cpp:34-39 (6 significant digits);
    the golden's P is what common.h:36-67 produces from a default-seeded std::mt19937 (regenerated here through compiled C++)."""
    K = kat()
    assert K["pts"].shape == (16264, 3)
    assert np.allclose(K["pts"][0], [-0.076899, -0.081785, 0.421], atol=5e-7) and np.allclose(K["nor"][0], [-0.226502, -0.628639, -0.743983], atol=5e-7)
    assert np.allclose(K["pts"][9], [-0.076168, -0.074065, 0.417], atol=5e-7) and np.allclose(K["nor"][9], [-0.437831, -0.405047, -0.802647], atol=5e-7)
    P = orc.add_noise(K["Pclean"], 0.1, 0.1, reset=True)
    assert np.array_equal(P, K["P"])
    assert np.allclose(P[:3, :3] @ P[:3, :3].T, np.eye(3), atol=1e-15)
    # std::mt19937 default seed 5489: the 10000th output is 4123659995 (C++ standard, [rand.predef]); the noise really is N(0,1)-sized
    w = K["Pclean"][:3, :3].T @ P[:3, :3]
    assert 0.01 < np.arccos((np.trace(w) - 1) / 2) < 0.5 and 0.01 < np.linalg.norm(P[:3, 3] - K["Pclean"][:3, 3]) < 0.5


@pytest.mark.parametrize("plane", [0, 1])
@pytest.mark.parametrize("param", PARAMS)
def test_pairwise_known_answer_reference_inputs(orc, param, plane):
    """main_pairwise.cpp:44-61,117-133 on its own inputs (all 16 264 rows of cloudXYZ_0, the default-seeded P).  README.md:141-146:
    diff_tra 6.3e-11 .. 7.8e-11, diff_rot 1.7e-6 deg (poseDiff's acos floor) for the three Ceres variants (point-to-point).
    The solve ends on Ceres' parameter tolerance WITHOUT taking the last step, so the answer's error is the size of that
    untaken step: anywhere below 1e-8 * |x|; here it lands at 3e-10 .. 8e-10 (point-to-point) and 5e-11 .. 2e-10 (point-to-plane),
    the README's sample at 6e-11 .. 8e-11.  Bar: one decade above the README figure, same termination type."""
    K = kat()
    pts, nrm, P = K["pts"], K["nor"], K["P"]
    dstp = pts @ P[:3, :3].T + P[:3, 3]
    dstn = nrm @ P[:3, :3].T
    N = len(pts)
    corr = [(np.arange(N, dtype=np.int32), np.arange(N, dtype=np.int32))]
    prob = orc.make_problem([dstp, pts], [dstn, nrm], [1, 0], [1], [0], corr, [0.0], param, plane, 0)
    Pout, sm = orc.optimize(prob, np.array([np.eye(4), np.eye(4)]), 50)
    dt, dr_deg = orc.pose_diff(P, Pout[1])       # the reference's own poseDiff (acos form, degrees)
    assert sm["termination"] == 2 and 5 <= sm["iterations"] <= 10, sm
    assert dt <= 1e-9 and dr_deg <= 2e-6, (param, plane, dt, dr_deg)
    assert synth.pose_diff(P, Pout[1])[1] <= 3e-9   # radians, atan2 form (no acos floor)
    # the value this restatement actually reaches is recorded in the golden (3.2e-10 / 4.0e-10 / 8.2e-10 point-to-point, 5.1e-11 / 6.6e-11 /
    # 1.6e-10 point-to-plane): a drift towards the 1e-9 bar shows up here long before the bar itself fails
    assert dt <= 1.25 * K["reached_dt"][plane, param] + 1e-13 and sm["iterations"] == K["reached_iters"][plane, param], (param, plane, dt, sm)


def readme_problem(orc, K, param, stdlib="libc++", phantom=True):
    """The reference's own run of main_pairwise.cpp (point-to-point): cloudXYZ_0 as ITS loadXYZ delivers it (the last row twice,
    common.h:233-238), P = addNoise(Pclean, 0.1, 0.1) from the default-seeded mt19937 through the given standard library's
    std::normal_distribution, index-aligned pairs (dst[i], src[i]), frame 0 = dst fixed at identity, frame 1 = src from identity."""
    pts = np.vstack([K["pts"], K["pts"][-1:]]) if phantom else K["pts"]
    P = orc.add_noise(K["Pclean"], 0.1, 0.1, reset=True, stdlib=stdlib)
    dstp = pts @ P[:3, :3].T + P[:3, 3]
    ids = np.arange(len(pts), dtype=np.int32)
    return orc.make_problem([dstp, pts], [None, None], [1, 0], [1], [0], [(ids, ids)], [0.0], param, 0, 0), P


def test_readme_known_answer_reproduced(orc):
    """THE PIN OF THE LM HALF.  README.md:141-146 prints what the reference itself — real Ceres — reached on its only
    known-answer test:  `ceres CeresAngleAxis diff_tra:7.76957e-11`, `ceres EigenQuaternion diff_tra:6.31278e-11` (the third line
    re-prints the quaternion result, main_pairwise.cpp:132).  The restated trust-region loop (oracle.cpp lm_solve, Ceres-default
    schedule of icp-ceres.cpp:66-95) reproduces BOTH numbers to all six printed digits on the reference's inputs — once the inputs are
    really the reference's: the libc++ variate order of std::normal_distribution (the README run was an OS X / clang build) and the
    duplicated last row its loadXYZ appends.  diff_tra here is ~1e-6 of the last LM step, a smooth function of the whole trajectory
    (functors, local parameterizations, damping, radius updates, the parameter-tolerance stop): six digits on two parameterizations
    do not happen by accident — and the controls below show the vector tells details apart."""
    K = kat()
    assert np.array_equal(orc.add_noise(K["Pclean"], 0.1, 0.1, reset=True, stdlib="libc++"), K["P_libcxx"])
    reached = {}
    for param in PARAMS:
        prob, P = readme_problem(orc, K, param)
        Pout, sm = orc.optimize(prob, np.array([np.eye(4), np.eye(4)]), 50)
        reached[param] = orc.pose_diff(P, Pout[1])[0]
        assert sm["termination"] == 2 and sm["iterations"] == 6, sm
        assert abs(reached[param] / K["readme_reached"][param] - 1) < 1e-6, (param, reached[param])
    assert "%.5e" % reached[orclib.PARAM_ANGLEAXIS] == "7.76957e-11" and "%.5e" % reached[orclib.PARAM_QUAT] == "6.31278e-11", reached
    assert K["readme_dt"][orclib.PARAM_ANGLEAXIS] == 7.76957e-11 and K["readme_dt"][orclib.PARAM_QUAT] == 6.31278e-11
    # controls: what does NOT reproduce the README (each off in the 2nd-4th digit or by factors)
    def run(param, **kw):
        opts = {k: kw.pop(k) for k in list(kw) if k in orc.LM_OPTION_ORDER}
        orc.set_lm_options(**opts)
        try:
            prob, P = readme_problem(orc, K, param, **kw)
            Pout, _ = orc.optimize(prob, np.array([np.eye(4), np.eye(4)]), 50)
        finally:
            orc.set_lm_options()
        return orc.pose_diff(P, Pout[1])[0] / K["readme_dt"][param]
    for param in (orclib.PARAM_ANGLEAXIS, orclib.PARAM_QUAT):
        assert abs(run(param, phantom=False) - 1) > 2e-4            # without the duplicated row: 7.77251e-11 / 6.31468e-11
        assert run(param, stdlib="libstdc++") > 4                     # g++'s variate order = another P: 4.0e-10 / 3.2e-10
        assert run(param, initial_radius=3e3) > 10 and run(param, initial_radius=1e5) > 3   # initial_trust_region_radius must be 1e4
        assert run(param, radius_rule=1) < 0.3 and run(param, radius_rule=2) > 5              # the radius update must be Ceres' cubic rule
        assert run(param, parameter_tolerance=1e-6) > 100            # the stopping rule must be 1e-8 (|x| + 1e-8)


# ---------------------------------------------------------------- how much could a detail of Ceres' schedule that the README vector does not reach matter?
def _registration(orc, ref, pb, param, rounds=20):
    import cpupath
    cp = cpupath.CpuPath(pb["pts"], pb["nor"], pb["src"], pb["dst"], pb["fixed"], param, 1, orc=orc, ref=ref)
    P = pb["init"].copy()
    its = []
    for _ in range(rounds):
        P, sm = cp.round(P)
        its.append(sm["iterations"])
    cp.close()
    return P, sm["final_cost"], its


def test_schedule_sensitivity_of_the_20_round_registration(orc, refnn):
    """VERDICT r2 item 3.  Ceres is not installed, so the trust-region schedule of oracle.cpp / host/lm.cpp is restated from upstream
    and pinned to nothing.  This bounds what a wrong detail could do to the reference's whole loop (main_multiview.cpp:150-169: 20
    rounds) on a reduced config 3 (8 views, angle-axis, point-to-plane, robust).  Measured (tools/schedule_sensitivity.py,
    profiles/r03_schedule_sensitivity.txt):
      * min_relative_decrease x/÷ 10, Jacobi scaling off, min_lm_diagonal, parameter_tolerance, a x3 radius rule: final poses move by
        0 .. 1e-14 — the solve never rejects a step here and Marquardt's diag(J^T J) damping is scale-invariant;
      * initial radius x/÷ 10, function_tolerance ÷ 2, a frozen radius, or the pre-1.12 control flow (the tolerance-meeting step is
        taken): 4e-5 .. 4e-4 m / rad.  NOT within north_star's 1e-5: every solve stops on `|cost change| <= 1e-6 cost`, the outer loop
        stalls as soon as the first candidate of a round meets that, so the registration's set of fixed points is a basin of that
        diameter on this weakly constrained scene (all end states agree in cost to ~1.5e-3 relative).
    Consequence stated in DESIGN.md §6: GPU-vs-CPU-path parity (1e-8) holds because both sides run the SAME restated schedule; parity
    with a real Ceres build to 1e-5 additionally needs initial radius, function tolerance and control flow to be Ceres' — they are
    upstream's documented defaults, but only a Ceres-linked run could confirm it.  The well-conditioned pairwise KAT is immune
    (every variant <= 5e-9 m, see the profile)."""
    pb = synth.make_problem(8, 1500)
    P0, c0, its0 = _registration(orc, refnn, pb, orclib.PARAM_ANGLEAXIS)
    dev = {}
    try:
        for name, kw in (("min_relative_decrease", {"min_relative_decrease": 1e-2}), ("jacobi_scaling", {"jacobi_scaling": 0}),
                         ("radius_x3", {"radius_rule": 1}),
                         ("initial_radius", {"initial_radius": 1e3}), ("function_tolerance", {"function_tolerance": 5e-7}),
                         ("legacy_flow", {"legacy_minimizer": 1})):
            orc.set_lm_options(**kw)
            P, c, its = _registration(orc, refnn, pb, orclib.PARAM_ANGLEAXIS)
            dev[name] = (max(max(synth.pose_diff(P[k], P0[k])) for k in range(8)), abs(c - c0) / c0, its)
    finally:
        orc.set_lm_options()
    print(dev)
    for name in ("min_relative_decrease", "jacobi_scaling", "radius_x3"):
        assert dev[name][0] <= 1e-9, (name, dev[name])
    for name in ("initial_radius", "function_tolerance", "legacy_flow"):
        assert dev[name][0] <= 2e-3 and dev[name][1] <= 1e-2, (name, dev[name])   # bounded by the function-tolerance basin, not by 1e-5


def test_lm_options_default_restores_the_pinned_schedule(orc):
    """orc_set_lm_options with no arguments = the schedule every other test is pinned on."""
    pb, prob = small_problem(orc, orclib.PARAM_SOPHUS, 1, 1)
    a, sa = orc.optimize(prob, pb["init"], 50)
    orc.set_lm_options(initial_radius=10.0)
    b, sb = orc.optimize(prob, pb["init"], 50)
    orc.set_lm_options()
    c, sc = orc.optimize(prob, pb["init"], 50)
    assert np.array_equal(a, c) and sa == sc and not np.array_equal(a, b)


def test_closed_form_point_to_point_is_kabsch():
    """ICP_Closedform::pointToPoint (icp-closedform.cpp:9-26) through the host-only C ABI: on the KAT it returns P (README.md:148:
    diff_tra 6.6e-15); on noisy pairs it equals the SVD (Kabsch) solution the reference computes."""
    from mvicp import lib as L
    K = kat()
    pts, P = K["pts"], K["P"]
    dstp = pts @ P[:3, :3].T + P[:3, 3]
    T = L.closedform_point_to_point(pts, dstp)
    dt, dr = synth.pose_diff(P, T)
    assert dt < 1e-13 and dr < 1e-13, (dt, dr)
    rng = np.random.default_rng(7)
    noisy = dstp + rng.normal(0, 2e-3, dstp.shape)
    T = L.closedform_point_to_point(pts, noisy)
    pm, qm = pts.mean(0), noisy.mean(0)
    Kc = (noisy - qm).T @ (pts - pm)                      # icp-closedform.cpp:19
    U, S, Vt = np.linalg.svd(Kc)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R = U @ np.diag([1, 1, -1]) @ Vt
    assert np.allclose(T[:3, :3], R, atol=1e-12) and np.allclose(T[:3, 3], qm - R @ pm, atol=1e-13)
    with pytest.raises(Exception):
        L.closedform_point_to_point(pts[:2], dstp[:2])


def test_closed_form_point_to_plane_matches_the_6x6_system():
    """ICP_Closedform::pointToPlane (icp-closedform.cpp:30-54): C x = d with rows [p x n ; n], R = Rx Ry Rz of the solved angles."""
    from mvicp import lib as L
    K = kat()
    pts, nrm = K["pts"][::4], K["nor"][::4]
    small = np.eye(4); small[:3, :3] = synth.so3_exp(np.array([0.01, -0.02, 0.015])); small[:3, 3] = [0.002, -0.001, 0.003]
    dstp = pts @ small[:3, :3].T + small[:3, 3]
    dstn = nrm @ small[:3, :3].T
    T = L.closedform_point_to_plane(pts, dstp, dstn)
    u = np.hstack([np.cross(pts, dstn), dstn])
    r = np.sum((pts - dstp) * dstn, axis=1)
    x = np.linalg.solve(u.T @ u, -(u.T @ r))
    c, s = np.cos, np.sin
    Rx = np.array([[1, 0, 0], [0, c(x[0]), -s(x[0])], [0, s(x[0]), c(x[0])]])
    Ry = np.array([[c(x[1]), 0, s(x[1])], [0, 1, 0], [-s(x[1]), 0, c(x[1])]])
    Rz = np.array([[c(x[2]), -s(x[2]), 0], [s(x[2]), c(x[2]), 0], [0, 0, 1]])
    assert np.allclose(T[:3, :3], Rx @ Ry @ Rz, atol=1e-12) and np.allclose(T[:3, 3], x[3:], atol=1e-12)
    dt, dr = synth.pose_diff(small, T)
    assert dt < 2e-4 and dr < 1e-3   # one linearised step from identity: first-order accurate (README.md:148 lists the p2p variant only)
