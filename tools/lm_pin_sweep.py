"""try to pin the LM half on the ONE numeric vector the reference publishes for it.

/root/reference/README.md:141-146 (pairwise known-answer test, src/main_pairwise.cpp:34-61,117-133, point-to-point, real Ceres):
    ceres CeresAngleAxis   diff_tra:7.76957e-11
    ceres EigenQuaternion  diff_tra:6.31278e-11        (the "SophusSE3" line re-prints the quaternion result, main_pairwise.cpp:132)

What this script varies (CPU only; the oracle restatement of the Ceres trust-region loop, oracle/oracle.cpp lm_solve):
  * the noisy pose P = addNoise(Pclean, 0.1, 0.1) (include/common.h:36-67): a default-seeded std::mt19937 (standard-mandated
    stream) through std::normal_distribution, whose ALGORITHM is implementation-defined.  libstdc++ (g++) and libc++ (clang on the
    author's OS X, README "Mac OSX (>=El Capitan)") both use the Marsaglia polar method on the same uniform stream but hand out the two
    variates of a pair in opposite order; both are generated here (the libstdc++ one is checked against the compiled oracle);
  * the phantom trailing pair of the reference's loadXYZ (include/common.h:233-238): with / without;
  * every control-flow / constant of the trust-region schedule that is upstream knowledge (orc_set_lm_options).
Output: one row per combination with the two numbers, the ratio to the README's, and the verdict.  Run in the build container:
    python tools/lm_pin_sweep.py > profiles/r05_lm_pin_sweep.txt
"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import orclib  # noqa: E402

README_AA, README_QUAT = 7.76957e-11, 6.31278e-11


def mt19937_raw(n):
    bg = np.random.MT19937()
    bg._legacy_seeding(5489)   # std::mt19937's default seed: the standard-mandated stream
    return bg.random_raw(n)


def canonical(raw, pos):
    """std::generate_canonical<double, 53>(mt19937): two 32-bit draws, low word first, in double arithmetic."""
    s = float(raw[pos]) + float(raw[pos + 1]) * 4294967296.0
    r = s / 18446744073709551616.0
    if r >= 1.0:
        r = np.nextafter(1.0, 0.0)
    return r, pos + 2


def normals(order, count=6):
    """`count` draws of std::normal_distribution<double>(0, 1) from a default-seeded mt19937.
    order = 'libstdc++': returns y * mult first, keeps x * mult; 'libc++': returns u * F first, keeps v * F."""
    raw = mt19937_raw(4096)
    pos, out, saved = 0, [], None
    while len(out) < count:
        if saved is not None:
            out.append(saved); saved = None
            continue
        while True:
            a, pos = canonical(raw, pos); b, pos = canonical(raw, pos)
            x, y = 2.0 * a - 1.0, 2.0 * b - 1.0
            r2 = x * x + y * y
            if not (r2 > 1.0 or r2 == 0.0):
                break
        mult = np.sqrt(-2.0 * np.log(r2) / r2)
        if order == "libstdc++":
            out.append(y * mult); saved = x * mult
        else:
            out.append(x * mult); saved = y * mult
    return np.array(out)


def add_noise(Pclean, sigma, sigmat, order):
    z = normals(order)
    w = z[:3] * sigma
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    Rw = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * (W @ W)
    P = Pclean.copy()
    P[:3, :3] = Pclean[:3, :3] @ Rw
    P[:3, 3] = Pclean[:3, 3] + z[3:] * sigmat
    return P


def main():
    orc = orclib.load()
    g = np.load(os.path.join(ROOT, "tests", "golden", "pairwise_kat.npz"))
    pts0, Pclean = g["pts"], g["Pclean"]
    P_std = add_noise(Pclean, 0.1, 0.1, "libstdc++")
    P_cxx = add_noise(Pclean, 0.1, 0.1, "libc++")
    P_compiled = orc.add_noise(Pclean, 0.1, 0.1, reset=True)
    print("# python libstdc++ emulation vs the compiled oracle (g++ <random>): max |dP| = %.3e" % np.abs(P_std - P_compiled).max())
    assert np.abs(P_std - P_compiled).max() < 1e-14
    print("# |P_libstdc++ - P_libc++| max = %.3e  (different noise draw -> a different known-answer problem)" % np.abs(P_std - P_cxx).max())
    print("# README.md:141-146: angle-axis %.5e  quaternion %.5e  (ratio aa/quat %.4f)" % (README_AA, README_QUAT, README_AA / README_QUAT))

    schedules = []
    for radius_rule, legacy, jacobi, r0 in itertools.product((0, 1, 2), (0, 1), (1, 0), (1e4, 1e3, 1e5, 1e2, 1e6, 3e4, 3e3)):
        schedules.append(dict(radius_rule=radius_rule, legacy_minimizer=legacy, jacobi_scaling=jacobi, initial_radius=r0))
    extra = [dict(function_tolerance=t) for t in (1e-5, 1e-7, 1e-8, 1e-10, 1e-12)] + [dict(parameter_tolerance=t) for t in (1e-6, 1e-7, 1e-9, 1e-10)] + \
            [dict(min_diag=t) for t in (1e-8, 1e-4, 1e-2)] + [dict(min_relative_decrease=t) for t in (1e-4, 1e-2, 0.25)] + \
            [dict(function_tolerance=t, legacy_minimizer=1) for t in (1e-5, 1e-7, 1e-8)]
    schedules += extra

    rows = []
    for pname, P in (("libstdc++", P_std), ("libc++", P_cxx)):
        for phantom in (0, 1):
            pts = np.vstack([pts0, pts0[-1:]]) if phantom else pts0
            dstp = pts @ P[:3, :3].T + P[:3, 3]
            ids = np.arange(len(pts), dtype=np.int32)
            for sch in schedules:
                orc.set_lm_options(**sch)
                res = {}
                for param, tag in ((orclib.PARAM_ANGLEAXIS, "aa"), (orclib.PARAM_QUAT, "quat")):
                    prob = orc.make_problem([dstp, pts], [None, None], [1, 0], [1], [0], [(ids, ids)], [0.0], param, 0, 0)
                    Pout, sm = orc.optimize(prob, np.array([np.eye(4), np.eye(4)]), 50)
                    res[tag] = (orc.pose_diff(P, Pout[1])[0], sm["iterations"], sm["termination"])
                rows.append((pname, phantom, sch, res))
    orc.set_lm_options()

    def score(r):
        return abs(np.log(r[3]["aa"][0] / README_AA)) + abs(np.log(r[3]["quat"][0] / README_QUAT))

    print("# %d combinations; columns: noise stream | phantom pair | schedule (only non-default entries) | aa diff_tra / iters / term | quat diff_tra / iters / term | aa/README quat/README" % len(rows))
    best = sorted(rows, key=score)
    for title, lst in (("## the 25 combinations nearest to the README's two numbers (sum of |log ratio|)", best[:25]), ("## all combinations", rows)):
        print(title)
        for pname, phantom, sch, res in lst:
            s = " ".join("%s=%g" % kv for kv in sorted(sch.items()) if orclib.Oracle.LM_DEFAULTS[kv[0]] != kv[1]) or "default"
            print("%-10s phantom=%d  %-62s aa %.5e /%2d /%d   quat %.5e /%2d /%d   x%.3f x%.3f" % (
                pname, phantom, s, res["aa"][0], res["aa"][1], res["aa"][2], res["quat"][0], res["quat"][1], res["quat"][2],
                res["aa"][0] / README_AA, res["quat"][0] / README_QUAT))
    ok = [r for r in rows if abs(r[3]["aa"][0] / README_AA - 1) < 0.005 and abs(r[3]["quat"][0] / README_QUAT - 1) < 0.005]
    print("## verdict: %s" % ("%d combination(s) reproduce BOTH numbers to >= 2 significant digits" % len(ok) if ok else
                              "no combination reproduces both README numbers to 2 significant digits"))


if __name__ == "__main__":
    main()
