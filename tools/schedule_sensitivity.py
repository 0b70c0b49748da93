#!/usr/bin/env python3
"""How far can the registration move if Ceres' real trust-region schedule differs from the restated one?  

Ceres is absent from this image and from the GPU box, so the LM half of the oracle (oracle/oracle.cpp lm_solve = host/lm.cpp) is a
restatement of upstream's algorithm that nothing here can pin.  This tool perturbs every [upstream] constant / rule of that loop
(oracle orc_set_lm_options) and re-runs (a) the reference's 20-round multiview loop (src/main_multiview.cpp:150-169) on the CPU path
(real nanoflann + oracle LM) at a reduced cfg3 (8 views x N points, point-to-plane, robust) for two parameterizations and (b) the
reference's pairwise known-answer test (src/main_pairwise.cpp:44-61) — and reports how far the poses move relative to the default
schedule.  TEST INFRASTRUCTURE: CPU only, never part of the product path.

    python tools/schedule_sensitivity.py [--n 2000] > profiles/r03_schedule_sensitivity.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpupath  # noqa: E402
import orclib  # noqa: E402
from mvicp import synth  # noqa: E402

VARIANTS = [
    ("default (Ceres defaults, icp-ceres.cpp:66-89)", {}),
    ("initial_trust_region_radius 1e3", {"initial_radius": 1e3}),
    ("initial_trust_region_radius 1e5", {"initial_radius": 1e5}),
    ("min_relative_decrease 1e-4", {"min_relative_decrease": 1e-4}),
    ("min_relative_decrease 1e-2", {"min_relative_decrease": 1e-2}),
    ("jacobi_scaling off", {"jacobi_scaling": 0}),
    ("min_lm_diagonal 1e-8", {"min_diag": 1e-8}),
    ("parameter_tolerance 1e-9", {"parameter_tolerance": 1e-9}),
    ("function_tolerance 5e-7 (/2)", {"function_tolerance": 5e-7}),
    ("function_tolerance 2e-6 (x2)", {"function_tolerance": 2e-6}),
    ("radius x3 on every accepted step", {"radius_rule": 1}),
    ("radius unchanged on accepted steps", {"radius_rule": 2}),
    ("pre-1.12 minimizer flow (tolerance-meeting step is taken)", {"legacy_minimizer": 1}),
]


def registration(orc, ref, pb, param, rounds=20):
    cp = cpupath.CpuPath(pb["pts"], pb["nor"], pb["src"], pb["dst"], pb["fixed"], param, 1, orc=orc, ref=ref)
    P = pb["init"].copy()
    traj, its, costs = [], [], []
    for _ in range(rounds):
        P, sm = cp.round(P)
        traj.append(P); its.append(sm["iterations"]); costs.append(sm["final_cost"])
    cp.close()
    return traj, its, costs


def pose_dev(A, B):
    return max(max(synth.pose_diff(a, b)) for a, b in zip(A, B))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--views", type=int, default=8)
    args = ap.parse_args()
    orc, ref = cpupath.load_libs(False)
    pb = synth.make_problem(args.views, args.n)
    print(f"# schedule sensitivity of the 20-round registration: {args.views} views x {args.n} points, point-to-plane, robust, cutoff 0.05")
    print("# deviation = max over views of max(|dt| m, angle rad) against the DEFAULT schedule's poses at the same round")
    for param, pname in ((orclib.PARAM_ANGLEAXIS, "angle-axis"), (orclib.PARAM_SOPHUS, "SophusSE3")):
        base = None
        print(f"\n## {pname}")
        print(f"{'variant':58s} {'final dev':>10s} {'max dev':>10s} {'rel. final-cost diff':>21s}  LM iterations per round")
        for name, kw in VARIANTS:
            orc.set_lm_options(**kw)
            traj, its, costs = registration(orc, ref, pb, param)
            if base is None:
                base = (traj, costs)
            devs = [pose_dev(a, b) for a, b in zip(traj, base[0])]
            print(f"{name:58s} {devs[-1]:10.2e} {max(devs):10.2e} {abs(costs[-1] - base[1][-1]) / base[1][-1]:21.2e}  {its}")
    orc.set_lm_options()
    # ---- the pairwise known-answer test (zero-residual, well conditioned)
    K = np.load(os.path.join(ROOT, "tests", "golden", "pairwise_kat.npz"))
    pts, nrm, P = K["pts"], K["nor"], K["P"]
    dstp = pts @ P[:3, :3].T + P[:3, 3]; dstn = nrm @ P[:3, :3].T
    ids = np.arange(len(pts), dtype=np.int32)
    print("\n## pairwise known-answer test (main_pairwise.cpp:44-61; README.md:141-146: diff_tra 6.3e-11 .. 7.8e-11): diff_tra m / LM iterations")
    print(f"{'variant':58s} " + " ".join(f"{c + ' ' + p:>16s}" for c in ("p2p", "p2plane") for p in ("quat", "aa", "sophus")))
    for name, kw in VARIANTS:
        orc.set_lm_options(**kw)
        row = []
        for plane in (0, 1):
            for param in (0, 1, 2):
                prob = orc.make_problem([dstp, pts], [dstn, nrm], [1, 0], [1], [0], [(ids, ids)], [0.0], param, plane, 0)
                Pout, sm = orc.optimize(prob, np.array([np.eye(4), np.eye(4)]), 50)
                row.append(f"{orc.pose_diff(P, Pout[1])[0]:.1e}/{sm['iterations']:d}")
        print(f"{name:58s} " + " ".join(f"{r:>16s}" for r in row))
    orc.set_lm_options()


if __name__ == "__main__":
    main()
