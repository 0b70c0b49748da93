"""Micro-benchmark of the NN kernels alone (GPU box): grid vs tree-only, at the noisy initial poses (round-1
regime: every query is 'far') and at ground truth (converged regime)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import numpy as np
import mvicp
from mvicp import lib as L, synth

K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
targets = [float(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [6.0]
pb = synth.make_problem(K, N)
for tgt in targets:
    eng = mvicp.Engine(0)
    eng.set_option("grid_target", tgt)
    eng.set_frames(pb["pts"], pb["nor"])
    eng.set_graph(pb["src"], pb["dst"])
    eng.profile(True)
    for name, poses in (("init", pb["init"]), ("gt", pb["gt"])):
        for mode in ("grid", "hash-only", "tile"):
            M = L.NN_TILE if mode == "tile" else L.NN_GRID
            eng.set_option("nn_skip_far", 1 if mode == "hash-only" else 0)
            eng.set_option("nn_census", 1)
            eng.profile_reset()
            eng.correspond(poses, pb["fixed"], 0.05, M)
            _, _, b1 = eng.profile_get("nn")
            c = eng.nn_census()
            eng.set_option("nn_census", 0)
            eng.profile_reset()
            for _ in range(3):
                eng.correspond(poses, pb["fixed"], 0.05, M)
            ms, n, b = eng.profile_get("nn")
            b = b1 * n
            q = c["queries"]
            print(f"target {tgt:4.1f} {name:4s} {mode:4s}: {ms/n:8.3f} ms  {q/(ms/n)/1e3:8.1f} Mq/s  cand/q {c['candidates']/c['queries']:6.1f} nodes/q {c['nodes']/c['queries']:6.1f} far {c['far']/c['queries']:.3f}  alg {b/n/(ms/n)/1e6:8.1f} GB/s")
    eng.close()
