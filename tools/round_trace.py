"""Per-round NN kernel time + median correspondence distance for the grid and tile kernels (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import numpy as np
import mvicp
from mvicp import lib as L, synth
K, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pb = synth.make_problem(K, N)
for name, M in (("grid", L.NN_GRID), ("tile", L.NN_TILE)):
    eng = mvicp.Engine(0)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    eng.profile(True)
    poses = pb["init"].copy()
    out = []
    for r in range(R):
        eng.profile_reset()
        c, w = eng.correspond(poses, pb["fixed"], 0.05, M)
        ms, n, b = eng.profile_get("nn")
        poses, sm = eng.optimize(poses, pb["fixed"], 2, 1, 1, 50)
        out.append((round(ms, 2), round(float(np.mean(w)) / 1.5 * 1e3, 3), sm["iterations"]))
    print(name, "nn_ms, mean median-dist mm, lm iters:", out)
    eng.close()
