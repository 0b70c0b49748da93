"""Per-round NN kernel time + median correspondence distance for the grid and tile kernels (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import numpy as np
import mvicp
from mvicp import lib as L, synth
K, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
OPTS = [kv.split("=") for kv in sys.argv[4:]]   # library options name=value
METHODS = os.environ.get("TRACE_METHODS", "grid,tile").split(",")
pb = synth.make_problem(K, N)
for name, M in (("grid", L.NN_GRID), ("tile", L.NN_TILE), ("auto", L.NN_AUTO)):
    if name not in METHODS:
        continue
    eng = mvicp.Engine(0)
    for k, v in OPTS:
        eng.set_option(k, float(v))
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    eng.profile(True)
    census = os.environ.get("TRACE_CENSUS") == "1"
    if census:
        eng.set_option("nn_census", 1)
    poses = pb["init"].copy()
    out = []
    prev = {"candidates": 0.0, "queries": 0.0, "hits": 0.0, "far": 0.0}
    for r in range(R):
        eng.profile_reset()
        c, w = eng.correspond(poses, pb["fixed"], 0.05, M)
        ms, n, b = eng.profile_get("nn")
        sel_ms = eng.profile_get("select")[0]
        poses, sm = eng.optimize(poses, pb["fixed"], 2, 1, 1, 50)
        row = (round(ms, 2), round(float(np.mean(w)) / 1.5 * 1e3, 3), sm["iterations"], round(sel_ms, 3), round(float(np.sqrt(2 * sm["final_cost"] / max(1, c.sum()))) * 1e3, 3))
        if census:
            cs = eng.nn_census()
            dq = max(cs["queries"], 1.0)   # profile_reset() zeroes the census too: per-round figures
            row += (float(round(cs["candidates"] / dq, 1)), float(round(cs["hits"] / dq, 3)), float(round(cs["far"] / dq, 3)))
        out.append(row)
    print(name, "nn_ms, mean median-dist mm, lm iters, select_ms, rms residual after LM mm [, cand/q, cache-hit, far]:", out)
    eng.close()
