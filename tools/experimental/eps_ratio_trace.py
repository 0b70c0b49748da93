"""Per round: the AUTO policy's eps / mu (median displacement bound of the queries since the last search, in guard bands) next to the temporal-cache hit
fraction and the NN-stage ms.  python tools/experimental/eps_ratio_trace.py K N ROUNDS [name=value ...]   (AB_WORKLOAD=cfg4_partial as tools/tile_ab.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import numpy as np
import mvicp
from mvicp import lib as L, synth
K, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
EXTRA = {"cfg4_partial": {"cone_deg": 20.0, "sigma": 0.004, "sigmat": 0.002, "cutoff": 0.005}}.get(os.environ.get("AB_WORKLOAD", ""), {})
CUTOFF = EXTRA.pop("cutoff", 0.05)
pb = synth.make_problem(K, N, **EXTRA)
eng = mvicp.Engine(0)
eng.set_option("cache_mfma_ratio", 1e30)     # record the ratio, never act on it
for kv in sys.argv[4:]:
    k, v = kv.split("="); eng.set_option(k, float(v))
eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
eng.profile(True); eng.set_option("nn_census", 1)
poses = pb["init"].copy()
for r in range(R):
    eng.profile_reset()
    c, w = eng.correspond(poses, pb["fixed"], CUTOFF, L.NN_AUTO)
    ms = eng.profile_get("nn")[0]
    ratio, n, _ = eng.profile_get("auto.eps_over_mu")
    cs = eng.nn_census()
    poses, sm = eng.optimize(poses, pb["fixed"], 2, 1, 1, 50)
    print("round %2d  nn %.3f ms  hit %.3f  eps/mu %s" % (r + 1, ms, cs["hits"] / max(cs["queries"], 1), ("%.3g" % ratio) if n else "-"), flush=True)
eng.close()
