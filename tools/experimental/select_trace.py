"""Per round: select-stage ms, whether the queued first evaluation was served (spec.hit; 0 after a bracket miss), evaluations.  python tools/experimental/select_trace.py K N ROUNDS [name=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import numpy as np
import mvicp
from mvicp import lib as L, synth
K, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pb = synth.make_problem(K, N)
eng = mvicp.Engine(0)
for kv in sys.argv[4:]:
    k, v = kv.split("="); eng.set_option(k, float(v))
eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
eng.profile(True)
poses = pb["init"].copy()
import time
for r in range(R):
    eng.profile_reset()
    t0 = time.perf_counter()
    c, w = eng.correspond(poses, pb["fixed"], 0.05, L.NN_AUTO)
    t1 = time.perf_counter()
    sel = eng.profile_get("select")
    poses, sm = eng.optimize(poses, pb["fixed"], 2, 1, 1, 50)
    print("round %2d  correspond %.3f ms  select %.3f ms in %d scopes  spec.hit %d  evals %d  median w %.6g" % (r + 1, (t1 - t0) * 1e3, sel[0], sel[1], eng.profile_get("spec.hit")[1], sm["evaluations"], float(np.median(w))), flush=True)
eng.close()
