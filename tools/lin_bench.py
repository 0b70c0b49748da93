"""Micro-benchmark of the linearize kernel alone (GPU box): explicit random correspondences, stream > 256 MiB so the
Infinity Cache cannot hold it between launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import numpy as np
import mvicp
from mvicp import lib as L, synth

K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
reps = 20
pb = synth.make_problem(K, N)
eng = mvicp.Engine(0)
if len(sys.argv) > 3:
    eng.set_option("lin_chunk", int(sys.argv[3]))
eng.set_frames(pb["pts"], pb["nor"])
eng.set_graph(pb["src"], pb["dst"])
eng.correspond(pb["init"], pb["fixed"], 0.05, L.NN_GRID)
eng.profile(True)
for plane, robust in ((1, 1), (0, 1), (1, 0)):
    eng.linearize(pb["init"], plane, robust)
    eng.profile_reset()
    for _ in range(reps):
        eng.linearize(pb["init"], plane, robust)
    ms, n, b = eng.profile_get("linearize")
    ms2, n2, _ = eng.profile_get("reduce")
    print(f"plane {plane} robust {robust}: linearize {ms/n*1e3:8.1f} us  {b/n/(ms/n)/1e6:8.1f} GB/s   reduce {ms2/n2*1e3:6.1f} us   ({b/n/1e6:.1f} MB/launch)")
eng.close()
