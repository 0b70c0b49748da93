"""Where the CYCLES of nn_mfma_kernel go (tuning build with -DMVICP_MFMA_TIMING through MVICP_LIB): per ICP round of cfg4's AUTO method, the
shader-clock cycles per wave spent in each section of the kernel (the census slots of that build carry clock64() deltas, see nn_mfma.hip).
    MVICP_LIB=tools/_build/libmvicp_timing.so python tools/experimental/mfma_timing.py 32 200000 5"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import numpy as np
import mvicp
from mvicp import lib as L, synth
K, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pb = synth.make_problem(K, N)
eng = mvicp.Engine(0)
eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
eng.profile(True); eng.set_option("nn_census", 1)
poses = pb["init"].copy()
names = ["prologue", "above_blocks", "block_setup", "first_screens", "second_screens", "confirm_rounds", "refresh_pick_max", "epilogue"]
for r in range(R):
    eng.profile_reset()
    eng.correspond(poses, pb["fixed"], 0.05)
    ms = eng.profile_get("nn")[0]
    c = eng.nn_census()
    waves = max(c["queries"], 1) / 64.0
    t = [c["fetched"], c["nodes"], c["candidates"], c["hits"], c["rescreens"], c["confirm_rounds"], c["blocks"], c["confirmations"]]
    tot = sum(t)
    print("round %d  nn %.3f ms  cycles/wave %.0f : " % (r + 1, ms, tot / waves) + "  ".join("%s %.0f (%.0f%%)" % (n, v / waves, 100 * v / max(tot, 1)) for n, v in zip(names, t)), flush=True)
    poses, sm = eng.optimize(poses, pb["fixed"], 2, 1, 1, 50)
eng.close()
