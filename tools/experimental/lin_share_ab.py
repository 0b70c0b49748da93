"""linearize_kernel alone at cfg4 size: p read from the shared sorted source cloud (lin_share_p = 1, default) vs from the stream's private copy (0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import mvicp
from mvicp import lib as L, synth
K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
pb = synth.make_problem(K, N)
for share, inter in ((1, 0), (1, 1), (1, 0), (1, 1), (0, 0)):
    eng = mvicp.Engine(0)
    eng.set_option("lin_share_p", share); eng.set_option("lin_interleave", inter)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    eng.correspond(pb["init"], pb["fixed"], 0.05)
    eng.profile(True)
    eng.linearize(pb["init"], 1, 1); eng.profile_reset()
    for _ in range(40):
        eng.linearize(pb["init"], 1, 1)
    ms, n, b = eng.profile_get("linearize")
    print(f"lin_share_p={share} lin_interleave={inter}: linearize {ms / n * 1e3:7.1f} us  ({b / n / 1e6:.1f} MB model bytes per launch)", flush=True)
    eng.close()
