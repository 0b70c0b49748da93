"""linearize_kernel alone: chunk size (correspondences per workgroup) with the XCD-aligned interleaving on.  python tools/experimental/lin_chunk_ab.py K N chunk ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import mvicp
from mvicp import lib as L, synth
K, N = int(sys.argv[1]), int(sys.argv[2])
pb = synth.make_problem(K, N)
for chunk in [int(x) for x in sys.argv[3:]] * 2:
    eng = mvicp.Engine(0)
    if chunk:
        eng.set_option("lin_chunk", chunk)
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    eng.correspond(pb["init"], pb["fixed"], 0.05)
    eng.profile(True)
    eng.linearize(pb["init"], 1, 1); eng.profile_reset()
    for _ in range(40):
        eng.linearize(pb["init"], 1, 1)
    ms, n, b = eng.profile_get("linearize")
    ms2, n2, _ = eng.profile_get("reduce")
    print(f"lin_chunk={chunk or 'auto'}: linearize {ms / n * 1e3:7.1f} us   reduce {ms2 / n2 * 1e3:5.1f} us", flush=True)
    eng.close()
