"""A/B of tile-kernel variants on the GPU box: per-round NN stage ms of the AUTO method (and the forced tile method) for each option set,
with a bit-exactness check of every variant's counts / weights / poses against the first one.
    python tools/tile_ab.py K N ROUNDS "name=value,name=value" "..." ...      (an empty string = library defaults)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
import numpy as np
import mvicp
from mvicp import lib as L, synth
K, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variants = sys.argv[4:] or [""]
method = {"auto": L.NN_AUTO, "tile": L.NN_TILE}[os.environ.get("AB_METHOD", "auto")]
# AB_WORKLOAD=cfg4_partial: bench.py's partial-overlap variant (20-degree views, 5 mm cutoff) at the K, N given
EXTRA = {"cfg4_partial": {"cone_deg": 20.0, "sigma": 0.004, "sigmat": 0.002, "cutoff": 0.005}}.get(os.environ.get("AB_WORKLOAD", ""), {})
CUTOFF = EXTRA.pop("cutoff", 0.05)
pb = synth.make_problem(K, N, **EXTRA)
ref = None
for v in variants:
    eng = mvicp.Engine(0)
    for kv in [x for x in v.split(",") if x]:
        k, val = kv.split("="); eng.set_option(k, float(val))
    eng.set_frames(pb["pts"], pb["nor"]); eng.set_graph(pb["src"], pb["dst"])
    eng.profile(True)
    if os.environ.get("AB_CENSUS") == "1":
        eng.set_option("nn_census", 1)
    poses = pb["init"].copy(); rows = []; trace = []
    for r in range(R):
        eng.profile_reset()
        c, w = eng.correspond(poses, pb["fixed"], CUTOFF, method)
        ms = eng.profile_get("nn")[0]
        cs = eng.nn_census() if os.environ.get("AB_CENSUS") == "1" else None
        poses, sm = eng.optimize(poses, pb["fixed"], 2, 1, 1, 50)
        if cs is None:
            rows.append(round(ms, 3))
        else:   # ms, candidates per query, [per wave: second screens, confirmation rounds, blocks], fp64 confirmations per query, temporal-cache hit fraction
            wv = max(cs["queries"], 1) / 64.0
            rows.append((round(ms, 3), round(float(cs["candidates"] / max(cs["queries"], 1)), 1), round(float(cs["rescreens"] / wv), 2), round(float(cs["confirm_rounds"] / wv), 2),
                         round(float(cs["blocks"] / wv), 2), round(float(cs["confirmations"] / max(cs["queries"], 1)), 2), "hit %.3f" % float(cs["hits"] / max(cs["queries"], 1))))
        trace.append((c.copy(), w.copy(), poses.copy()))
    same = None
    if ref is None:
        ref = trace
    else:
        same = all(np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2]) for a, b in zip(ref, trace))
    import hashlib
    dg = hashlib.sha256(b"".join(a[0].tobytes() + a[1].tobytes() + a[2].tobytes() for a in trace)).hexdigest()[:12]   # compare across processes (MVICP_LIB builds)
    print(f"[{v or 'defaults'}] nn_ms per round: {rows}  sum {sum(x if not isinstance(x, tuple) else x[0] for x in rows):.3f}  identical_to_first: {same}  digest {dg}", flush=True)
    eng.close()
