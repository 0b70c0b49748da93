"""A scripted registration whose rounds cross every regime of mvicp_correspond (TEST INFRASTRUCTURE): cutoff changes, fixed-mask changes,
mvicp_reset_history, forced kernel methods, option flips and rounds without a solve (bit-identical poses), drawn from a seeded generator —
every rank of a sharded job and the single process it is compared with replay the SAME script.  After every round the runner records
(counts, weight bits, poses): with N > 1 ranks the skip / arm decisions of mvicp_correspond (tie_skip, far_skip, spec_arm, the bracket
select, list reuse) are taken per rank from exchanged data, and every rank's record must equal the single process's bit for bit
(tests/test_gpu_multirank.py: host-staged transport on one GPU; tests/test_gpu_rccl2.py: RCCL on two GPUs)."""
import numpy as np


def script(seed, K, rounds=14):
    """-> list of per-round event dicts (pure function of its arguments)."""
    rng = np.random.default_rng(4200 + seed)
    ev = []
    for rnd in range(rounds):
        kind = str(rng.choice(["none", "none", "cutoff", "fixed", "reset", "method", "option", "hold", "hold"])) if rnd > 0 else "none"
        e = {"kind": kind}
        if kind == "cutoff":
            e["cutoff"] = float(rng.choice([0.05, 0.02, 0.008]))
        elif kind == "fixed":
            e["frame"] = int(rng.integers(1, K))
        elif kind == "method":
            e["method"] = int(rng.choice([0, 0, 1, 2, 3]))          # AUTO, AUTO, BRUTE, GRID, TILE
        elif kind == "option":
            e["name"] = str(rng.choice(["list_reuse", "nn_cache", "sel_bracket", "tile_cache", "tile_seed", "tile_miss", "mfma_entry", "reject_cache"]))
            e["value"] = float(rng.integers(0, 2)) * (8.0 if e["name"] == "tile_miss" else 1.0)
        e["param"] = int(rng.integers(0, 3)); e["plane"] = int(rng.integers(0, 2)); e["robust"] = bool(rng.integers(0, 2))
        ev.append(e)
    return ev


def run(eng, pb, events):
    """Replays `events` on an engine that already holds the clouds and the graph.  -> list of (counts, weights-as-bytes, poses) per round."""
    poses = pb["init"].copy()
    fixed = pb["fixed"].copy()
    cutoff, method = 0.05, 0
    out = []
    for e in events:
        k = e["kind"]
        if k == "cutoff":
            cutoff = e["cutoff"]
        elif k == "fixed":
            fixed[e["frame"]] = 1 - fixed[e["frame"]]
        elif k == "reset":
            eng.reset_history()
        elif k == "method":
            method = e["method"]
        elif k == "option":
            eng.set_option(e["name"], e["value"])
        counts, weights = eng.correspond(poses, fixed, cutoff, method)
        if k != "hold" and counts.sum() > 0:
            poses, _ = eng.optimize(poses, fixed, e["param"], e["plane"], e["robust"], 50)
        out.append((counts.copy(), weights.tobytes(), poses.copy()))
    return out


def save(path, log):
    np.savez(path, counts=np.array([l[0] for l in log]), weights=np.array([np.frombuffer(l[1], dtype=np.float32) for l in log]), poses=np.array([l[2] for l in log]))


def assert_equal(path, log, tag):
    z = np.load(path)
    for r, (c, w, P) in enumerate(log):
        assert np.array_equal(z["counts"][r], c), (tag, r, "counts")
        assert z["weights"][r].tobytes() == w, (tag, r, "weights")
        assert np.array_equal(z["poses"][r], P), (tag, r, float(np.abs(z["poses"][r] - P).max()))
