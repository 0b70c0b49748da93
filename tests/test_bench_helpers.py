"""bench.py pieces that do not need a GPU: the CPU-baseline leg (oracle + real nanoflann on a bounded sample, started from given round
poses) and the source hash that ties a committed rocprofv3 summary to the code it measured."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from mvicp import synth  # noqa: E402


def test_cpu_baseline_runs_on_a_bounded_sample_and_reports_every_variant():
    pb = synth.make_problem(4, 1500)
    round_poses = [pb["init"], pb["gt"], pb["gt"]]
    out = bench.cpu_baseline(pb, 1, 2, len(pb["src"]), round_poses, [True, False, False], sample_views=3)
    assert out["kind"] == "port" and out["cores"] == 1 and out["value"] > 0
    v = out["variants"]
    assert "O2_1thread" in v and set(v["O2_1thread"]["by_regime"]) == {"moving", "fixed_point"}
    assert v["O2_1thread"]["by_regime"]["moving"]["round"] == 0 and v["O2_1thread"]["by_regime"]["fixed_point"]["round"] == 2
    for name, r in v.items():
        assert r["value"] > 0 and r["cores"] >= 1, name
    json.dumps(out)   # must be serialisable into the bench line


def test_source_hash_matches_the_committed_profiles_when_present():
    sha = bench.source_sha16()
    assert len(sha) == 16 and sha == bench.source_sha16()
    for name in ("r02_cfg4_w5s20_kernels.json", "r02_cfg4_w1s19_kernels.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            j = json.load(open(p))
            assert {"scopes", "source_sha16", "warmup_skipped", "timed_rounds"} <= set(j)
            # a stale summary is legal (bench.py then reports traffic = null) but worth noticing in the log
            if j["source_sha16"] != sha:
                print(f"note: {name} was measured on other sources ({j['source_sha16']} != {sha}): bench.py will not quote its traffic")
