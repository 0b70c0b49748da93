"""bench.py pieces that do not need a GPU: the CPU legs (reference-equivalent CPU path = oracle + real nanoflann: pose comparison on all
edges, all-cores timing, single-thread sample) and the source hash that ties a committed rocprofv3 summary to the code it measured."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from mvicp import synth  # noqa: E402


def test_cpu_reference_legs_run_and_report_every_variant():
    """bench.py's CPU legs on a small problem: the all-edge CPU-path walk (pose_diff_vs_cpu_path, all-cores timing, LM iteration counts per
    round) and the single-thread sample, weighted over a timed window that spans a registration boundary (rounds 6..20 then 1..5)."""
    import cpupath
    pb = synth.make_problem(4, 1500)
    cp = cpupath.CpuPath(pb["pts"], pb["nor"], pb["src"], pb["dst"], pb["fixed"], 2, 1)
    P = pb["init"].copy(); after = []; iters = []
    for _ in range(4):                                   # stands in for the GPU run's poses after rounds 1..4
        P, sm = cp.round(P); after.append(P.copy()); iters.append(sm["iterations"])
    cp.close()
    window = list(range(6, 21)) + list(range(1, 6))
    moved = [bool(np.any(after[r] != (pb["init"] if r == 0 else after[r - 1]))) for r in range(4)]
    out = bench.cpu_reference_legs(pb, 1, 2, after, iters, window, moved, 4)
    pd = out["pose_diff_vs_cpu_path"]
    assert pd["rounds_compared"] == 4 and pd["max_translation_m"] < 1e-9 and pd["max_rotation_rad"] < 1e-9     # the same CPU path twice (other build: last bits)
    assert pd["lm_iterations_gpu"] == pd["lm_iterations_cpu"] == iters
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    v = cb["variants"]
    assert "O2_1thread" in v and set(v["O2_1thread"]["by_regime"]) == {"moving", "fixed_point"}
    allc = [r for k, r in v.items() if "allcores" in k or "threadpool" in k]
    assert len(allc) == 1 and len(allc[0]["per_round"]) == 4 and allc[0]["edges"] == len(pb["src"])
    for name, r in v.items():
        assert r["value"] > 0 and r["cores"] >= 1, name
    json.dumps(out)   # must be serialisable into the bench line


def test_source_hash_matches_the_committed_profiles_when_present():
    sha = bench.source_sha16()
    assert len(sha) == 16 and sha == bench.source_sha16()
    for name in ("r04_cfg4_w5s20_kernels.json", "r03_cfg4_w5s20_kernels.json", "r03_cfg4_w1s19_kernels.json", "r02_cfg4_w5s20_kernels.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            j = json.load(open(p))
            assert {"scopes", "source_sha16", "warmup_skipped", "timed_rounds"} <= set(j)
            # a stale summary is legal (bench.py then reports traffic = null) but worth noticing in the log
            if j["source_sha16"] != sha:
                print(f"note: {name} was measured on other sources ({j['source_sha16']} != {sha}): bench.py will not quote its traffic")


def test_gpus_n_relaunches_itself_as_n_ranks():
    """`python bench.py --gpus N` started WITHOUT a launcher (RANK unset) must turn into the driver's multi-GPU form — N ranks on this node,
    rendezvous on 127.0.0.1 — instead of silently measuring one rank under an N-GPU label."""
    argv = bench.self_launch_argv(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29517)
    assert argv[0] == sys.executable and argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in argv and argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29517"
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert 1024 < bench.free_port() < 65536


def test_gpus_more_than_visible_devices_is_an_error_not_a_smaller_job():
    """On a box with fewer GPUs than --gpus asks for (here: none) bench.py exits non-zero with a one-line reason and prints no JSON line."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "--gpus 2 needs 2 visible GPUs" in r.stderr
    # a launcher that started a different number of ranks than --gpus says is refused as well
    env.update(RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE=4 but --gpus 2" in r.stderr


def test_compact_line_fits_the_driver_tail_and_round_trips():
    """BENCH_r05.parsed was null because the printed line had grown to 20 KB and the driver keeps an 8-KB tail.  The printed line is now built by
    bench.compact_line from the full record (which goes to --detail-file): it must stay under 4 KB, parse, and carry the contract's keys with
    `roofline` and `cpu_baseline` — checked on the recorded full records of earlier rounds (20 KB / 12 KB / 9 KB)."""
    must = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline", "source_sha16", "detail_file"}
    seen = 0
    for name in ("r06_cfg4_w5s20_bench_line.json", "r06_cfg5_bench_line.json", "r06_cfg4_partial_bench_line.json", "r05_cfg4_w5s20_bench_line.json", "r05_cfg4_bench_line.json", "r04_cfg4_w5s20_bench_line.json", "r05_cfg5_bench_line.json", "r05_shard8_bench_line.json"):
        p = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(p):
            continue
        full = json.loads([l for l in open(p).read().splitlines() if l.startswith("{")][-1])
        line = bench.compact_line(full, "bench_detail.json")
        assert len(line) <= bench.COMPACT_LIMIT < 6000 and "\n" not in line, (name, len(line))
        d = json.loads(line)
        need = must - ({"cpu_baseline"} if "cpu_baseline" not in full else set())
        assert need <= set(d), (name, need - set(d))
        assert d["value"] == float("%.6g" % full["value"]) and d["config"]["workload"] == full["config"]["workload"] and "model" not in d["config"]
        assert abs(d["ms_per_step"] * d["steps"] / 1e3 * d["value"] / d["steps"] - 1.0) < 1e-4          # value x ms_per_step consistent after rounding
        r = d["roofline"]
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_us"} <= set(r) and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-5
        if "cpu_baseline" in full:
            assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and len(d["cpu_baseline"]["sample"]) <= 300
        seen += 1
    assert seen >= 1
    # a pathological record (long strings everywhere) still fits: optional blocks are dropped before the limit is broken
    full["dropin"] = {"copyback": {"it_per_s_after_first_round": 1.0, "first_round_ms": 2.0}, "device_only": {"it_per_s_after_first_round": 1.0, "first_round_ms": 2.0}, "fixed_point_note": "x" * 5000}
    assert len(bench.compact_line(full, "d.json")) <= bench.COMPACT_LIMIT
