#!/usr/bin/env python3
"""bench.py — ICP iterations/sec of the MI355X multiview LM-ICP hot path (BASELINE.json metric).

One "step" = one outer ICP iteration = the loop body of the reference's src/main_multiview.cpp:150-169 without
visualisation: correspondence search over all E edges (NN + cutoff + median) followed by one LM solve (<= 50 iterations;
one device linearization per LM iteration + the host solve).  The reference program IS 20 such rounds from the noisy initial
poses (main_multiview.cpp:150), so the rounds of this bench walk through that registration again and again: global round g
(warm-up first, then timed) is round g mod 20 + 1 of registration g div 20 + 1; every registration starts from the same noisy
initial poses (the reference's noise generator is default-seeded: every run of the program draws the same poses) after
mvicp_reset_history() has dropped everything the library remembers (NN cache, seeds, lists, medians, AUTO state) — inside
the timed region when it falls there.  With the driver's `--warmup 5 --steps 20` the timed window is rounds 6-20 of
registration 1 and rounds 1-5 of registration 2: each of the reference's 20 rounds exactly once.  `regimes` splits the timed
rounds by whether the LM solve still moved the poses.

    python bench.py [--gpus N --steps K --warmup W] [--workload cfg2|cfg3|cfg4|cfg5|tiny|shard8|shard8_cfg5]

N > 1: launched by torch.distributed.run, one rank per GPU; edges are sharded across ranks and the per-edge normal-equation
blocks are summed with an RCCL all-reduce per LM evaluation (strong scaling: the problem is fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (K views, N pts/view, point_to_plane, param, description)  — BASELINE.json configs 2..5
    "tiny": (4, 20_000, 1, 2, "tiny: multiview 4 views x 20k pts, point-to-plane, SophusSE3"),
    "cfg2": (2, 100_000, 1, 2, "cfg2: pairwise point-to-plane, 2 synthetic clouds x 100k pts (E=1), SophusSE3"),
    "cfg3": (8, 100_000, 1, 1, "cfg3: multiview 8 views x 100k pts, point-to-plane, angle-axis"),
    "cfg4": (32, 200_000, 1, 2, "cfg4: multiview 32 views x 200k pts, point-to-plane, SophusSE3 (E=62)"),
    "cfg5": (64, 1_000_000, 1, 2, "cfg5: multiview 64 views x 1M pts, point-to-plane, SophusSE3 (E=126)"),
    "shard8": (5, 200_000, 1, 2, "shard8: 5 views x 200k pts (E=8): the per-rank share of cfg4 on 8 GPUs, for fixed-cost analysis"),
    "shard8_cfg5": (9, 1_000_000, 1, 2, "shard8_cfg5: 9 views x 1M pts (E=16): the per-rank share of cfg5 on 8 GPUs, for fixed-cost analysis"),
    # NOT a BASELINE config: cfg4's shape with views that only partly overlap (20-degree cones: a third of a view's points has no counterpart in
    # a ring neighbour) and a cutoff (5 mm) well inside the non-overlap band, so the filter of frame.cpp:156-160 rejects ~1/3 of the queries
    # and the membership of every list changes while the poses move (compaction + gather every moving round, no shared source operands)
    "cfg4_partial": (32, 200_000, 1, 2, "cfg4_partial: 32 views x 200k pts, 20-degree views (partial overlap), cutoff 5 mm, point-to-plane, SophusSE3 (E=62)"),
}
WORKLOAD_EXTRAS = {"cfg4_partial": {"cone_deg": 20.0, "sigma": 0.004, "sigmat": 0.002, "cutoff": 0.005}}
ROUNDS_PER_REGISTRATION = 20   # main_multiview.cpp:150
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
PROFILE_ROUND = "r06"


def source_sha16():
    """Hash of the sources the device path is built from: a committed rocprofv3 summary is only quoted for the code it measured."""
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "mv-lm-icp_amd")
    files = []
    for sub in ("csrc", "host"):
        d = os.path.join(pkg, sub)
        files += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".cpp", ".h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_reference_legs(pb, plane, param, gpu_poses_after, gpu_iters, window_rounds, moved_by_round, cpu_rounds, cutoff=0.05):
    """The two CPU legs of the line, both on tests/cpupath.py = the reference-equivalent CPU path (real vendored nanoflann from
    oracle/_ref + the oracle's Jet/autodiff restatement of Ceres; Ceres itself is not installed).  Run AFTER the timed region.

    (1) `pose_diff_vs_cpu_path` (SURVEY.md §8(d) secondary metric) and the all-cores CPU timing: the CPU path walks rounds
        1..cpu_rounds of the SAME registration from the same noisy initial poses on ALL edges (its own trajectory: its own
        correspondences, its own solves), -O3 AVX2 + OpenMP on every usable core; after every round its poses are compared with the
        GPU run's poses after the same round, and its LM iteration count is listed next to the GPU's.
    (2) `cpu_baseline.value` = the reference's own build and threading (-O2, 1 thread; CMakeLists.txt:14-22, Ceres num_threads 1): one moving
        round (round 2, started from the GPU run's poses) and one fixed-point round (the last compared round), on ALL edges up to cfg4's size
        (about 110 s), on a 16-edge sample scaled by edge count beyond it (`sampled: true`).
    Both are weighted over the rounds of the timed window: a window round r <= cpu_rounds uses the measured round r, later rounds (all
    fixed-point rounds) use the last measured fixed-point round."""
    import cpupath
    K = len(pb["pts"])
    E = len(pb["src"])
    ncores = cpupath.usable_cores(64)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    out = {}
    per_round = []
    fast = cpupath.fast_build_usable()
    # ---- (1) all edges, every usable core
    cp = cpupath.CpuPath(pb["pts"], pb["nor"], pb["src"], pb["dst"], pb["fixed"], param, plane, cutoff=cutoff, fast=fast, threads=ncores)
    P = pb["init"].copy()
    dev_t, dev_r = [], []
    for r in range(cpu_rounds):
        P, sm = cp.round(P)
        dts = [synth_pose_diff(P[k], gpu_poses_after[r][k]) for k in range(K)]
        dev_t.append(max(d[0] for d in dts)); dev_r.append(max(d[1] for d in dts))
        per_round.append({"round": r + 1, "nn_s": cp.last["nn_s"], "lm_s": cp.last["lm_s"], "lm_iterations_cpu": sm["iterations"], "lm_iterations_gpu": gpu_iters[r],
                          "moved": sm["successful_steps"] > 0, "correspondences": int(cp.last["counts"].sum())})
    tree_build_s = cp.tree_build_s
    cp.close()
    out["pose_diff_vs_cpu_path"] = {
        "rounds_compared": cpu_rounds, "max_translation_m": max(dev_t), "max_rotation_rad": max(dev_r),
        "per_round_max_translation_m": dev_t, "per_round_max_rotation_rad": dev_r,
        "lm_iterations_gpu": [p["lm_iterations_gpu"] for p in per_round], "lm_iterations_cpu": [p["lm_iterations_cpu"] for p in per_round],
        "note": ("GPU path vs the reference-equivalent CPU path (real nanoflann + oracle LM, all edges), each on its own trajectory from the same noisy initial "
                 "poses, compared after each of the first rounds of the registration; target 1e-5 m / rad.  The CPU path's LM is a restatement of Ceres "
                 "(pinned on the reference's published pairwise vector, README.md:141-146, to six digits: tests/test_oracle_lm.py), see DESIGN.md section 7")}

    def window_rate(sec_of_round):
        tot = 0.0
        for r in window_rounds:                       # r = 1-based round index inside its registration
            tot += sec_of_round(r)
        return len(window_rounds) / tot, tot / len(window_rounds)

    fixed_meas = [p for p in per_round if not p["moved"]]
    last_fixed = fixed_meas[-1] if fixed_meas else per_round[-1]

    def sec_all(r):
        p = per_round[r - 1] if r <= len(per_round) else last_fixed
        return p["nn_s"] + p["lm_s"]

    rate_all, spr_all = window_rate(sec_all)
    variants = {("O3_avx2_allcores" if fast else "O2_threadpool_nn"): {
        "value": rate_all, "unit": "iterations/s", "cores": ncores, "s_per_round": spr_all, "edges": E, "sample": "ALL edges, rounds 1..%d measured one by one" % cpu_rounds,
        "per_round": per_round, "tree_build_s_once": tree_build_s}}
    # ---- (2) the reference's build and threading (-O2, one thread): ONE moving and ONE fixed-point round, started from the GPU run's poses, on ALL edges
    #      while that stays a bounded job (E x N <= 1.3e7: cfg4 = 62 x 200 k is about 110 s); beyond that on the smallest prefix of the views with >= 16
    #      edges, scaled by edge count — and then the line says `sampled: true` (a scaled sample must not pass for a measurement)
    def sample_views(min_edges):
        """smallest prefix of the views whose mutual edges number at least min_edges (all views if the graph is smaller)"""
        for ks in range(2, K + 1):
            kp = [e for e, (s_, d_) in enumerate(zip(pb["src"], pb["dst"])) if s_ < ks and d_ < ks]
            if len(kp) >= min(min_edges, E):
                return ks, kp
        return K, list(range(E))
    r_mov = 2 if cpu_rounds >= 2 and per_round[1]["moved"] else 1
    r_fix = last_fixed["round"]
    starts = {"moving": (r_mov, pb["init"] if r_mov == 1 else gpu_poses_after[r_mov - 2]), "fixed_point": (r_fix, pb["init"] if r_fix == 1 else gpu_poses_after[r_fix - 2])}
    whole = float(E) * len(pb["pts"][0]) <= 1.3e7
    Ks, keep = (K, list(range(E))) if whole else sample_views(16)
    scale = E / max(1, len(keep))
    c1 = cpupath.CpuPath(pb["pts"][:Ks], pb["nor"][:Ks], pb["src"][keep], pb["dst"][keep], pb["fixed"][:Ks], param, plane, cutoff=cutoff, fast=False, threads=1)
    c1.correspond(np.ascontiguousarray(pb["init"][:Ks]))   # build the trees outside the timed rounds
    kinds = {}
    for kind, (r, P0) in starts.items():
        _, sm = c1.round(np.ascontiguousarray(P0[:Ks]))
        kinds[kind] = {"round": r, "nn_s": c1.last["nn_s"], "lm_s": c1.last["lm_s"], "lm_iterations": sm["iterations"], "edges_measured": len(keep),
                       "s_per_round_full": (c1.last["nn_s"] + c1.last["lm_s"]) * scale}
    tree_build_o2 = c1.tree_build_s
    c1.close()
    rate, spr = window_rate(lambda r: kinds["moving" if (r <= len(moved_by_round) and moved_by_round[r - 1]) else "fixed_point"]["s_per_round_full"])
    variants["O2_1thread"] = {"value": rate, "unit": "iterations/s", "cores": 1, "s_per_round": spr, "by_regime": kinds, "sampled": not whole, "tree_build_s_once": tree_build_o2,
                              "sample": (f"all {E} edges" if whole else f"first {Ks} views ({len(keep)} of {E} edges), scaled x{scale:.2f}")}
    what = f"ALL {E} edges" if whole else f"the first {Ks} views ({len(keep)} of {E} edges, scaled x{scale:.2f} by edge count)"
    out["cpu_baseline"] = {
        "value": rate, "unit": "iterations/s", "cores": 1, "kind": "port", "sampled": not whole,
        "sample_short": (f"real nanoflann + oracle LM (restated Ceres), -O2, 1 thread, {what}, N={len(pb['pts'][0])}: one moving round (round {r_mov}: "
                         f"{kinds['moving']['s_per_round_full']:.1f} s) + one fixed-point round (round {r_fix}: {kinds['fixed_point']['s_per_round_full']:.1f} s) from the GPU run's poses, "
                         f"weighted over the timed window's rounds"),
        "sample": (f"reference-equivalent CPU path (real vendored nanoflann + oracle Jet/LM restatement of Ceres, pinned on README.md:141-146).  value = -O2, 1 thread (the "
                   f"reference's build and threading) on {what}, N={len(pb['pts'][0])}: one moving round (round {r_mov}) and one "
                   f"fixed-point round (round {r_fix}) started from the GPU run's poses, weighted over the rounds of the timed window.  "
                   f"variants.{'O3_avx2_allcores' if fast else 'O2_threadpool_nn'} = ALL {E} edges, rounds 1..{cpu_rounds} of the registration measured one by one on {ncores} cores"),
        "variants": variants, "host_cores": ncores,
        "fast_build_note": "-O3 -march=x86-64-v3 (AVX2/FMA, portable stand-in for -march=native: the .so is built off-box), -ffp-contract=off; all-cores = OpenMP over correspondences / queries",
    }
    return out


COMPACT_LIMIT = 4000   # bytes: the driver keeps an 8-KB tail of stdout+stderr, and the line has to sit inside it whole (round 5's 20-KB line did not: BENCH_r05.parsed = null)


def _sig(x, n=6):
    """numbers of the compact line: n significant digits (the detail file keeps full precision)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def compact_line(out, detail_file=None):
    """The ONE JSON line rank 0 prints last: the contract's keys + `roofline` + `cpu_baseline` in at most COMPACT_LIMIT bytes.  Everything else
    the run worked out (window arrays, regimes, per-kernel rooflines, the drop-in leg, the per-round CPU table) is in `detail_file`."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d}
    line = pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    line["config"] = pick(out.get("config", {}), ("workload", "views", "pts_per_view", "edges", "cutoff", "parallelism", "nn"))
    line.update(pick(out, ("value_moving_rounds", "value_fixed_point_rounds", "timed_region_s", "windows")))
    rf = out.get("roofline")
    line["roofline"] = None if not rf else pick(rf, ("kernel", "device_function", "bound", "achieved", "peak", "unit", "frac", "traffic", "valu_busy", "launches", "avg_us",
                                                     "alg_bytes_per_launch", "compulsory_frac"))
    others = {}
    for k, r in out.items():
        if k.startswith("roofline_") and r and k != "roofline_nn" and r.get("kernel") != (rf or {}).get("kernel"):
            others[r["kernel"]] = pick(r, ("frac", "avg_us", "launches", "traffic"))
    if others:
        line["roofline_other_kernels"] = others
    cb = out.get("cpu_baseline")
    if cb is not None:
        c = pick(cb, ("value", "unit", "cores", "kind", "sampled", "host_cores", "error"))
        if "sample" in cb:
            c["sample"] = cb.get("sample_short", cb["sample"][:300])
        allc = [v for k, v in cb.get("variants", {}).items() if "allcores" in k or "threadpool" in k]
        if allc:
            c["all_cores_value"] = allc[0]["value"]
        line["cpu_baseline"] = c
    if "speedup_vs_cpu_baseline" in out:
        line["speedup_vs_cpu_baseline"] = pick(out["speedup_vs_cpu_baseline"], ("value", "vs_all_cores"))
    if "pose_diff_vs_cpu_path" in out:
        line["pose_diff_vs_cpu_path"] = pick(out["pose_diff_vs_cpu_path"], ("rounds_compared", "max_translation_m", "max_rotation_rad"))
    if "pose_error_vs_gt" in out:
        line["pose_error_vs_gt"] = out["pose_error_vs_gt"]
    dp = out.get("dropin")
    if isinstance(dp, dict):
        line["dropin"] = {k: pick(dp[k], ("it_per_s_after_first_round", "first_round_ms")) for k in ("copyback", "device_only") if isinstance(dp.get(k), dict)} or pick(dp, ("error",))
        if "fixed_point_note" in dp:
            line["dropin"]["note"] = dp["fixed_point_note"]
    if "setup_s" in out:
        line["setup_s"] = out["setup_s"].get("total_s") if isinstance(out["setup_s"], dict) else out["setup_s"]
    line.update(pick(out, ("comm_ms_per_step", "rccl_nranks")))
    line["source_sha16"] = out.get("source_sha16")
    line["detail_file"] = detail_file
    line = _sig(line)
    s = json.dumps(line, separators=(",", ":"))
    for drop in ("roofline_other_kernels", "dropin", "pose_error_vs_gt", "speedup_vs_cpu_baseline"):   # (never needed at today's sizes; a guard, not a plan)
        if len(s) <= COMPACT_LIMIT:
            break
        line.pop(drop, None)
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= COMPACT_LIMIT, len(s)
    return s


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch_argv(n, argv, port=None):
    """The command `python bench.py --gpus N ...` turns itself into when it was not started by a launcher (RANK unset): the driver's own
    multi-GPU form, one rank per GPU on this node, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port if port is not None else free_port()), os.path.abspath(__file__)] + list(argv)


def synth_pose_diff(A, B):
    from mvicp import synth
    return synth.pose_diff(A, B)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=19)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("MVICP_WORKLOAD", "cfg4"))
    ap.add_argument("--nn", default="auto", choices=["auto", "brute", "grid", "tile"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rounds", type=int, default=None, help="rounds of the registration the CPU reference path walks after the timed region (default 8; 2 for cfg5-sized problems)")
    ap.add_argument("--no-replay", action="store_true", help="skip the untimed census replay pass (profiling runs: keeps the kernel trace to the timed protocol)")
    ap.add_argument("--allow-host-exchange", action="store_true", help="N > 1 only: if the RCCL communicator cannot be created, fall back to a host-staged gloo all-reduce instead of failing")
    ap.add_argument("--grid-target", type=float, default=None)
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (mvicp_set_option), repeatable; tuning / A-B runs")
    ap.add_argument("--windows", type=int, default=0, help="the timed window of --steps rounds is repeated this many times; value = the MEDIAN window (all in window_values).  "
                    "0 (default) = at least 5 and as many as make the whole timed region >= 2 s (sized from the first window)")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"), help="where rank 0 writes the full record (windows, regimes, every kernel's roofline, the drop-in "
                    "and CPU legs); the printed line stays compact and names this file")
    ap.add_argument("--no-dropin", action="store_true", help="skip the drop-in leg (bin/multiview on the same problem written to disk, with and without copy-back)")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.windows < 0:
        print("[bench] --gpus and --steps must be >= 1, --windows and --warmup >= 0", file=sys.stderr)
        sys.exit(2)

    import torch

    if args.gpus > 1 and "RANK" not in os.environ:
        # Started as plain `python bench.py --gpus N`: become N ranks (one per GPU) through torch.distributed.run instead of silently
        # measuring one.  The parent only launches and relays the exit code; rank 0 of the children prints the JSON line.
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            print(f"[bench] --gpus {args.gpus} needs {args.gpus} visible GPUs, this node has {n_dev}: not measuring a smaller job under that label", file=sys.stderr)
            sys.exit(2)
        import subprocess
        sys.exit(subprocess.call(self_launch_argv(args.gpus, sys.argv[1:])))

    import mvicp
    from mvicp import lib as L
    from mvicp import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MVICP_FORCE_DEVICE") is not None:   # debugging aid: several ranks on one GPU
        local = int(os.environ["MVICP_FORCE_DEVICE"])
    if world != args.gpus:
        # a launcher that started a different number of ranks than --gpus says is a mislabelled measurement: refuse
        if rank == 0:
            print(f"[bench] launched with WORLD_SIZE={world} but --gpus {args.gpus}: refusing to print a line labelled with either", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product path has no CPU fallback"
    if os.environ.get("MVICP_FORCE_DEVICE") is None and torch.cuda.device_count() < world:
        if rank == 0:
            print(f"[bench] {world} ranks but only {torch.cuda.device_count()} visible GPUs (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    K, N, plane, param, desc = WORKLOADS[args.workload]
    extra = dict(WORKLOAD_EXTRAS.get(args.workload, {}))
    cutoff = extra.pop("cutoff", 0.05)
    pb = synth.make_problem(K, N, **extra)
    eng = mvicp.Engine(local, rank, world)
    if args.grid_target:
        eng.set_option("grid_target", args.grid_target)
    for kv in args.opt:
        name, val = kv.split("=")
        eng.set_option(name, float(val))
    t_setup = time.perf_counter()
    eng.set_frames(pb["pts"], pb["nor"])
    t_frames = time.perf_counter()
    eng.set_graph(pb["src"], pb["dst"])
    eng.sync()
    setup_s = {"set_frames_s": t_frames - t_setup, "set_graph_s": time.perf_counter() - t_frames, "total_s": time.perf_counter() - t_setup,
               "note": "one-off per registration problem: upload + per-cloud structures (k-d order, box hierarchy, matrix-pipe operands, hash) in mvicp_set_frame; "
                       "buffers + the reference-equivalent trees (tie rule) in mvicp_set_graph.  The reference pays its lazy KD-tree build instead (cpu_baseline...tree_build_s_once)"}
    exchange = "none"
    if world > 1:
        rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        rccl = rccl if os.path.exists(rccl) else None
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(mvicp.Engine.comm_unique_id(rccl)), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        err = ""
        try:
            eng.comm_init(bytes(uid.cpu().numpy().tobytes()), rccl)
            ok = torch.ones(1, device="cuda")
        except Exception as ex:
            err = str(ex)
            ok = torch.zeros(1, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        exchange = "rccl all-reduce of per-edge blocks"
        if ok.item() == 0:
            if not args.allow_host_exchange:
                # the measured path must be the RCCL one: fail loudly instead of silently timing a host-staged exchange
                print(f"[bench] rank {rank}: RCCL communicator unavailable ({err or 'failed on another rank'}); pass --allow-host-exchange to time the "
                      f"host-staged gloo fallback instead", file=sys.stderr)
                eng.close()
                dist.destroy_process_group()
                sys.exit(3)
            gloo = dist.new_group(backend="gloo")

            def _host_allreduce(a):
                t = torch.from_numpy(a)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=gloo)

            eng.comm_set_callback(_host_allreduce)
            exchange = "host-staged gloo all-reduce (RCCL init failed; --allow-host-exchange)"
    method = {"auto": L.NN_AUTO, "brute": L.NN_BRUTE, "grid": L.NN_GRID, "tile": L.NN_TILE}[args.nn]

    # The timed loop keeps the poses in the engine's own K x 16 column-major buffer between the two calls of a round (what a C++ driver
    # does) and only records raw numbers; everything derived (dicts, pose comparisons, the (K,4,4) views) is worked out after the clock stops.
    rb = eng.round_state(K, pb["fixed"])
    Pc = rb["P"]                                   # in/out of mvicp_optimize
    init_c = L.poses_to_c(pb["init"])
    np.copyto(Pc, init_c)
    raw = []                                       # per global round: (g, t_nn, t_lm, iterations, evaluations, successful_steps, corr, poses after)
    state = {"g": 0}              # global round counter (warm-up + timed): round g % 20 + 1 of registration g // 20 + 1
    thresh32 = np.float32(cutoff)

    def step():
        g = state["g"]
        t0 = time.perf_counter()
        if g % ROUNDS_PER_REGISTRATION == 0 and g > 0:
            # a new run of the reference program on the same clouds: same noisy initial poses (default-seeded noise), no memory of the last run
            eng.reset_history()
            np.copyto(Pc, init_c)
        eng.correspond_raw(thresh32, method)
        t1 = time.perf_counter()
        sm = eng.optimize_raw(param, plane, True, 50)
        t2 = time.perf_counter()
        raw.append((g, t1 - t0, t2 - t1, sm.iterations, sm.evaluations, sm.successful_steps, int(rb["counts"].sum()), Pc.copy()))
        state["g"] = g + 1

    poses_after = {}              # registration 1 only: poses after round r (1-based), for the CPU-path comparison
    iters_of = {}

    def digest(rows):
        """raw rows -> the per-round log (after the clock stopped); also fills poses_after / iters_of and checks that later registrations
        retrace the first one bit for bit."""
        out = []
        for (g, t_nn, t_lm, its, evals, steps, corr, P16) in rows:
            reg, rnd = divmod(g, ROUNDS_PER_REGISTRATION)
            P = L.poses_from_c(P16)
            before = pb["init"] if rnd == 0 else prev_of.get((reg, rnd))
            out.append({"registration": reg + 1, "round": rnd + 1, "nn_ms": t_nn * 1e3, "lm_ms": t_lm * 1e3, "lm_iters": its, "evals": evals, "corr": corr,
                        "steps_taken": steps, "moved": steps > 0, "poses_bit_identical": bool(before is not None and np.array_equal(before, P))})
            prev_of[(reg, rnd + 1)] = P
            if reg == 0:
                poses_after[rnd + 1] = P; iters_of[rnd + 1] = its
            elif rnd + 1 in poses_after and not np.array_equal(poses_after[rnd + 1], P):
                state["replay_mismatch"] = True      # a later registration must retrace registration 1 bit for bit
        return out

    prev_of = {}

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    for _ in range(args.warmup):
        step()
    eng.profile(2)   # live HIP-event scopes around the roofline kernels only (one scope per NN kernel, "linearize") + the collective; everything else: replay pass below
    eng.profile_reset()
    # R back-to-back windows of exactly --steps rounds, each bracketed by barrier + synchronize on both sides and reduced with MAX over
    # ranks; `value` is the MEDIAN window (a 20-round window is ~20 ms: one window alone is at the mercy of a single slow launch)
    R = args.windows if args.windows > 0 else 5
    window_s, window_local_s = [], []
    w = 0
    while w < R:
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        window_local_s.append(dt)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        window_s.append(dt)
        if w == 0 and args.windows == 0:
            # size the timed region from the first window: >= 2 s in total (the driver's 5-second GPU-activity sampler then sees it), at most 400 windows;
            # every rank takes the same decision (dt is the MAX over ranks)
            R = int(min(400, max(5, np.ceil(2.0 / max(dt, 1e-6)))))
        w += 1
    NSTEPS = args.steps * R          # rounds under the live scopes (per-step averages below)
    med = int(np.argsort(window_s)[(R - 1) // 2])   # (lower median for an even count)
    elapsed = window_s[med]
    local_elapsed = window_local_s[med]
    digest(raw[:args.warmup])                       # warm-up rounds: only their poses matter (registration 1)
    log_all = digest(raw[args.warmup:])
    log = log_all[med * args.steps:(med + 1) * args.steps]   # the per-round log of the median window
    raw.clear()

    NN_SCOPES = ("nn_mfma", "nn_tile", "nn_grid", "nn_brute")   # one HIP-event scope per NN kernel; "nn_far" = the grid stage's second phase; "nn" = all together
    timed = {k: eng.profile_get(k) for k in ("nn", "linearize", "comm", "nn_far") + NN_SCOPES}
    spec_hits = eng.profile_get("spec.hit")[1] / R
    host = {k: eng.profile_get(k)[0] / NSTEPS for k in ("host.correspond", "host.corr.setup", "host.corr.nn_launch", "host.corr.post_launch", "host.corr.wait",
                                                          "host.corr.finish", "host.optimize", "host.evaluate")}
    eng.profile(False)
    timed_log = list(log)
    final_poses = L.poses_from_c(Pc)
    g_end = state["g"]

    # Replay pass (UNTIMED): the same global rounds again (same registrations, same resets), now with every profiling scope and the NN
    # census on (per-launch candidate / box / cache-hit counts = the algorithmic bytes of every NN launch).  The engine is
    # deterministic, so the replay walks through exactly the same poses and launches as the timed loop (checked below);
    # keeping the census kernels, their memsets and 20 extra event packets per round out of the timed region.
    replay = None
    census = None
    replay_identical = None
    if not args.no_replay:
        eng.reset_history()
        np.copyto(Pc, init_c)
        state["g"] = 0
        for _ in range(args.warmup):
            step()
        eng.profile(1)
        eng.set_option("nn_census", 1)
        eng.profile_reset()
        for _ in range(NSTEPS):
            step()
        fence()
        replay = {k: eng.profile_get(k) for k in ("nn", "compact", "gather", "select", "linearize", "reduce")}
        replay_ex = {k: eng.profile_get_ex(k) for k in ("nn",) + NN_SCOPES}
        census = eng.nn_census()
        eng.set_option("nn_census", 0)
        eng.profile(False)
        prev_of.clear(); digest(raw); raw.clear()
        replay_identical = bool(np.array_equal(L.poses_from_c(Pc), final_poses)) and not state.get("replay_mismatch", False)
    # registration 1 beyond the rounds the loop walked (untimed): the CPU-path comparison needs its first rounds
    cpu_rounds = args.cpu_rounds if args.cpu_rounds is not None else (8 if 2.0 * K * N <= 3e7 else 2)
    cpu_rounds = max(1, min(cpu_rounds, ROUNDS_PER_REGISTRATION))
    if world == 1 and not args.no_cpu_baseline and max(poses_after, default=0) < cpu_rounds:
        eng.reset_history()
        np.copyto(Pc, init_c)
        state["g"] = 0
        for _ in range(cpu_rounds):
            step()
        prev_of.clear(); digest(raw); raw.clear()
    log = timed_log
    window_rounds = [l["round"] for l in log]

    # ---- roofline.  Time and launch count: live HIP events in the timed region.  Algorithmic bytes:
    #   linearize  56 B (point-to-plane: p, n, n.q) / 48 B (point-to-point: p, q) per correspondence per evaluation  [library scope bytes]
    #   nn         per launch 24 N_src (queries) + 12 N_src (index + distance) + 24 x candidate points FETCHED FROM MEMORY + 8 x cells / boxes
    #              looked up, from the replay's exact census (hash-slot reads of the per-lane kernel are not counted: conservative).
    #              `compulsory` = the structure-independent bound 60 B per query the survey quotes next to it.
    # one entry per kernel: (timed ms, timed launches, SURVEY bytes over those launches, compulsory bytes, library-model bytes)
    prof = {"linearize": (timed["linearize"][0], timed["linearize"][1], timed["linearize"][2], None, None)}
    for k in ("nn",) + NN_SCOPES:
        ms_k, n_k, _ = timed[k]
        if n_k == 0:
            continue
        alg = comp = model = float("nan")
        # the replay pass repeats exactly the rounds under the live scopes (all R windows): same launches, now with the census on
        if replay is not None and replay_ex[k]["launches"] == n_k:
            alg = replay_ex[k]["survey_bytes"]
            comp = 60.0 * replay_ex[k]["queries"]
            model = replay_ex[k]["model_bytes"]
        prof[k] = (ms_k, n_k, alg, comp, model)

    sha = source_sha16()

    # FETCH_SIZE correction (profiles/r05_pmc_calibration.txt: tools/pmc_calib.hip under rocprofv3 --pmc, 1-GiB known-byte kernels in this library's own
    # access patterns): every coalesced read stream — 4, 8, 16 B per lane, 1-KiB tile fragments in random order — reads 0.500 of its bytes, and random 32-B
    # record gathers read 2.03x their USEFUL bytes = 0.5 of the 128-B lines they pull: the counter tallies each 128-B fabric request as 64 B in every
    # pattern, so HBM-side read traffic = 2 x FETCH_SIZE for every kernel; WRITE_SIZE reads 1.000 of 4 / 8 / 16-B coalesced stores
    FETCH_FACTOR_ALL = 2.0

    def pmc_traffic(name, what="traffic"):
        """HBM bytes per launch of THIS command from a committed rocprofv3 PMC summary (tools/profile.sh -> profiles/), used only if
        that summary was taken with the same workload / warm-up / steps / NN method AND the same device sources (hash) — otherwise null.
        FETCH_SIZE is doubled for every kernel: MI355X_MICROARCH.md §HBM prescribes it for wide coalesced streams on gfx950, and the calibration
        run in this library's own access patterns (profiles/r05_pmc_calibration.txt) found the same factor for every pattern the kernels use."""
        path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{args.workload}_w{args.warmup}s{args.steps}_kernels.json")
        if world != 1 or args.nn != "auto" or args.opt or not os.path.exists(path):
            return None
        try:
            j = json.load(open(path))
            if j.get("source_sha16") != sha or j.get("warmup_skipped") != args.warmup or j.get("timed_rounds") != args.steps:   # (per-launch averages: the number of windows does not matter)
                return None
            sc = j["scopes"][name]
            if what == "valu_busy":
                return sc.get("valu_busy")
            return (sc["FETCH_SIZE_KiB"] * FETCH_FACTOR_ALL + sc["WRITE_SIZE_KiB"]) * 1024.0
        except Exception:
            return None

    KERNEL_OF = {"nn_mfma": "nn_mfma_kernel", "nn_tile": "nn_tile_kernel", "nn_grid": "nn_grid_kernel", "nn_brute": "nn_brute_kernel", "linearize": "linearize_kernel",
                 "nn": "(all NN kernels together: a scope, not a kernel)"}

    def roof(name):
        if name not in prof:
            return None
        ms, n, b, comp, model = prof[name]
        if n == 0 or ms <= 0 or b is None or not np.isfinite(b):
            return None
        ach = (b / n) / (ms / n * 1e-3) / 1e9
        r = {"kernel": name, "device_function": KERNEL_OF.get(name), "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
             "traffic": pmc_traffic(name), "valu_busy": pmc_traffic(name, "valu_busy"), "launches": n,
             "traffic_note": "HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE of profiles/%s_%s_w%ds%d_kernels.json (same command, same device sources; factor 2: profiles/r05_pmc_calibration.txt); "
                             "valu_busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / kernel cycles from the same summary" % (PROFILE_ROUND, args.workload, args.warmup, args.steps), "avg_us": ms / n * 1e3, "total_ms": ms, "alg_bytes_per_launch": b / n}
        if name.startswith("nn"):
            r["compulsory_bytes_per_launch"] = comp / n
            r["compulsory_frac"] = comp / n / (ms / n * 1e-3) / 1e9 / HBM_PEAK_GBS
            r["overhead_bytes"] = max(0.0, (model - b) / n)   # temporal-cache / list state and record padding in the library's own finer byte model
            r["bytes_formula"] = "36 B/query + 24 B/candidate point fetched + 8 B/cell or box looked up (SURVEY.md §8d), census of the untimed replay pass"
        return r

    # the dominant KERNEL: largest total time under the live scopes among the individual kernels ("nn" is a scope over several kernels, never the headline)
    dominant = max([k for k in prof if k != "nn"], key=lambda k: prof[k][0])
    err_t = max(synth.pose_diff(final_poses[k], pb["gt"][k])[0] for k in range(K))
    err_r = max(synth.pose_diff(final_poses[k], pb["gt"][k])[1] for k in range(K))

    def regime(sel):
        rows = [l for l in log if sel(l)]
        if not rows:
            return {"rounds": 0}
        ms = float(np.mean([l["nn_ms"] + l["lm_ms"] for l in rows]))
        return {"rounds": len(rows), "ms_per_step": ms, "iterations_per_s": 1e3 / ms, "correspond_ms": float(np.mean([l["nn_ms"] for l in rows])),
                "optimize_ms": float(np.mean([l["lm_ms"] for l in rows])), "lm_iterations": float(np.mean([l["lm_iters"] for l in rows])),
                "device_evaluations": float(np.mean([l["evals"] for l in rows]))}

    per_rank = None
    if world > 1:
        # per-rank phase times (ms per step) so that a first real multi-GPU run is diagnosable: who waits for whom
        mine = torch.tensor([local_elapsed / args.steps * 1e3, float(np.mean([l["nn_ms"] for l in log])), float(np.mean([l["lm_ms"] for l in log])),
                             timed["nn"][0] / NSTEPS, timed["linearize"][0] / NSTEPS, timed["comm"][0] / NSTEPS, host["host.corr.wait"], host["host.evaluate"]],
                            dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        keys = ("ms_per_step", "correspond_ms", "optimize_ms", "nn_kernel_ms", "linearize_kernel_ms", "comm_ms", "host_corr_wait_ms", "host_evaluate_ms")
        per_rank = [{k: float(v) for k, v in zip(keys, t.cpu().tolist())} for t in allr]
    if rank == 0:
        rejected = sum(l["lm_iters"] - l["steps_taken"] for l in log)
        out = {
            "metric": "ICP iterations/sec (NN+Jacobian+LM)", "value": args.steps / elapsed, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "views": K, "pts_per_view": N, "edges": int(eng.E), "cutoff": cutoff, "knn": 2, "robust": True,
                       "parallelism": f"edge-sharded x{world}, {exchange}" if world > 1 else "single GPU", "nn": args.nn},
            "protocol": {"rounds_per_registration": ROUNDS_PER_REGISTRATION,
                         "registrations": [{"registration": int(r), "timed_rounds": [int(min(l["round"] for l in log if l["registration"] == r)),
                                                                                   int(max(l["round"] for l in log if l["registration"] == r))]}
                                           for r in sorted({l["registration"] for l in log})],
                         "start": "every registration starts from the same noisy initial poses (the reference's noise generator is default-seeded, common.h:36) "
                                  "after mvicp_reset_history() — inside the timed region when the boundary falls there",
                         "later_registrations_retrace_the_first_bit_for_bit": not state.get("replay_mismatch", False),
                         "note": "global round g (warm-up, then timed) = round g % 20 + 1 of registration g // 20 + 1 (main_multiview.cpp:150: 20 rounds); value = all timed "
                                 "rounds; `regimes` splits them by whether the LM solve took a step (moving) or ended without stepping (fixed point: the registration "
                                 "has converged and a round re-verifies it)"},
            "value_moving_rounds": regime(lambda l: l["moved"]).get("iterations_per_s"), "value_fixed_point_rounds": regime(lambda l: not l["moved"]).get("iterations_per_s"),
            "timed_region_s": float(sum(window_s)), "setup_s": setup_s,
            "regimes": {"moving_rounds": regime(lambda l: l["moved"]), "fixed_point_rounds": regime(lambda l: not l["moved"]),
                        "rounds_with_bit_identical_poses": int(sum(l["poses_bit_identical"] for l in log))},
            "round_ms": [round(l["nn_ms"] + l["lm_ms"], 4) for l in log], "round_index": window_rounds,
            "roofline": roof(dominant), "roofline_nn": roof("nn"), "roofline_linearize": roof("linearize"),
            **{"roofline_" + k: roof(k) for k in NN_SCOPES if k in prof},
            "window_values": [args.steps / t for t in window_s], "window_ms_per_step": [t / args.steps * 1e3 for t in window_s], "windows": R,
            "window_note": f"{R} back-to-back windows of {args.steps} rounds, each fenced; value / ms_per_step / round_ms / regimes = the median window; kernel_ms_per_step, host_ms_per_step and the rooflines = all windows",
            "phase_ms_per_step": {"correspond": float(np.mean([l["nn_ms"] for l in log])), "optimize": float(np.mean([l["lm_ms"] for l in log])),
                                  "lm_iterations": float(np.mean([l["lm_iters"] for l in log])), "device_evaluations": float(np.mean([l["evals"] for l in log])),
                                  "correspondences": float(np.mean([l["corr"] for l in log]))},
            "lm_step_economy": {"lm_iterations": int(sum(l["lm_iters"] for l in log)), "rejected_or_terminal_steps": int(rejected),
                                "evaluations": int(sum(l["evals"] for l in log)), "first_evaluations_served_by_the_queued_launch": int(spec_hits)},
            "kernel_ms_per_step": {**{k: timed[k][0] / NSTEPS for k in ("nn", "linearize", "nn_far") + NN_SCOPES if timed[k][1] > 0}},
            "host_ms_per_step": host,
            "pose_error_vs_gt": {"max_translation_m": err_t, "max_rotation_rad": err_r},
            "source_sha16": sha,
        }
        if replay is not None:
            out["kernel_ms_per_step"].update({k: replay[k][0] / NSTEPS for k in ("compact", "gather", "select", "reduce")})   # secondary scopes: replay pass
            out["replay_pass"] = {"identical_poses": replay_identical, "nn_ms_per_step": replay["nn"][0] / NSTEPS,
                                  "note": "untimed re-run of the same rounds with all scopes + NN census on (algorithmic bytes per launch)"}
            q = max(1.0, census["queries"])
            out["nn_census_per_query"] = {"candidates_examined": census["candidates"] / q, "candidate_points_fetched": census["fetched"] / q,
                                          "cells_or_boxes": census["nodes"] / q, "tree_fallback_fraction": census["far"] / q,
                                          "temporal_cache_hit_fraction": census["hits"] / q}
        if world > 1:
            out["comm_ms_per_step"] = timed["comm"][0] / NSTEPS
            out["comm_launches_per_step"] = timed["comm"][1] / NSTEPS
            out["rccl_nranks"] = eng.comm_nranks() if "rccl" in exchange else None
            out["per_rank"] = per_rank
        if world == 1 and not args.no_dropin and float(K) * N <= 8e6:   # (the leg writes the clouds to a temporary directory: not for the 64 x 1 M problem)
            # the literal drop-in route (bin/multiview: Frame / Session / ICP_Ceres mirror over the C ABI) on the same problem, after the clock stopped
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import dropin_bench
                eng.close()   # (one context at a time on the device: the driver builds its own)
                out["dropin"] = dropin_bench.run(pb, param, plane, cutoff, ROUNDS_PER_REGISTRATION)
            except Exception as ex:
                out["dropin"] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                moved_by_round = [bool(np.any(poses_after[r] != (pb["init"] if r == 1 else poses_after[r - 1]))) for r in sorted(poses_after)]
                legs = cpu_reference_legs(pb, plane, param, [poses_after[r] for r in range(1, cpu_rounds + 1)], [iters_of[r] for r in range(1, cpu_rounds + 1)],
                                          window_rounds, moved_by_round, cpu_rounds, cutoff)
                out.update(legs)
                out["speedup_vs_cpu_baseline"] = {"value": out["value"] / out["cpu_baseline"]["value"],
                                                  "vs_all_cores": out["value"] / [v for k, v in out["cpu_baseline"]["variants"].items() if "allcores" in k or "threadpool" in k][0]["value"],
                                                  "note": "GPU whole-job rate / CPU path rate over the same timed rounds (1 thread -O2: one moving + one fixed-point round on all edges up to cfg4's size; all cores: every compared round on all edges); a reported baseline, not a kernel-quality figure"}
            except Exception as ex:  # the baseline is a reported extra, never the measurement
                out["cpu_baseline"] = {"error": repr(ex)}
        detail = args.detail_file
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail)), exist_ok=True)
            with open(detail, "w") as f:
                json.dump(out, f)
                f.write("\n")
            detail = os.path.relpath(detail, ROOT)
        except OSError as ex:   # a read-only checkout still gets its line
            detail = None
            print(f"[bench] could not write {args.detail_file}: {ex}", file=sys.stderr)
        sys.stdout.flush()
        print(compact_line(out, detail), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
