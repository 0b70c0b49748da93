"""Condenses rocprofv3 CSV output (tools/profile.sh) into small text/JSON summaries for profiles/.

Per-kernel table (calls, total, avg, min, max) from the kernel trace; per-kernel PMC averages; and the SCOPES that match
bench.py's HIP-event scopes: one per NN kernel ("nn_mfma", "nn_tile", "nn_grid", "nn_brute": that kernel's launches alone; "nn_far" =
nn_far_kernel + dirty_reduce_kernel of a grid stage), "nn" = every kernel of one mvicp_correspond NN stage, and "linearize" = one
linearize_kernel launch.  Scope figures cover bench.py's TIMED rounds only: the first `warmup` rounds (untimed warm-up steps) are
skipped and only the next `steps` x `windows` rounds are used, which also leaves out the untimed replay pass bench.py runs afterwards —
so they are per timed launch like bench.py's `roofline.achieved`.  The per-kernel table above the scopes covers every dispatch of the
process (warm-up and replay included)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 19
windows = int(sys.argv[5]) if len(sys.argv) > 5 else 1

KERNELS = ("nn_grid_kernel", "nn_far_kernel", "nn_tile_kernel", "nn_mfma_kernel", "nn_brute_kernel", "nn_brute_merge_kernel", "dirty_reduce_kernel", "census_sum_kernel",
           "linearize_kernel", "reduce_expand_kernel", "gather_kernel", "scatter_kernel", "count_kernel", "scan_kernel", "select_pass_kernel", "select_pick_kernel", "bracket_pass_kernel", "bracket_final_kernel",
           "select_final_kernel", "normals_kernel")
NN_SCOPE = ("nn_grid_kernel", "nn_far_kernel", "nn_tile_kernel", "nn_mfma_kernel", "nn_brute_kernel", "nn_brute_merge_kernel", "dirty_reduce_kernel", "census_sum_kernel")
NN_HEAD = ("nn_grid_kernel", "nn_tile_kernel", "nn_mfma_kernel", "nn_brute_kernel")  # first kernel of an NN stage
SCOPE_OF = {"nn_grid_kernel": "nn_grid", "nn_tile_kernel": "nn_tile", "nn_mfma_kernel": "nn_mfma", "nn_brute_kernel": "nn_brute", "nn_brute_merge_kernel": "nn_brute",
            "nn_far_kernel": "nn_far", "dirty_reduce_kernel": "nn_far"}   # bench.py's per-kernel HIP-event scopes


def find(sub, pat):
    r = glob.glob(os.path.join(out_dir, sub, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    for k in KERNELS:
        i = name.find(k)
        if i >= 0:
            j = i + len(k)
            extra = name[j:name.find(">", j) + 1] if name[j:j + 1] == "<" else ""
            return k + extra
    return name.split("(")[0][:60]


def base(name):
    return short(name).split("<")[0]


def scope_calls(rows, value_of):
    """rows: dispatches in Dispatch_Id order -> per-NN-stage sums of the timed rounds, and the linearize launches of those
    rounds (a round = one NN stage + the linearize launches up to the next NN stage)."""
    nn_calls, lin_rounds, per = [], [], []
    for r in rows:
        b = base(r["Kernel_Name"])
        if b in NN_HEAD:
            nn_calls.append(0.0); lin_rounds.append([]); per.append({})
        if b in NN_SCOPE and nn_calls:
            nn_calls[-1] += value_of(r)
            if b in SCOPE_OF and not (b == "dirty_reduce_kernel" and "nn_grid" not in per[-1]):   # (the tile kernels' dirty_reduce runs outside their scope)
                per[-1][SCOPE_OF[b]] = per[-1].get(SCOPE_OF[b], 0.0) + value_of(r)
        if b == "linearize_kernel" and lin_rounds:
            lin_rounds[-1].append(value_of(r))
    sel = slice(warmup, warmup + steps * windows)
    by_scope = defaultdict(list)
    for d in per[sel]:
        for k, v in d.items():
            by_scope[k].append(v)
    return nn_calls[sel], [v for rnd in lin_rounds[sel] for v in rnd], by_scope


lines = []
kern = {}
scopes = defaultdict(dict)
scopes["nn"]; scopes["linearize"]
tr = find("trace", "*kernel_trace.csv")
if tr:
    rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Dispatch_Id"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for r in rows:
        a = agg[short(r["Kernel_Name"])]
        d = dur(r)
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines.append(f"# rocprofv3 --kernel-trace --stats : {tag}")
    lines.append(f"{'kernel':48s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:48s} {a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100*a[1]/tot:6.2f}")
        kern.setdefault(k, {})["avg_us"] = a[1] / a[0]; kern[k]["calls"] = a[0]
    nn_calls, lin, by_scope = scope_calls(rows, dur)
    if nn_calls:
        t = nn_calls
        scopes["nn"].update(avg_us=sum(t) / len(t), calls=len(t))
    if lin:
        scopes["linearize"].update(avg_us=sum(lin) / len(lin), calls=len(lin))
    for k, t in by_scope.items():
        scopes[k].update(avg_us=sum(t) / len(t), calls=len(t), total_us=sum(t))
    lines.append("")
    lines.append(f"# scopes (bench.py HIP-event scopes; timed rounds only: {warmup} warm-up round(s) skipped, next {steps} x {windows} rounds used, replay pass excluded)")
    for k, v in sorted(scopes.items()):
        if v:
            lines.append(f"{k:12s} calls {v['calls']:5d}  avg_us {v['avg_us']:10.2f}" + (f"  total_us {v['total_us']:12.1f}" if "total_us" in v else ""))
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", "SQ_ACTIVE_INST_VALU"), ("pmc_sq", "SQ_INSTS_VALU"), ("pmc_sq", "SQ_WAVE_CYCLES")):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    if sub == "pmc_sq":
        # instruction-issue counters (own pass): per scope, and VALU busy = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / 1024 SIMDs / (avg_us x 2400 cycles per us)
        rows = sorted((r for r in csv.DictReader(open(cc)) if r.get("Counter_Name") == ctr), key=lambda r: int(r["Dispatch_Id"]))
        nn_calls, lin, by_scope = scope_calls(rows, lambda r: float(r["Counter_Value"]))
        if lin:
            scopes["linearize"][ctr] = sum(lin) / len(lin)
        for k, t in by_scope.items():
            scopes[k][ctr] = sum(t) / len(t)
        if ctr == "SQ_ACTIVE_INST_VALU":
            lines.append("")
            lines.append("# rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES (own pass): per timed launch; valu_busy = ACTIVE_INST_VALU x 4 / 1024 SIMDs / (avg_us x 2400)")
            for k, v in sorted(scopes.items()):
                if ctr in v and v.get("avg_us"):
                    v["valu_busy"] = v[ctr] * 4.0 / 1024.0 / (v["avg_us"] * 2400.0)
                    lines.append(f"{k:12s} SQ_ACTIVE_INST_VALU {v[ctr]:.4g}  avg_us {v['avg_us']:.1f}  valu_busy {v['valu_busy']:.3f}")
        continue
    rows = sorted((r for r in csv.DictReader(open(cc)) if r.get("Counter_Name") == ctr), key=lambda r: int(r["Dispatch_Id"]))
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    lines.append("")
    lines.append(f"# rocprofv3 --pmc {ctr} (KiB per dispatch, raw counter; gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream, MI355X_MICROARCH.md §HBM)")
    lines.append(f"{'kernel':48s} {'dispatches':>10s} {'avg_KiB':>14s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:48s} {a[0]:10d} {a[1]/a[0]:14.1f}")
        kern.setdefault(k, {})[ctr + "_KiB"] = a[1] / a[0]
    nn_calls, lin, by_scope = scope_calls(rows, lambda r: float(r["Counter_Value"]))
    if nn_calls:
        t = nn_calls
        scopes["nn"][ctr + "_KiB"] = sum(t) / len(t)
    if lin:
        scopes["linearize"][ctr + "_KiB"] = sum(lin) / len(lin)
    for k, t in by_scope.items():
        scopes[k][ctr + "_KiB"] = sum(t) / len(t)
p = os.path.join(out_dir, "bench_trace.json")
if os.path.exists(p):
    try:
        j = json.loads(open(p).read().strip().splitlines()[-1])
        lines.append("")
        lines.append("# bench line of the traced run (HIP-event figures measured live in bench.py)")
        lines.append(json.dumps({k: j[k] for k in ("value", "ms_per_step", "steps", "warmup", "windows", "window_values", "regimes", "roofline", "roofline_nn", "roofline_nn_mfma", "roofline_nn_tile", "roofline_nn_grid", "roofline_linearize", "kernel_ms_per_step") if k in j}))
    except Exception as ex:
        lines.append(f"# bench line unreadable: {ex}")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (source_sha16: the summary is only quoted by bench.py for the device sources it was measured on)
json.dump({"tag": tag, "warmup_skipped": warmup, "timed_rounds": steps, "windows": windows, "kernels": kern, "scopes": dict(scopes), "source_sha16": bench.source_sha16()},
          open(os.path.join(out_dir, f"{tag}_kernels.json"), "w"), indent=1, sort_keys=True)
txt = "\n".join(lines) + "\n"
open(os.path.join(out_dir, f"{tag}_summary.txt"), "w").write(txt)
print(txt)
