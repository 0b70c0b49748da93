"""Condenses rocprofv3 CSV output (tools/profile.sh) into small text summaries for profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]


def find(sub, pat):
    r = glob.glob(os.path.join(out_dir, sub, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    n = name
    for k in ("nn_grid_kernel", "nn_tile_kernel", "nn_brute_kernel", "linearize_kernel", "reduce_expand_kernel", "gather_kernel", "scatter_kernel",
              "count_kernel", "scan_kernel", "select_hist_kernel", "select_pick_kernel", "select_init_kernel", "census_sum_kernel", "nn_brute_merge_kernel"):
        if k in n:
            i = n.find(k) + len(k)
            extra = n[i:n.find(">", i) + 1] if n[i:i + 1] == "<" else ""
            return k + extra
    return n.split("(")[0][:60]


lines = []
tr = find("trace", "*kernel_trace.csv")
if tr:
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    with open(tr) as f:
        for r in csv.DictReader(f):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines.append(f"# rocprofv3 --kernel-trace --stats : {tag}")
    lines.append(f"{'kernel':48s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:48s} {a[0]:7d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100*a[1]/tot:6.2f}")
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    with open(cc) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != ctr:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    lines.append("")
    lines.append(f"# rocprofv3 --pmc {ctr} (KiB per dispatch, raw counter; gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream, MI355X_MICROARCH.md §HBM)")
    lines.append(f"{'kernel':48s} {'dispatches':>10s} {'avg_KiB':>14s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:48s} {a[0]:10d} {a[1]/a[0]:14.1f}")
for nm in ("bench_trace.json",):
    p = os.path.join(out_dir, nm)
    if os.path.exists(p):
        try:
            j = json.loads(open(p).read().strip().splitlines()[-1])
            lines.append("")
            lines.append("# bench line of the traced run (HIP-event figures measured live in bench.py)")
            lines.append(json.dumps({k: j[k] for k in ("value", "ms_per_step", "roofline", "roofline_nn", "roofline_linearize", "kernel_ms_per_step") if k in j}))
        except Exception as ex:
            lines.append(f"# bench line unreadable: {ex}")
# machine-readable per-kernel figures for bench.py's `traffic` field
kern = {}
if tr:
    with open(tr) as f:
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(f):
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for k, a in agg.items():
            kern.setdefault(k, {})["avg_us"] = a[1] / a[0]; kern[k]["calls"] = a[0]
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cc = find(sub, "*counter_collection.csv")
    if cc:
        agg = defaultdict(lambda: [0, 0.0])
        with open(cc) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == ctr:
                    a = agg[short(r["Kernel_Name"])]
                    a[0] += 1; a[1] += float(r["Counter_Value"])
        for k, a in agg.items():
            kern.setdefault(k, {})[ctr + "_KiB"] = a[1] / a[0]
json.dump({"tag": tag, "kernels": kern}, open(os.path.join(out_dir, f"{tag}_kernels.json"), "w"), indent=1, sort_keys=True)
txt = "\n".join(lines) + "\n"
open(os.path.join(out_dir, f"{tag}_summary.txt"), "w").write(txt)
print(txt)
