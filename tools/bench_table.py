"""Markdown tables from bench lines (DESIGN.md's measurement section):  python tools/bench_table.py gpurun_out/fin6_*.json   (full records: bench.py --detail-file)"""
import json, sys

def load(f):
    try:
        return json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except (IndexError, OSError):
        return None

rows = []
for f in sys.argv[1:]:
    d = load(f)
    if d is None:
        continue
    rg = d["regimes"]; mv, fx = rg.get("moving_rounds", {}), rg.get("fixed_point_rounds", {})
    wv = d.get("window_values", [d["value"]])
    rows.append((d["config"]["workload"].split(":")[0], d["warmup"], d["steps"], d["value"], min(wv), max(wv), d["ms_per_step"], mv.get("rounds", 0), mv.get("ms_per_step", 0.0), fx.get("rounds", 0), fx.get("ms_per_step", 0.0), d))
print("| workload | warm-up + steps | it/s (median window; min … max of N windows) | ms/step | moving rounds: n × ms | fixed-point rounds: n × ms | set-up s |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r[0]} | {r[1]} + {r[2]} | **{r[3]:.1f}** ({r[4]:.0f} … {r[5]:.0f} of {r[-1].get('windows', 1)}) | {r[6]:.3f} | {r[7]} × {r[8]:.3f} | {r[9]} × {r[10]:.3f} | {r[-1].get('setup_s', {}).get('total_s', float('nan')):.2f} |")
print()
print("| workload | kernel | launches | avg µs | SURVEY-formula GB/launch | frac of 8 TB/s | 60 B/query frac | PMC traffic GB/launch (frac of 8 TB/s) | VALU busy |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    d = r[-1]
    for k in ("roofline_nn_mfma", "roofline_nn_tile", "roofline_nn_grid", "roofline_linearize"):
        q = d.get(k)
        if not q:
            continue
        tr = q.get("traffic")
        print(f"| {r[0]} ({r[1]}+{r[2]}) | {q['device_function']}{' **(dominant)**' if d.get('roofline') and d['roofline']['kernel'] == q['kernel'] else ''} | {q['launches']} | {q['avg_us']:.1f} | {q['alg_bytes_per_launch'] / 1e9:.3f} | {q['frac']:.3f} | "
              f"{q.get('compulsory_frac', float('nan')):.3f} | {('%.3f (%.2f)' % (tr / 1e9, tr / (q['avg_us'] * 1e-6) / 8e12)) if tr else '—'} | {('%.2f' % q['valu_busy']) if q.get('valu_busy') else '—'} |")
