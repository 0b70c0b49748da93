"""Print the interesting fields of bench.py JSON lines: python tools/show_bench.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
    except (IndexError, OSError):
        print(f, "no JSON line"); continue
    d = json.loads(line)
    print(f"{f}: {d['value']:.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step")
    for k in ("kernel_ms_per_step", "host_ms_per_step", "phase_ms_per_step"):
        if k in d:
            print("   ", k.split("_")[0], {a: round(b, 3) for a, b in d[k].items()})
    for k in ("roofline_nn", "roofline_linearize"):
        if d.get(k):
            r = d[k]
            print("   ", k, f"{r['achieved']:.0f} GB/s frac {r['frac']:.3f} avg_us {r['avg_us']:.1f}")
