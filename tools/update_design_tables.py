"""Regenerates the two measurement tables of DESIGN.md section 8 (between the bench-table markers) from the full bench records gpurun_out/fin6_*.json (bench.py --detail-file)."""
import io, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = [os.path.join(ROOT, "gpurun_out", f"fin6_{w}.json") for w in ("cfg4_w5s20", "cfg4", "cfg2", "cfg3", "cfg5", "shard8", "shard8_cfg5", "cfg4_partial")]
out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "bench_table.py")] + files).decode()
t1, t2 = out.strip().split("\n\n")
keep = ("| cfg4 (5+20)", "| cfg5 (5+20)", "| cfg4_partial (5+20)")
rows = t2.split("\n")
t2 = "\n".join(rows[:2] + [r for r in rows[2:] if any(r.startswith(k) for k in keep)]).replace(" nan ", " — ")
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for tag, t in (("bench-table-1", t1), ("bench-table-2", t2)):
    a, b = s.index(f"<!-- {tag} -->"), s.index(f"<!-- /{tag} -->")
    s = s[:a] + f"<!-- {tag} -->\n" + t + "\n" + s[b:]
open(p, "w").write(s)
print(t1); print(t2)
