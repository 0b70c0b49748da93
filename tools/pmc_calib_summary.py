"""profiles/r05_pmc_calibration.txt from the two rocprofv3 passes over tools/_build/pmc_calib (tools/pmc_calib.hip):
counter bytes / known bytes per access pattern = the factor a FETCH_SIZE / WRITE_SIZE figure of that pattern must be DIVIDED by."""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOWN = float(1 << 30)
out = []
for tag, ctr in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc", tag, "**", "p_counter_collection.csv"), recursive=True)
    if not f:
        out.append(f"{ctr}: no counter csv under gpurun_out/pmc/{tag}")
        continue
    acc = defaultdict(float); n = defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != ctr:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    out.append(f"## {ctr} (KiB as rocprofv3 reports it) vs the {KNOWN / 2**20:.0f} MiB every launch moves")
    for k in sorted(acc):
        per = acc[k] / max(1, len(n[k])) * 1024.0
        out.append(f"{k:36s} launches {len(n[k])}  {ctr} {per / 2**20:9.1f} MiB per launch   counter / known = {per / KNOWN:.3f}")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
