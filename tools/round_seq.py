"""Prints the kernel sequence of the LAST ICP round of a rocprofv3 --kernel-trace run (start offset, duration, gap to the previous kernel):
where a fixed-point round's time goes between its kernels.   python tools/round_seq.py <dir with *kernel_trace.csv> [rounds_back]"""
import csv, glob, os, sys
d = sys.argv[1]; back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("nn_grid_kernel", "nn_tile_kernel", "nn_mfma_kernel", "nn_brute_kernel<", "nn_brute_kernel("))]
a = heads[-back]; b = heads[-back + 1] if back > 1 else len(rows)
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
tot = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("mvicp::", "").split("(")[0][:48]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}")
    prev_end = e; tot += (e - s) / 1e3
print(f"round span {(prev_end - t0) / 1e3:.1f} us, kernel time {tot:.1f} us, {b - a} kernels")
