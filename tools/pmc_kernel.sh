#!/bin/bash
# rocprofv3 --pmc pass over an arbitrary command (GPU box): per-kernel averages of ONE counter group per pass.
#   bash tools/pmc_kernel.sh <tag> "<counters>" <cmd...>
# Keep TCC counters in separate passes (FETCH_SIZE and WRITE_SIZE together do not fit the TCC slots: that pass never returned on
# this pool and burnt a 30-minute gpurun call in round 2) and always under a timeout.
set -u
TAG=$1; CTR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc/$TAG
mkdir -p $OUT
timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -o p -- "$@" > $OUT/cmd.out 2> $OUT/cmd.err
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/**/p_counter_collection.csv", recursive=True)
if not f:
    print("no counter csv"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], k)
    if key not in seen:
        seen.add(key); n[k] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in acc:
        line = f"{k:60s} disp {n[k]:4d} " + " ".join(f"{c}={v / n[k]:.4g}" for c, v in acc[k].items())
        print(line); fo.write(line + "\n")
PY
