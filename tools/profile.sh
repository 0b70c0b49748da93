#!/bin/bash
# Collects the rocprofv3 evidence for one bench command on the GPU box (run through gpurun):
#   pass 1: --kernel-trace --stats                (per-kernel time)
#   pass 2: --pmc FETCH_SIZE   (+ --kernel-trace) (TCC fetch bytes, own run: 3 of 4 TCC slots)
#   pass 3: --pmc WRITE_SIZE   (+ --kernel-trace)
# Outputs land in gpurun_out/prof/<tag>/ ; tools/summarize_prof.py turns them into profiles/<tag>_*.txt
set -u
TAG=${1:-r01_cfg4}
shift || true
CMD=${@:-python bench.py --workload cfg4 --steps 10 --warmup 1 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $CMD > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $CMD > $OUT/bench_write.json 2> $OUT/write.err
python tools/summarize_prof.py $OUT $TAG
find $OUT -name '*.csv' | head -20
