#!/bin/bash
# Collects the rocprofv3 evidence for one bench command on the GPU box (run through gpurun):
#   pass 1: --kernel-trace --stats                (per-kernel time)
#   pass 2: --pmc FETCH_SIZE   (+ --kernel-trace) (TCC fetch bytes, own run: 3 of 4 TCC slots)
#   pass 3: --pmc WRITE_SIZE   (+ --kernel-trace)
#   pass 4: --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES (+ --kernel-trace)   (instruction issue: the NN kernels' real bound -> valu_busy)
# usage: bash tools/profile.sh <workload> <warmup> <steps> [windows]      ->  gpurun_out/prof/r06_<workload>_w<W>s<K>/
# tools/summarize_prof.py turns the CSVs into <tag>_summary.txt / <tag>_kernels.json (copy those into profiles/): per-kernel tables
# plus the two bench scopes restricted to the TIMED rounds, and the hash of the device sources (bench.py quotes `traffic` from the
# JSON only when workload, warm-up, steps and that hash all match its own run).
set -u
WL=${1:-cfg4}; W=${2:-1}; K=${3:-19}; R=${4:-5}
TAG=${ROUND:-r06}_${WL}_w${W}s${K}
CMD="python bench.py --workload $WL --warmup $W --steps $K --windows $R --no-cpu-baseline --no-replay --no-dropin --detail-file"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD $OUT/bench_trace.json > $OUT/bench_trace.line 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- $CMD $OUT/bench_fetch.json > $OUT/bench_fetch.line 2> $OUT/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- $CMD $OUT/bench_write.json > $OUT/bench_write.line 2> $OUT/write.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -o q -- $CMD $OUT/bench_sq.json > $OUT/bench_sq.line 2> $OUT/sq.err
python tools/summarize_prof.py $OUT $TAG $W $K $R
