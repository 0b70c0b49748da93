#!/bin/bash
# Round-end evidence in ONE gpurun call (GPU box):  gpurun --timeout 2700 -- 'bash tools/final_run.sh'
#   1. the GPU test suite — everything below is skipped unless it is green;
#   2. tools/profile.sh for both bench protocols (rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE / SQ passes) -> gpurun_out/prof/<tag>/: <tag>_kernels.json and
#      <tag>_summary.txt are copied into profiles/ (bench.py quotes `traffic` / `valu_busy` from them only while their source hash matches);
#   3. one bench run per workload: the printed (compact) line -> gpurun_out/fin6_<workload>.line, the full record (--detail-file) -> gpurun_out/fin6_<workload>.json;
#   4. the per-round NN trace of the AUTO method and the tile-kernel A/B of this round.
R=${ROUND:-r06}
mkdir -p gpurun_out
if [ -z "${SKIP_TESTS:-}" ]; then
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8 > gpurun_out/fin6_tests.txt
cat gpurun_out/fin6_tests.txt
if grep -q "failed\|Error" gpurun_out/fin6_tests.txt || ! grep -q "passed" gpurun_out/fin6_tests.txt; then echo TESTS_NOT_GREEN; exit 0; fi
fi
ROUND=$R bash tools/profile.sh cfg4 5 20 5 > gpurun_out/fin6_profile_w5s20.txt 2>&1
ROUND=$R bash tools/profile.sh cfg4 1 19 5 > gpurun_out/fin6_profile_w1s19.txt 2>&1
if [ -n "${PROFILE_MORE:-}" ]; then
ROUND=$R bash tools/profile.sh cfg4_partial 5 20 3 > gpurun_out/fin6_profile_partial.txt 2>&1
ROUND=$R bash tools/profile.sh cfg5 5 20 1 > gpurun_out/fin6_profile_cfg5.txt 2>&1
fi
# the bench lines below quote `traffic` from these summaries (same sources, same arguments): put them where bench.py looks
for t in cfg4_w5s20 cfg4_w1s19 cfg4_partial_w5s20 cfg5_w5s20; do cp gpurun_out/prof/${R}_$t/${R}_${t}_kernels.json gpurun_out/prof/${R}_$t/${R}_${t}_summary.txt profiles/ 2>/dev/null; done
timeout 600 python bench.py --warmup 5 --steps 20 --detail-file gpurun_out/fin6_cfg4_w5s20.json > gpurun_out/fin6_cfg4_w5s20.line 2> gpurun_out/fin6_cfg4_w5s20.err
timeout 600 python bench.py --detail-file gpurun_out/fin6_cfg4.json > gpurun_out/fin6_cfg4.line 2> gpurun_out/fin6_cfg4.err
for wl in cfg2 cfg3 shard8 shard8_cfg5 cfg4_partial; do timeout 300 python bench.py --workload $wl --warmup 5 --steps 20 --no-cpu-baseline --detail-file gpurun_out/fin6_$wl.json > gpurun_out/fin6_$wl.line 2> gpurun_out/fin6_$wl.err; done
timeout 600 python bench.py --workload cfg5 --warmup 5 --steps 20 --no-cpu-baseline --detail-file gpurun_out/fin6_cfg5.json > gpurun_out/fin6_cfg5.line 2> gpurun_out/fin6_cfg5.err
TRACE_CENSUS=1 TRACE_METHODS=auto timeout 60 python tools/round_trace.py 32 200000 10 2>&1 | grep -a nn_ms > gpurun_out/fin6_trace_auto.txt
AB_CENSUS=1 timeout 120 python tools/tile_ab.py 32 200000 7 "tile_miss=0" "" > gpurun_out/fin6_tile_ab.txt 2>&1
ls gpurun_out/fin6_*
