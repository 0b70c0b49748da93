#!/bin/bash
# Round-end evidence in ONE gpurun call (GPU box):  gpurun --timeout 500 -- 'bash tools/final_run.sh'
#   1. the GPU test suite — everything below is skipped unless it is green;
#   2. tools/profile.sh for both bench protocols (rocprofv3 kernel trace + the two PMC passes) -> gpurun_out/prof/<tag>/: copy <tag>_kernels.json and
#      <tag>_summary.txt into profiles/ afterwards (bench.py quotes `traffic` from them only while their source hash matches);
#   3. one bench line per workload -> gpurun_out/fin2_<workload>.json (re-run the two cfg4 lines once the new profiles are in profiles/ if their
#      `traffic` field is wanted in the committed lines);
#   4. the per-round NN trace of the AUTO method.
timeout 330 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8 > gpurun_out/fin2_tests.txt
cat gpurun_out/fin2_tests.txt
if grep -q "failed\|Error" gpurun_out/fin2_tests.txt || ! grep -q "passed" gpurun_out/fin2_tests.txt; then echo TESTS_NOT_GREEN; exit 0; fi
bash tools/profile.sh cfg4 5 20
bash tools/profile.sh cfg4 1 19
timeout 120 python bench.py > gpurun_out/fin2_cfg4.json 2> gpurun_out/fin2_cfg4.err
timeout 120 python bench.py --warmup 5 --steps 20 > gpurun_out/fin2_cfg4_w5s20.json 2> gpurun_out/fin2_cfg4_w5s20.err
for wl in cfg2 cfg3 shard8 cfg5; do timeout 150 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/fin2_$wl.json 2> gpurun_out/fin2_$wl.err; done
timeout 60 python bench.py --workload shard8 --warmup 5 --steps 20 --no-cpu-baseline > gpurun_out/fin2_shard8_w5s20.json 2> gpurun_out/fin2_shard8_w5s20.err
TRACE_CENSUS=1 TRACE_METHODS=auto timeout 60 python tools/round_trace.py 32 200000 10 2>&1 | grep -a nn_ms > gpurun_out/fin2_trace_auto.txt
ls gpurun_out/fin2_*
