#!/bin/bash
# Round-end evidence in ONE gpurun call (GPU box):  gpurun --timeout 1500 -- 'bash tools/final_run.sh'
#   1. the GPU test suite — everything below is skipped unless it is green;
#   2. tools/profile.sh for both bench protocols (rocprofv3 kernel trace + the two PMC passes) -> gpurun_out/prof/<tag>/: copy <tag>_kernels.json and
#      <tag>_summary.txt into profiles/ afterwards (bench.py quotes `traffic` from them only while their source hash matches);
#   3. one bench line per workload -> gpurun_out/fin3_<workload>.json;
#   4. the per-round NN trace of the AUTO method.
R=${ROUND:-r04}
if [ -z "${SKIP_TESTS:-}" ]; then
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8 > gpurun_out/fin3_tests.txt
cat gpurun_out/fin3_tests.txt
if grep -q "failed\|Error" gpurun_out/fin3_tests.txt || ! grep -q "passed" gpurun_out/fin3_tests.txt; then echo TESTS_NOT_GREEN; exit 0; fi
fi
ROUND=$R bash tools/profile.sh cfg4 5 20 5
ROUND=$R bash tools/profile.sh cfg4 1 19 5
# the bench lines below quote `traffic` from these summaries (same sources, same arguments): put them where bench.py looks
for t in w5s20 w1s19; do cp gpurun_out/prof/${R}_cfg4_$t/${R}_cfg4_${t}_kernels.json gpurun_out/prof/${R}_cfg4_$t/${R}_cfg4_${t}_summary.txt profiles/ 2>/dev/null; done
timeout 400 python bench.py --warmup 5 --steps 20 > gpurun_out/fin3_cfg4_w5s20.json 2> gpurun_out/fin3_cfg4_w5s20.err
timeout 400 python bench.py > gpurun_out/fin3_cfg4.json 2> gpurun_out/fin3_cfg4.err
for wl in cfg2 cfg3 shard8 shard8_cfg5 cfg4_partial cfg5; do timeout 300 python bench.py --workload $wl --warmup 5 --steps 20 --no-cpu-baseline > gpurun_out/fin3_$wl.json 2> gpurun_out/fin3_$wl.err; done
TRACE_CENSUS=1 TRACE_METHODS=auto timeout 60 python tools/round_trace.py 32 200000 10 2>&1 | grep -a nn_ms > gpurun_out/fin3_trace_auto.txt
# the two tile kernels side by side: per-round NN stage + census, then the VALU / wave-cycle counters of their plain-round launches
AB_CENSUS=1 timeout 120 python tools/tile_ab.py 32 200000 7 "tile_mfma=0" "" "mfma_kacc=8" > gpurun_out/fin3_tile_ab.txt 2>&1
for v in 0 1; do TRACE_METHODS=auto bash tools/pmc_kernel.sh fin3_pmc_mfma$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" python tools/round_trace.py 32 200000 4 tile_mfma=$v | grep -a "nn_" > gpurun_out/fin3_pmc_mfma$v.txt; TRACE_METHODS=auto bash tools/pmc_kernel.sh fin3_pmc2_mfma$v "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" python tools/round_trace.py 32 200000 4 tile_mfma=$v | grep -a "nn_" >> gpurun_out/fin3_pmc_mfma$v.txt; done
ls gpurun_out/fin3_*
