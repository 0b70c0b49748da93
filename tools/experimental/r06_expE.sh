#!/bin/bash
# round 6: "provably still rejected" as a temporal-cache hit (reject_cache) — partial-overlap workload, cfg4, cfg5's density
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert\|^E " | tail -15 > gpurun_out/expE_tests.txt; cat gpurun_out/expE_tests.txt
for cen in 0 1; do
AB_WORKLOAD=cfg4_partial AB_CENSUS=$cen timeout 400 python tools/tile_ab.py 32 200000 20 "reject_cache=0" "reject_cache=1" "reject_cache=1,tile_mfma=2" > gpurun_out/expE_partial_cen$cen.txt 2>&1
done
timeout 300 python tools/tile_ab.py 32 200000 8 "reject_cache=0" "reject_cache=1" > gpurun_out/expE_cfg4.txt 2>&1
timeout 400 python tools/tile_ab.py 6 1000000 10 "reject_cache=0" "reject_cache=1" > gpurun_out/expE_6x1M.txt 2>&1
cut -c1-420 gpurun_out/expE_partial_cen0.txt gpurun_out/expE_cfg4.txt gpurun_out/expE_6x1M.txt
timeout 300 python bench.py --workload cfg4_partial --warmup 5 --steps 20 --no-cpu-baseline --no-dropin --detail-file gpurun_out/expE_partial.json 2>/dev/null | cut -c1-300
