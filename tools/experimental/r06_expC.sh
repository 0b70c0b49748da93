#!/bin/bash
# experiment C (round 6): miss_block path of nn_tile_kernel in cache-aware rounds — tile_miss = 0 (off) / 4 / 8 / 16 / 32
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert\|^E " | tail -15 > gpurun_out/expC_tests.txt; cat gpurun_out/expC_tests.txt
timeout 300 python tools/tile_ab.py 32 200000 8 "tile_miss=0" "tile_miss=4" "tile_miss=8" "tile_miss=16" "tile_miss=32" "tile_miss=0" "tile_miss=8" > gpurun_out/expC_cfg4.txt 2>&1
AB_CENSUS=1 timeout 300 python tools/tile_ab.py 32 200000 8 "tile_miss=0" "tile_miss=8" > gpurun_out/expC_cfg4_cen.txt 2>&1
timeout 400 python tools/tile_ab.py 6 1000000 12 "tile_miss=0" "tile_miss=8" "tile_miss=16" "tile_miss=0" "tile_miss=8" > gpurun_out/expC_6x1M.txt 2>&1
AB_WORKLOAD=cfg4_partial timeout 300 python tools/tile_ab.py 32 200000 20 "tile_miss=0" "tile_miss=8" "tile_miss=16" > gpurun_out/expC_partial.txt 2>&1
cut -c1-700 gpurun_out/expC_cfg4.txt gpurun_out/expC_6x1M.txt gpurun_out/expC_partial.txt
