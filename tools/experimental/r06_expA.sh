#!/bin/bash
# experiment A (round 6): bounds in every seeded round + cache prologue from round 3 on; per-round NN-stage ms and temporal-cache hit fractions
mkdir -p gpurun_out
for cen in 0 1; do
AB_CENSUS=$cen timeout 300 python tools/tile_ab.py 32 200000 8 "" "tile_bounds=2" "tile_bounds=2,tile_cache=2" "tile_bounds=2,tile_cache=2,tile_mfma=2" "tile_bounds=2,tile_cache=2,tile_mfma=2,mfma_lbt=2" > gpurun_out/expA_cfg4_cen$cen.txt 2>&1
AB_WORKLOAD=cfg4_partial AB_CENSUS=$cen timeout 300 python tools/tile_ab.py 32 200000 12 "" "tile_bounds=2" "tile_bounds=2,tile_cache=2" "tile_bounds=2,tile_cache=2,tile_mfma=2" > gpurun_out/expA_partial_cen$cen.txt 2>&1
done
cat gpurun_out/expA_*.txt | cut -c1-1500
