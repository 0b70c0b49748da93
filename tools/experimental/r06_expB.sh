#!/bin/bash
# experiment B (round 6): nn_mfma_kernel seeded launches entered at the seeds' blocks + flat block sweep (mfma_entry = 1) vs the top-down walk;
# and the BND guard band (tile_mu) on slowly converging workloads (cache hit rates)
mkdir -p gpurun_out
timeout 300 python tools/tile_ab.py 32 200000 8 "" "mfma_entry=1" "" "mfma_entry=1" > gpurun_out/expB_cfg4.txt 2>&1
timeout 400 python tools/tile_ab.py 6 1000000 8 "" "mfma_entry=1" "" "mfma_entry=1" > gpurun_out/expB_6x1M.txt 2>&1
AB_WORKLOAD=cfg4_partial timeout 300 python tools/tile_ab.py 32 200000 8 "" "mfma_entry=1" > gpurun_out/expB_partial.txt 2>&1
AB_CENSUS=1 timeout 300 python tools/tile_ab.py 32 200000 6 "" "mfma_entry=1" > gpurun_out/expB_cfg4_cen.txt 2>&1
for cen in 0 1; do
AB_CENSUS=$cen timeout 500 python tools/tile_ab.py 6 1000000 12 "tile_mu=0.02" "tile_mu=0.05" "tile_mu=0.1" "tile_mu=0.2" "tile_mu=0.1,tile_mfma=2" > gpurun_out/expMU_6x1M_cen$cen.txt 2>&1
AB_WORKLOAD=cfg4_partial AB_CENSUS=$cen timeout 500 python tools/tile_ab.py 32 200000 20 "tile_mu=0.02" "tile_mu=0.05" "tile_mu=0.1" "tile_mu=0.2" "tile_mu=0.1,tile_mfma=2" > gpurun_out/expMU_partial_cen$cen.txt 2>&1
done
cut -c1-600 gpurun_out/expB_*.txt gpurun_out/expMU_*cen0.txt
