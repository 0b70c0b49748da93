#!/bin/bash
# round 6: neighbour-graph walk (nn_graph) — correctness digest + per-round NN-stage ms; census: 3rd tuple field / 64 = fraction of lanes finished by the walk
mkdir -p gpurun_out
timeout 400 python tools/tile_ab.py 32 200000 8 "" "nn_graph=1" "nn_graph=2" "nn_graph=1,graph_dist=1.0" > gpurun_out/expG_cfg4.txt 2>&1
AB_CENSUS=1 timeout 400 python tools/tile_ab.py 32 200000 8 "" "nn_graph=2" > gpurun_out/expG_cfg4_cen.txt 2>&1
cut -c1-700 gpurun_out/expG_cfg4.txt gpurun_out/expG_cfg4_cen.txt
