#!/bin/bash
# round 6, after making `miss` a run-time flag (one copy of the upper levels) and the tie trees lazy
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert\|^E " | tail -15 > gpurun_out/expD_tests.txt; cat gpurun_out/expD_tests.txt
timeout 300 python tools/tile_ab.py 32 200000 8 "tile_miss=0" "tile_miss=8" "tile_miss=0" "tile_miss=8" > gpurun_out/expD_cfg4.txt 2>&1
timeout 400 python tools/tile_ab.py 6 1000000 10 "tile_miss=0" "tile_miss=8" > gpurun_out/expD_6x1M.txt 2>&1
cut -c1-400 gpurun_out/expD_cfg4.txt gpurun_out/expD_6x1M.txt
for wl in cfg4 cfg5; do timeout 400 python bench.py --workload $wl --warmup 5 --steps 20 --no-cpu-baseline --no-dropin --detail-file gpurun_out/expD_$wl.json 2>/dev/null | cut -c1-300; python -c "
import json; d=json.load(open('gpurun_out/expD_$wl.json')); print('setup', d['setup_s']['set_frames_s'], d['setup_s']['set_graph_s'])"; done
