#!/bin/bash
# After `gpurun -- 'PROFILE_MORE=1 bash tools/final_run.sh'`: copy what the judge reads from gpurun_out/ (scratch) into profiles/ (tracked).
R=${ROUND:-r06}
for t in cfg4_w5s20 cfg4_w1s19 cfg4_partial_w5s20 cfg5_w5s20; do
  cp gpurun_out/prof/${R}_$t/${R}_${t}_kernels.json gpurun_out/prof/${R}_$t/${R}_${t}_summary.txt profiles/ 2>/dev/null || echo "missing profile $t"
done
for w in cfg4_w5s20 cfg4 cfg2 cfg3 cfg5 shard8 shard8_cfg5 cfg4_partial; do
  cp gpurun_out/fin6_$w.json profiles/${R}_${w}_bench_line.json 2>/dev/null || echo "missing record $w"
  cp gpurun_out/fin6_$w.line profiles/${R}_${w}_bench_line.txt 2>/dev/null || echo "missing line $w"
done
cp gpurun_out/fin6_trace_auto.txt profiles/${R}_cfg4_round_trace.txt 2>/dev/null
cp gpurun_out/fin6_tests.txt profiles/${R}_gpu_tests.txt 2>/dev/null
python tools/update_design_tables.py > /dev/null && echo "DESIGN.md tables updated"
