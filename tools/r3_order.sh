#!/bin/bash
out=gpurun_out/r3_order.txt; : > $out
for o in 0 -1; do echo "=== SAT_TILE_ORDER=$o" >> $out; SAT_TILE_ORDER=$o python3 tools/shape_profile.py 2>&1 | grep -E "M640 |M13440|GEMM total" >> $out; done
for o in 0 -1; do echo "=== SAT_TILE_ORDER=$o" >> $out; SAT_TILE_ORDER=$o CFG=c2 N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager" >> $out; done
cat $out
