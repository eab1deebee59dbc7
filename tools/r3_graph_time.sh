#!/bin/bash
out=gpurun_out/r3_graph_time2.txt; : > $out
for c in c1 c3 c2; do echo "--- $c SAT_WGRAD_STREAM=0" >> $out; SAT_WGRAD_STREAM=0 CFG=$c timeout -k 5 300 python3 tools/graph_step_time.py >> $out 2>&1 || echo "FAILED $c" >> $out; done
grep -v "amdgpu.ids" $out
