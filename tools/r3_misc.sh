#!/bin/bash
out=gpurun_out/r3_misc.txt; : > $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputests.log 2>&1; echo "pytest rc=$?" >> $out; tail -3 gpurun_out/r3_gputests.log >> $out
for f in 0 1; do echo "--- c2 SAT_ATT_FUSED=$f" >> $out; SAT_ATT_FUSED=$f CFG=c2 N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager" >> $out; done
echo "--- c1 default (side stream gated off)" >> $out; CFG=c1 N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager\|graph" >> $out
cat $out
