#!/bin/bash
out=gpurun_out/r3_side2.txt; : > $out
timeout -k 10 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_encoder.py tests/test_gpu_train_step.py -x -q > gpurun_out/r3_t.log 2>&1 || { tail -30 gpurun_out/r3_t.log; exit 1; }
tail -2 gpurun_out/r3_t.log
for c in c1 c3 c2; do for th in 0 1000000000; do echo "--- $c SAT_WGRAD_SIDE_MIN_PIXELS=$th" >> $out; SAT_WGRAD_SIDE_MIN_PIXELS=$th CFG=$c N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager\|graph" >> $out; done; done
cat $out
