#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_modules.py tests/test_gpu_train_step.py tests/test_gpu_graph.py -x -q 2>&1 | tail -4
out=gpurun_out/r3_chunks.txt; : > $out
for c in c2 c1 c4 c3; do for k in 1 2 3 4; do echo "--- $c SAT_DEC_CHUNKS=$k" >> $out; SAT_DEC_CHUNKS=$k CFG=$c N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager" >> $out; done; done
cat $out
