#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_train_step.py tests/test_gpu_graph.py tests/test_gpu_modules.py -x -q 2>&1 | tail -3
for s in 0 1 0 1; do echo "--- SAT_DEC_SIDE=$s"; SAT_DEC_SIDE=$s CFG=c2 N=20 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager"; done
for s in 0 1; do echo "--- c4 SAT_DEC_SIDE=$s"; SAT_DEC_SIDE=$s CFG=c4 N=10 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager"; done
for s in 0 1; do echo "--- c1 SAT_DEC_SIDE=$s"; SAT_DEC_SIDE=$s CFG=c1 N=20 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager\|graph"; done
