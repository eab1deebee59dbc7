#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_train_step.py tests/test_gpu_graph.py -x -q 2>&1 | tail -3
for i in 1 2; do TF=none CFG=c2 N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager"; done
