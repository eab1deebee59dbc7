#!/bin/bash
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_loop_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3_loop_tests.log
for c in c3 c1 c2 c4; do for l in 0 1; do echo "--- $c SAT_BLOCK_LOOP=$l"; SAT_BLOCK_LOOP=$l CFG=$c N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager\|graph"; done; done
