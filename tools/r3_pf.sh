#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_encoder.py -x -q 2>&1 | tail -3
python3 tools/shape_profile.py 2>&1 | grep -E "nn_128x128|conv_dgrad_128x128 M(131072|32768|8192) N(128|256|512) K(256|512|1024|128) |GEMM total" | head -30
for i in 1 2; do CFG=c2 N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager"; done
