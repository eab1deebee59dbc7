#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_decoder.py tests/test_gpu_encoder.py -x -q 2>&1 | tail -3
python3 tools/shape_profile.py 2>&1 | grep -E "M640 |M13440|_f32|tn_|wgrad|GEMM total" | head -40
CFG=c2 N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager"
CFG=c4 N=10 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep "eager"
