#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_modules.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -5
for c in c4 c2; do for f in 0 1; do SAT_ATT_FAST=$f CFG=$c timeout -k 5 200 python3 tools/att_profile.py 2>&1 | grep -v amdgpu; done; done
