#!/bin/bash
# round-3 profile collection: C2 and C4 kernel stats + PMC, register-staged vs direct-to-LDS convolution forward shapes (BatchNorm folding estimate)
set -o pipefail
export TMPDIR=/tmp
TAG=r3 bash tools/collect_profiles.sh c2 > gpurun_out/r3_collect_c2.log 2>&1 || { echo "collect c2 failed"; tail -5 gpurun_out/r3_collect_c2.log; }
TAG=r3 bash tools/collect_profiles.sh c4 > gpurun_out/r3_collect_c4.log 2>&1 || { echo "collect c4 failed"; tail -5 gpurun_out/r3_collect_c4.log; }
python3 tools/shape_profile.py > gpurun_out/r3_shapes_glds.txt 2>&1
SAT_NO_GLDS=1 python3 tools/shape_profile.py > gpurun_out/r3_shapes_regstaged.txt 2>&1
ls gpurun_out/prof_c2_r3 gpurun_out/prof_c4_r3
tail -3 gpurun_out/r3_collect_c2.log
