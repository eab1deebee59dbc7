#!/bin/bash
# round-3 final artifacts: full GPU test suite, profiles of C2 / C4 (kernel stats + PMC), default bench line, trace analysis, graph timing
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3_final_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3_final_tests.log
TAG=r3 bash tools/collect_profiles.sh c2 > gpurun_out/r3_collect_c2.log 2>&1 || { echo "collect c2 failed"; tail -5 gpurun_out/r3_collect_c2.log; }
TAG=r3 bash tools/collect_profiles.sh c4 > gpurun_out/r3_collect_c4.log 2>&1 || { echo "collect c4 failed"; tail -5 gpurun_out/r3_collect_c4.log; }
root=$(pwd)
cd /tmp && rm -rf /tmp/tr && ROUNDS=1 PER=6 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o run -- python3 $root/tools/ab_step.py base > $root/gpurun_out/r3_trace.log 2>&1
cd $root && python3 tools/critical_path.py /tmp/tr/run_kernel_trace.csv > gpurun_out/r3_critical_c2.txt 2>&1
for c in c1 c2 c3 c4 cli; do CFG=$c N=15 timeout -k 5 300 python3 tools/graph_step_time.py 2>&1 | grep -v amdgpu; done > gpurun_out/r3_graph_time_final.txt
python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo "bench rc=$?"
tail -c 300 gpurun_out/r3_bench_default.json
