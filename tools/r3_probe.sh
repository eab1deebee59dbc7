#!/bin/bash
# Dev tool (round 3), run ON the GPU box from the repo root: baseline bench line, kernel trace of a few C2 steps -> critical-path table,
# per-shape GEMM table, host issue times of C1 / C3.
set -o pipefail
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/r3_base.json 2> $out/r3_base.err || exit 1
tail -c 400 $out/r3_base.json; echo
cd /tmp && rm -rf /tmp/tr && ROUNDS=1 PER=6 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o run -- python3 $root/tools/ab_step.py base > $out/r3_trace.log 2>&1 || exit 2
cd $root && python3 tools/critical_path.py /tmp/tr/run_kernel_trace.csv > $out/r3_critical_c2.txt 2>&1 || exit 3
head -5 $out/r3_critical_c2.txt
python3 tools/shape_profile.py > $out/r3_shapes.txt 2>&1 || exit 4
CFG=c1 python3 tools/cpu_issue_time.py > $out/r3_issue_c1.txt 2>&1 || exit 5
CFG=c3 python3 tools/cpu_issue_time.py > $out/r3_issue_c3.txt 2>&1 || exit 6
tail -1 $out/r3_issue_c1.txt; tail -1 $out/r3_issue_c3.txt
