#!/bin/bash
# Dev tool, run ON the GPU box from the repo root: bench line + rocprofv3 kernel statistics + the two PMC passes (HBM read / write bytes)
# for one config.  usage: tools/collect_profiles.sh c2 [extra bench flags]   -> gpurun_out/prof_<cfg>/...
# (rocprofv3 needs the program itself after "--", and --pmc runs alone with --kernel-trace: see README "profiling")
set -o pipefail
cfg=${1:-c2}; shift
root=$(pwd); out=$root/gpurun_out/prof_$cfg${TAG:+_$TAG}; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
B="$root/bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-extras $*"
cd /tmp
python3 $B > "$out/bench_plain.json" 2> "$out/bench_plain.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 $B > "$out/bench_trace.json" 2> "$out/trace.log" || exit 2
cp "$out/trace/run_kernel_stats.csv" "$out/kernel_stats.csv"; rm -rf "$out/trace"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o run -- python3 $B > "$out/bench_fetch.json" 2> "$out/fetch.log" || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o run -- python3 $B > "$out/bench_write.json" 2> "$out/write.log" || exit 4
sf=$(python3 -c "import json,sys; d=json.loads([l for l in open('$out/bench_fetch.json') if l.startswith('{')][-1]); print(d['train_steps_in_process'])")
sw=$(python3 -c "import json,sys; d=json.loads([l for l in open('$out/bench_write.json') if l.startswith('{')][-1]); print(d['train_steps_in_process'])")
cd "$root" && python3 tools/pmc_aggregate.py "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_hbm.json" $sf $sw > "$out/pmc_aggregate.log" 2>&1 || exit 5
rm -rf "$out/pmc_fetch" "$out/pmc_write"
tail -2 "$out/pmc_aggregate.log"; tail -c 600 "$out/bench_plain.json"
