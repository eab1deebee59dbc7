#!/bin/bash
out=gpurun_out/r3_diag.txt; : > $out
for st in enc dec step; do
  for side in 0 1; do
    echo "=== $st side=$side" >> $out
    timeout -k 5 120 python3 tools/diag_capture.py $st side=$side >> $out 2>&1; echo "rc=$?" >> $out
  done
done
grep -E "===|rc=|captured|replayed|Error|error|Warning" $out | cut -c1-200 | head -60
