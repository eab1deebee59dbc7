#!/bin/bash
set -o pipefail
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/tr && ROUNDS=1 PER=6 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o run -- python3 $root/tools/ab_step.py base > $out/r3_trace.log 2>&1 || exit 2
cd $root && python3 tools/timeline_gaps.py /tmp/tr/run_kernel_trace.csv list > $out/r3_gaps_c2.txt 2>&1 || exit 3
python3 - <<'PY' > $out/r3_seq_c2.txt
import csv
rows=[]
for r in csv.DictReader(open('/tmp/tr/run_kernel_trace.csv')):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id","0")))
rows.sort()
# last full step: from the last-but-one optimizer_step to the last
opt=[i for i,r in enumerate(rows) if "optimizer_step" in r[2]]
a,b=opt[-2],opt[-1]
t0=rows[a][1]
import re
def short(n):
    m=re.search(r"gemm_glds_kernelILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)ELi(\d+)E(\w)",n)
    if m: return "glds<%s,%s,A%s,B%s,%s>"%(m.group(1),m.group(2),m.group(3),m.group(4),"bf" if m.group(5)=="D" else "f32")
    return re.sub(r"^_ZN3sat","",n)[:48]
prev_end={}
for s,e,n,q in rows[a:b+1]:
    gap=(s-prev_end.get(q,s))/1e3
    print("%9.1f q%s dur %7.1f gap %6.1f  %s"%((s-t0)/1e3,q,(e-s)/1e3,gap,short(n)))
    prev_end[q]=e
PY
echo done
