#!/bin/bash
# dev: every kernel of ONE eager bench step in launch order with its duration (run on the GPU box)
#   tools/step_trace.sh <out-file> <bench flags ...>
export TMPDIR=/tmp
R=$PWD
OUT=$1; shift
rm -rf /tmp/prof_st; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_st -o tb -- python $R/bench.py "$@" --no-other-workloads --no-graph --no-cpu-baseline --no-gemm-tune --steps 3 --warmup 2 > $R/$OUT.log 2>&1
python - > $R/$OUT <<PY
import csv,glob
f=glob.glob("/tmp/prof_st/**/tb_kernel_trace.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last step: from the last noise / first projection back ... simply the last len/steps-ish rows: find the period by the
# last two launches of the first kernel name of the tail
while rows and any(t in rows[-1]["Kernel_Name"] for t in ("copy", "Copy", "fill", "Fill")):   # the bandwidth yardsticks after the steps
    rows.pop()
names=[r["Kernel_Name"] for r in rows]
last=names[-1]
idx=[i for i,n in enumerate(names) if n==last]
per=idx[-1]-idx[-2] if len(idx)>1 else len(rows)
tail=rows[-per:]
t0=None; tot=0
for r in tail:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-90s %8.1f us gap %6.1f" % (r["Kernel_Name"][:90], (e-s)/1e3, 0 if t0 is None else (s-t0)/1e3))
    tot+=(e-s)/1e3; t0=e
print("launches %d, kernel time %.1f us, span %.1f us" % (len(tail), tot, (int(tail[-1]["End_Timestamp"])-int(tail[0]["Start_Timestamp"]))/1e3))
PY
cd $R
