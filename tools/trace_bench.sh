#!/bin/bash
# dev: per-dispatch durations of the ea:: kernels of one bench step (run on the GPU box)
#   tools/trace_bench.sh <attn> [extra bench flags]
export TMPDIR=/tmp
R=$PWD
A=$1; shift
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o tb -- python $R/bench.py --attn $A "$@" --no-graph --no-cpu-baseline --no-gemm-tune --steps 2 --warmup 1 > /tmp/log.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/prof/**/tb_kernel_trace.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)//14
t0=None
for r in rows[-n:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if (e-s) > 5000 or "ea::" in r["Kernel_Name"]: print("%-78s %8.1f us gap %6.1f grid %s" % (r["Kernel_Name"][:78], (e-s)/1e3, 0 if t0 is None else (s-t0)/1e3, r.get("Grid_Size_X", r.get("Grid_Size"))))
    t0=e
PY
