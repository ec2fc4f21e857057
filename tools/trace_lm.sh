#!/bin/bash
# dev: per-dispatch kernel durations of one causal_eva LM layer step (run on the GPU box)
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o lm -- python $R/tools/debug_lm.py ${1:-18} train > /tmp/log.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/prof/**/lm_kernel_trace.csv",recursive=True)
if not f:
    print(open("/tmp/log.txt").read()[-2000:]); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
n=len(rows)//3
for r in rows[-n:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if (e-s) > 6000 or "ea::" in r["Kernel_Name"]: print("%-70s %8.1f us  gap %6.1f  grid %s lds %s" % (r["Kernel_Name"][:70], (e-s)/1e3, 0 if t0 is None else (s-t0)/1e3, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("LDS_Block_Size", "?")))
    t0=e
PY
