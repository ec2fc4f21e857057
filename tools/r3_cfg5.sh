#!/bin/bash
# dev: clean kernel-trace of the cfg5 (N=4096, width 512) layer step
export TMPDIR=/tmp
R=$PWD
for A in ${1:-lara}; do
OUT=$R/gpurun_out/cfg5_$A
rm -rf $OUT; mkdir -p $OUT
cd /tmp
python $R/bench.py --no-other-workloads --attn $A --workload cfg5 --steps 2 --warmup 2 --no-graph --no-cpu-baseline --gemm-tune-file $OUT/tune.csv > $OUT/tune.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --no-other-workloads --attn $A --workload cfg5 --steps 10 --warmup 2 --no-graph --no-cpu-baseline --gemm-tune-file $OUT/tune.csv > $OUT/trace.log 2>&1
cd $R
rm -f $OUT/trace/*/t_kernel_trace.csv $OUT/trace/t_kernel_trace.csv
python - $OUT <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/trace/**/t_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:28]:
    print("%-86s %5s %8.1f %5.1f%%"%(r["Name"][:86], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
tail -1 $OUT/trace.log | cut -c1-300
done
