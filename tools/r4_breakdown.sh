#!/bin/bash
# round 4 dev: rocprofv3 kernel stats of several workloads in one GPU call -> gpurun_out/r4bd/<name>.csv (top rows)
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r4bd; mkdir -p $OUT
cd /tmp
run() {  # name attn flags...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o t -- python $R/bench.py --no-other-workloads --no-graph --no-cpu-baseline --no-gemm-tune --steps 10 --warmup 3 "$@" > $OUT/$name.log 2>&1
  f=$(find /tmp/prof_$name -name 't_kernel_stats.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$OUT/$name.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in rows[:70]:
        w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
PY
}
for w in "$@"; do
  case $w in
    lara3) run lara3 --attn lara ;;
    lara2) run lara2 --attn lara --workload cfg2 ;;
    lara5b1) run lara5b1 --attn lara --workload cfg5 --batch 1 ;;
    lara5) run lara5 --attn lara --workload cfg5 ;;
    eva3) run eva3 --attn eva ;;
    eva2) run eva2 --attn eva --workload cfg2 ;;
    eva5) run eva5 --attn eva --workload cfg5 ;;
    sm5) run sm5 --attn softmax --workload cfg5 ;;
  esac
done
ls -la $OUT
