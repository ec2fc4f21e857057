#!/bin/bash
# round 6, GPU call 49: the whole-model lines whose layers now run as single nodes through GraphCore (fairseq encoder = LARA 'adaptive-1d'):
# eager + captured, and the DistributedDataParallel leg on a single-rank RCCL group
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --workload model_cfg5 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/m5.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('model_cfg5', d['value'], d['ms_per_step'], d.get('graph_ms_per_step'), d['config'].get('hipgraph'))" ; tail -c 300 gpurun_out/m5.err
EA_BENCH_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 python bench.py --workload model_cfg5 --steps 3 --warmup 2 --no-cpu-baseline 2> gpurun_out/m5d.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('model_cfg5 ddp', d['value'], d['ms_per_step'], (d.get('ddp') or d['config'].get('ddp') or {}) and 'ddp ok')"; tail -c 300 gpurun_out/m5d.err
EA_BENCH_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29582 python bench.py --attn lara --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads 2> gpurun_out/l5d.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 lara layer ddp path', d['ms_per_step'], d['config'].get('ddp_schedule'), d['config'].get('hipgraph'))"; tail -c 300 gpurun_out/l5d.err
