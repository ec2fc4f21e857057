#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r3prof
EA_HIP_LIB=$R/tools/bin/libea_hip_prof.so python tools/time_lara.py lara 2 2> gpurun_out/r3prof/stamps_lara.txt >/dev/null
grep "lara_f mode 0" gpurun_out/r3prof/stamps_lara.txt | tail -2 | cut -c1-1500
grep -A1 "lara_f mode 0" gpurun_out/r3prof/stamps_lara.txt | grep blocks | tail -1
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_VALU_MFMA[A-Z_0-9]*\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*" | sort -u | head -30 > $R/gpurun_out/r3prof/mfma_counters.txt
cd $R
cat gpurun_out/r3prof/mfma_counters.txt
bash tools/pmc_sq.sh lara > /dev/null 2>&1
ls gpurun_out/sq_lara/p1 gpurun_out/sq_lara/p2 2>&1 | head
