#!/bin/bash
# SQ counter passes over tools/time_lara.py (dev tool): where do the waves spend their cycles?
ATTN=${1:-lara}
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/sq_$ATTN; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -o a -- python $R/tools/time_lara.py $ATTN 3 > $OUT/log1 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p2 -o b -- python $R/tools/time_lara.py $ATTN 3 > $OUT/log2 2>&1
cd $R; ls $OUT/p1 $OUT/p2
