#!/bin/bash
# SQ / MFMA counter passes over tools/time_lara.py (the layer of bench.py's default workload, fwd+bwd):
# where do the waves spend their cycles, how busy are the matrix pipes, how much of the LDS time is bank conflicts.
# Counter passes only (no tracing options).  usage: tools/pmc_sq.sh <attn> [cfg3|cfg5]  ->  gpurun_out/sq_<attn>[_cfg5]/p{1,2,3}
ATTN=${1:-lara}; WL=${2:-cfg3}
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/sq_$ATTN; if [ "$WL" != "cfg3" ]; then OUT=$R/gpurun_out/sq_${ATTN}_$WL; fi; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/p1 -o a -- python $R/tools/time_lara.py $ATTN 3 $WL > $OUT/log1 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p2 -o b -- python $R/tools/time_lara.py $ATTN 3 $WL > $OUT/log2 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -o c -- python $R/tools/time_lara.py $ATTN 3 $WL > $OUT/log3 2>&1
cd $R
# keep the merged-back files small: this library's kernels only
for f in $OUT/p1/a_counter_collection.csv $OUT/p2/b_counter_collection.csv $OUT/p3/c_counter_collection.csv; do
  if [ -f $f ]; then (head -1 $f; grep "ea::" $f) > $f.tmp; mv $f.tmp $f; fi
done
ls $OUT/p1 $OUT/p2 $OUT/p3
