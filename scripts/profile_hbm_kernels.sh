#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): counters of the path's HBM / latency-bound kernels at the C5 chunk: is
# composite_importance_kernel instruction-bound?  One counter group per pass (no trace domains combined with --pmc, as
# MI355X_MICROARCH.md's rocprofv3 section prescribes), plus a kernel-trace pass for the durations.  Outputs: gpurun_out/hbm_<tag>/.
TAG=${1:-r06}
OUT=$PWD/gpurun_out/hbm_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
CMD="python $REPO/scripts/diag/hbm_kernels_driver.py"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C -d $OUT/pmc_$N -o pmc --output-format csv -- $CMD > $OUT/pmc_$N.log 2>&1
done
cd $REPO
find $OUT -name "*.csv" -size +8M -delete
python $REPO/scripts/diag/hbm_kernels_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
