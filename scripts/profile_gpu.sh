#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace stats + PMC passes of the default bench command.
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
CMD=${PROFILE_CMD:-"python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"}   # PROFILE_CMD overrides the profiled command
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
# PMC passes, each on its own (no trace domains combined with --pmc)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc --output-format csv -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_lds -o pmc --output-format csv -- $CMD > $OUT/pmc_lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d $OUT/pmc_tcc -o pmc --output-format csv -- $CMD > $OUT/pmc_tcc.log 2>&1
cd $REPO
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small files
find $OUT -name "*.csv" -size +8M -delete
