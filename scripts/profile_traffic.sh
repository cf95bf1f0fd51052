#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): HBM traffic of the fine-pass MLP kernels of every precision path, as MI355X_MICROARCH.md's
# HBM section prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no trace domains combined with --pmc) -- plus a
# kernel-trace pass of the same driver for durations / register / scratch columns.  Outputs: gpurun_out/traffic_<tag>/.
TAG=${1:-r04}
OUT=$PWD/gpurun_out/traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
CMD="python $REPO/scripts/diag/traffic_driver.py"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc --output-format csv -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc --output-format csv -- $CMD > $OUT/pmc_write.log 2>&1
cd $REPO
find $OUT -name "*.csv" -size +8M -delete
ls -R $OUT | head -40
