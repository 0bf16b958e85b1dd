#!/bin/bash
# PMC passes over a bench.py command: each counter group in its own rocprofv3 run, kernel-trace only (no --sys-trace or
# other trace domains beside --pmc), as MI355X_MICROARCH.md prescribes; plus the kernel-trace/stats run of the same command.
#   scripts/pmc_run.sh <outdir> <bench.py args...>
set -e
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
CMD="python3 bench.py $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.json 2> $OUT/sq.err
find $OUT -name "*counter_collection.csv" -o -name "*kernel_stats.csv" | head
