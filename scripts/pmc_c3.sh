#!/bin/bash
# PMC passes over the CART3D cycle (bench.py --only-c3): each counter group in its own rocprofv3 run,
# kernel-trace only, as the guide prescribes; plus the kernel-trace/stats run of the same command.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_c3
mkdir -p $OUT
CMD="python3 bench.py --only-c3 --steps 4 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.json 2> $OUT/sq.err
find $OUT -name "*counter_collection.csv" -o -name "*kernel_stats.csv" | head
