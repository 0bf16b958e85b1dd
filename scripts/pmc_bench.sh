#!/bin/bash
# PMC passes over the bench command (development + profiles/): each counter group in
# its own rocprofv3 run, kernel-trace only, as the guide prescribes.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extensions --no-strong-c4"   # (the 512^3 block has no place under the serialising PMC passes of the headline)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.json 2> $OUT/tcc.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.json 2> $OUT/sq.err
find $OUT -name "*counter_collection.csv" | head
