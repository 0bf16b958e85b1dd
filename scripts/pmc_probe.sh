#!/bin/bash
# PMC passes over scripts/probe_cycle.py (development): normal vs re-binning push launches.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_probe
mkdir -p $OUT
CMD="python3 scripts/probe_cycle.py --cycles 7 --sync-each 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc -- $CMD > $OUT/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1 || true
