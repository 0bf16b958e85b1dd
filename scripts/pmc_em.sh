#!/bin/bash
# PMC passes over the full-EM push at configs[4]'s lattice (development): each counter group in its own rocprofv3 run, kernel
# trace only.  scripts/pmc_em.sh <tag> [precision] [particles] [grid]
export TMPDIR=/tmp
OUT=gpurun_out/pmc_em_$1
P=${2:-fp64}; N=${3:-1000000000}; G=${4:-512}
mkdir -p $OUT
CMD="python3 bench.py --only-em --c3-particles $N --c3-grid $G --em-precision $P --steps 2 --warmup 1"
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.json 2> $OUT/$name.err || { echo "pass $name failed"; tail -3 $OUT/$name.err; }; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT
pass tcc1 TCC_ATOMIC_sum TCC_EA_ATOMIC_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + '/*/*/*counter_collection.csv') + glob.glob(out + '/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    print(f)
    for k, d in agg.items():
        if 'em_push_tiles' in k or 'em_chain' in k or 'em_update_e' in k:
            print('  ', k, {c: '%.4g' % v for c, v in d.items()})
PY
