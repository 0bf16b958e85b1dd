#!/bin/bash
# Round 4 probe: do two 384-thread workgroups of the single-precision EM push really share a CU?  SQ_WAVE_CYCLES / SQ_BUSY_CYCLES
# under both builds.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for TH in 768 384; do
  rm -f fusion-sim_amd/build/fes_api.o
  make -C fusion-sim_amd EXTRA_HIPFLAGS="-DFES_EM_THREADS_F32=$TH" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }
  OUT=gpurun_out/pmc_occ_$TH; mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 bench.py --only-em --c3-particles 500000000 --c3-grid 256 --em-precision fp32 --steps 2 --warmup 1 > $OUT/out.json 2> $OUT/err.txt || tail -3 $OUT/err.txt
  python3 - $OUT $TH <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float); n = 0
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'em_push_tiles' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
print(sys.argv[2], 'threads:', {k: '%.4g' % v for k, v in agg.items()})
PY
done
rm -f fusion-sim_amd/build/fes_api.o
make -C fusion-sim_amd all > gpurun_out/probe_build.log 2>&1
