#!/bin/bash
# Round 5 probe: the shape of a workgroup of the per-node sweeps (gradient, E update): 256 x 1, 64 x 4, 32 x 8 nodes (x, y)
cd $GRAFT_REPO_ROOT
build() { rm -f fusion-sim_amd/build/fes_api.o; make -C fusion-sim_amd EXTRA_HIPFLAGS="$1" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }; }
stat() { python3 - $1 "$2" <<'PY'
import csv,sys,glob
f=glob.glob('gpurun_out/%s/k_kernel_stats.csv'%sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('gradient_kernel','em_update_e')): print(sys.argv[2], r['Name'][:40], 'avg %.1f us'%(float(r['AverageNs'])/1e3))
PY
}
for BX in 256 64 32 128; do
  build "-DFES_NODE_BX=${BX}u"
  rm -rf gpurun_out/nbx; bash scripts/prof_bench.sh gpurun_out/nbx --only-c3 --c3-grid 512 --c3-particles 200000000 --steps 4 --warmup 1 --no-cpu-baseline; stat nbx "bx=$BX 512^3"
  rm -rf gpurun_out/nbx; bash scripts/prof_bench.sh gpurun_out/nbx --only-c3 --c3-particles 200000000 --steps 4 --warmup 1 --no-cpu-baseline; stat nbx "bx=$BX 256^3"
  rm -rf gpurun_out/nbx; bash scripts/prof_bench.sh gpurun_out/nbx --only-em --c3-particles 100000000 --steps 3 --warmup 1; stat nbx "bx=$BX em 256^3 fp64"
done
build ""
