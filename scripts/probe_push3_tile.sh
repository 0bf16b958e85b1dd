#!/bin/bash
# Round 5 probe: the electrostatic push with a smaller tile, so that TWO workgroups share a CU (one stages
# or flushes while the other computes), same box, same scenes:
#   A  16 x 16 x 8 cells, 1024 threads, one workgroup per CU (137.6 KB window)     (the default build)
#   B   8 x  8 x 8 cells,  512 threads, two workgroups per CU (13^3 nodes x 24 B = 52.7 KB each)
#   C  16 x  8 x 8 cells, 1024 threads, one workgroup per CU (85 KB: what the smaller tile alone costs)
#   D   8 x  8 x 8 cells, 1024 threads, one workgroup per CU  (LDS for three, waves for one)
# Rebuilds fes_api.o ON THE GPU BOX and restores the real build.  scripts/probe_push3_tile.sh > gpurun_out/r5_push3_tile.txt
cd $GRAFT_REPO_ROOT
c3() { python bench.py --only-c3 --c3-particles $2 --c3-grid $3 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['c3']; k=d['kernel_ms_per_substep']; print('$1  push %.3f ms (frac %.3f)  solve %.3f ms  sub-step %.3f ms  %s' % (k.get('push_gather_deposit', 0), d['roofline']['frac'], k.get('poisson_solve', 0), d['ms_per_substep'], {a: round(b, 3) for a, b in k.items()}))"; }
scenes() {
  c3 "$1 256^3 5e8   (29.8 per cell)" 500000000 256 &&
  c3 "$1 256^3 2.5e8 (14.9 per cell)" 250000000 256 &&
  c3 "$1 512^3 1e9   ( 7.45 per cell)" 1000000000 512
}
build() { rm -f fusion-sim_amd/build/fes_api.o; make -C fusion-sim_amd EXTRA_HIPFLAGS="$1" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }; }
for V in ${VARIANTS:-A B C D}; do
  case $V in
    A) build "" ;;
    B) build "-DFES_LTX=3 -DFES_LTY=3 -DFES_PUSH_THREADS=512" ;;
    C) build "-DFES_LTY=3" ;;
    D) build "-DFES_LTX=3 -DFES_LTY=3" ;;
  esac
  scenes $V || exit 1
done
build ""
