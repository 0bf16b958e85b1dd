#!/bin/bash
# Round 5 probe: the full-EM push as a persistent workgroup with a double-buffered window (FES_EM_PIPE=1,
# the build's default) against one item per workgroup with one window, same box, same scenes:
#   A  pipelined, double 8x4x8 tile / float 8x8x8, 768 threads                      (the default build)
#   B  one window, double 8x4x8 tile (61 KB) in TWO workgroups of 256 threads per CU  (what VERDICT r04 asked to be measured)
#   C  one window, double 8x8x8 tile (96 KB unpadded), one workgroup of 768           (round 4's shape with round 5's records)
#   D  FES_EM_PIPE=0 with its own defaults (double as C, float two workgroups of 256)
#   E  FES_EM_PIPE=2: persistent, 8x8x8 tile, the FIELD window double-buffered (the next item's records staged while this
#      item's particles are pushed), ONE accumulator window: double 2 x 64 + 32 KB, 768 threads; float 768 threads
# Rebuilds fes_api.o ON THE GPU BOX and restores the real build.  scripts/probe_em_pipe.sh > gpurun_out/r5_em_pipe.txt
cd $GRAFT_REPO_ROOT
em() { python bench.py --only-em --c3-particles $3 --c3-grid $4 --em-precision $2 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; p=d['kernel_ms_per_substep']['push_gather_current']; print('$1  push %.3f ms = %.1f ps per particle  lattice %.3f ms  sub-step %.3f ms' % (p, 1e9*p/$3, d['kernel_ms_per_substep']['fdtd_b_e_b'], d['ms_per_substep']))"; }
scenes() {
  em "$1 fp64 256^3 5e8   (29.8 per cell)" fp64 500000000 256 &&
  em "$1 fp64 256^3 2.5e8 (14.9 per cell)" fp64 250000000 256 &&
  em "$1 fp64 512^3 1e9   ( 7.45 per cell)" fp64 1000000000 512 &&
  em "$1 fp32 256^3 5e8   (29.8 per cell)" fp32 500000000 256 &&
  em "$1 fp32 512^3 1e9   ( 7.45 per cell)" fp32 1000000000 512
}
build() { rm -f fusion-sim_amd/build/fes_api.o; make -C fusion-sim_amd EXTRA_HIPFLAGS="$1" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }; }
for V in ${VARIANTS:-A B C}; do
  case $V in
    A) build "" ;;
    B) build "-DFES_EM_PIPE=0 -DFES_EM_THREADS_F64=256 -DFES_EM_THREADS_F32=256" ;;
    C) build "-DFES_EM_PIPE=0 -DFES_EM_LY_F64=3 -DFES_EM_THREADS_F64=768 -DFES_EM_THREADS_F32=768" ;;
    D) build "-DFES_EM_PIPE=0" ;;
    E) build "-DFES_EM_PIPE=2" ;;
  esac
  scenes $V || exit 1
done
build ""
