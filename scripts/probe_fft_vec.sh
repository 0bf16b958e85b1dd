#!/bin/bash
# Round 5 probe: the column passes of the Poisson solve with 16-byte (two complex floats) accesses per lane (FES_FFT_VEC=1, the
# build) against 8-byte ones (=0), one box: the solve per sub-step at 256^3 and 512^3, and the GPU solve tests on the new form.
cd $GRAFT_REPO_ROOT
c3() { python bench.py --only-c3 --c3-particles $2 --c3-grid $3 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['c3']; k=d['kernel_ms_per_substep']; print('$1  solve %.4f ms  push %.3f ms' % (k['poisson_solve'], k['push_gather_deposit']))"; }
build() { rm -f fusion-sim_amd/build/fes_api.o; make -C fusion-sim_amd EXTRA_HIPFLAGS="$1" all > gpurun_out/probe_build.log 2>&1 || { tail -5 gpurun_out/probe_build.log; exit 1; }; }
timeout -k 10 500 python -m pytest tests/test_gpu_es3d.py tests/test_gpu_full_size.py -x -q -m gpu -k "solve or poisson or decomposed or interface" 2>&1 | tail -2
for V in 1 0 1 0; do
  build "-DFES_FFT_VEC=$V"
  c3 "FES_FFT_VEC=$V 256^3 5e8" 500000000 256 && c3 "FES_FFT_VEC=$V 512^3 4e8" 400000000 512 || exit 1
done
build ""
