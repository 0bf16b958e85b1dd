#!/bin/bash
# Everything profiles/rNN_* is made from, in one GPU call:  scripts/refresh_profiles.sh   (outputs under gpurun_out/final)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
F=gpurun_out/final; mkdir -p $F
echo "[1] bench.py"; python bench.py > $F/bench.json 2> $F/bench.err || { tail -5 $F/bench.err; exit 1; }
echo "[2] bench.py under rocprofv3 --stats"; bash scripts/prof_bench.sh $F/prof || exit 1
echo "[3] PMC passes of the headline"; bash scripts/pmc_bench.sh final > $F/pmc.log 2>&1 || { tail -5 $F/pmc.log; exit 1; }
echo "[4] configs[3]: one handle + 8 in-process ranks under rocprofv3"; bash scripts/prof_bench.sh $F/c4 --only-c4 || exit 1
python scripts/per_rank_kernels.py $(ls $F/c4/*/k_kernel_trace.csv $F/c4/k_kernel_trace.csv 2>/dev/null | head -1) > $F/c4_per_rank.txt 2>&1
echo "[5] configs[4] at full size on one GPU"; python bench.py --only-em --c3-grid 512 --c3-particles 2000000000 --em-precision fp64 --steps 3 --warmup 1 > $F/c5_one_gpu.json 2> $F/c5.err || { tail -5 $F/c5.err; exit 1; }
echo done; ls -R $F | head -40
