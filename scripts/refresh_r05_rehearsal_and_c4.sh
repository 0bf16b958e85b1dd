cd $GRAFT_REPO_ROOT
export FPIC_RCCL_LIBRARY=$GRAFT_REPO_ROOT/tests/fake_rccl/libfakerccl_shm.so
: > gpurun_out/r5_rehearsal_procs.txt
for N in 2 4; do
  echo "# python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2955$N bench.py --gpus $N --steps 4 --warmup 1 --workload box --bootstrap gloo --no-cpu-baseline --c4-grid 128 --c4-particles 3.2e7 --c4-ghost 4   (FPIC_RCCL_LIBRARY=tests/fake_rccl/libfakerccl_shm.so)" >> gpurun_out/r5_rehearsal_procs.txt
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2955$N bench.py --gpus $N --steps 4 --warmup 1 --workload box --bootstrap gloo --no-cpu-baseline --c4-grid 128 --c4-particles 3.2e7 --c4-ghost 4 >> gpurun_out/r5_rehearsal_procs.txt 2> gpurun_out/r5_rehearsal_procs_$N.err || echo "FAILED rc=$?" >> gpurun_out/r5_rehearsal_procs.txt
done
echo "# python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 --steps 5 --warmup 1 --bootstrap gloo --no-cpu-baseline --side 2000 --grid 512 --c4-grid 128 --c4-particles 3.2e7 --c4-ghost 4" >> gpurun_out/r5_rehearsal_procs.txt
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29559 bench.py --gpus 2 --steps 5 --warmup 1 --bootstrap gloo --no-cpu-baseline --side 2000 --grid 512 --c4-grid 128 --c4-particles 3.2e7 --c4-ghost 4 >> gpurun_out/r5_rehearsal_procs.txt 2> gpurun_out/r5_rehearsal_procs_d.err || echo "FAILED rc=$?" >> gpurun_out/r5_rehearsal_procs.txt
unset FPIC_RCCL_LIBRARY
cut -c1-200 gpurun_out/r5_rehearsal_procs.txt
rm -rf gpurun_out/final_c4; bash scripts/prof_bench.sh gpurun_out/final_c4 --only-c4; echo c4prof=$?
python scripts/per_rank_kernels.py $(ls gpurun_out/final_c4/*/k_kernel_trace.csv gpurun_out/final_c4/k_kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/r5_c4_per_rank.txt 2>&1; head -12 gpurun_out/r5_c4_per_rank.txt
