#!/bin/bash
# PMC passes (scripts/pmc_run.sh) of the three extension lines whose `roofline.traffic` bench.py quotes: the electrostatic push
# at configs[2], the full-EM push in double, configs[3] on one handle.  Outputs under gpurun_out/pmc_{c3,em,c4}; turned into
# profiles/rNN_{c3,em,c4}_traffic.json by scripts/pmc_kernel_traffic.py (commands in profiles/README.md).
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_c3x gpurun_out/pmc_emx gpurun_out/pmc_c4x
bash scripts/pmc_run.sh gpurun_out/pmc_c3x --only-c3 --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_c3x.log 2>&1 || { tail -5 gpurun_out/pmc_c3x.log; exit 1; }
bash scripts/pmc_run.sh gpurun_out/pmc_emx --only-em --em-precision fp64 --steps 4 --warmup 1 > gpurun_out/pmc_emx.log 2>&1 || { tail -5 gpurun_out/pmc_emx.log; exit 1; }
bash scripts/pmc_run.sh gpurun_out/pmc_c4x --workload box --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_c4x.log 2>&1 || { tail -5 gpurun_out/pmc_c4x.log; exit 1; }
echo done
