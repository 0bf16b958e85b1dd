#!/bin/bash
# Round 5 probe: what the GPU's clocks and power do while the 3-D pushes run (the same build measures 10.07 and 11.17 ms for the
# double-precision EM push on two boxes of the pool).  Samples rocm-smi every 0.25 s beside bench.py --only-em / --only-c3.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5_clocks.txt; : > $OUT
sample() { while true; do echo "t=$(date +%s.%N) $(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E 'sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)' | tr -s ' ' | tr '\n' ';')" >> $OUT; sleep 0.25; done; }
sample & S=$!
echo "# idle" >> $OUT; sleep 2
echo "# bench.py --only-em fp64 256^3 5e8, 200 sub-steps" >> $OUT
python bench.py --only-em --em-precision fp64 --steps 100 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; print('# em push %.3f ms' % d['kernel_ms_per_substep']['push_gather_current'])" >> $OUT
echo "# bench.py --only-c3, 600 sub-steps" >> $OUT
python bench.py --only-c3 --steps 300 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['c3']; print('# c3 push %.3f ms' % d['kernel_ms_per_substep']['push_gather_deposit'])" >> $OUT
echo "# a long run: --only-em, 800 sub-steps" >> $OUT
python bench.py --only-em --em-precision fp64 --steps 400 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['em']; print('# em push %.3f ms' % d['kernel_ms_per_substep']['push_gather_current'])" >> $OUT
kill $S
grep -c sclk $OUT
