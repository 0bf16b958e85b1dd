#!/bin/bash
# rocprofv3 kernel stats of a bench.py command: scripts/prof_bench.sh <outdir> <bench args...>
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
