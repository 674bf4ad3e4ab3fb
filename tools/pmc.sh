#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
# Each pass is its own rocprofv3 --pmc run of a short bench (counters only, no tracing).
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS}"
i=0
for C in "$@"; do
  timeout 600 rocprofv3 --pmc $C -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1 || tail -5 $OUT/p$i.log
  i=$((i+1))
done
python $REPO/tools/pmc_table.py $OUT
