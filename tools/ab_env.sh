#!/bin/bash
# On the GPU box: bench value / value_unpipelined for several environment settings, interleaved REPS times.  tools/ab_env.sh "A=1" "A=0 B=2" ...
REPS=${REPS:-2}
for r in $(seq 1 $REPS); do
  for e in "$@"; do
    env $e timeout 300 python bench.py $BENCH_ARGS --steps ${STEPS:-600} --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['value_unpipelined'], d['whole_net']['sum_kernel_ms'])"
  done
done
