#!/bin/bash
# On the GPU box: headline (lean: the timed loop only) + per-launch table for several library variants, interleaved REPS times (boxes drift).
#   tools/ab.sh default nopeel ...      PAT=regex of launch names to print (default: chains), BENCH_ARGS, REPS (default 2)
for r in $(seq ${REPS:-2}); do
for v in "$@"; do
  LIB=f8net_amd/libf8net_$v.so; [ "$v" = default ] && LIB=f8net_amd/libf8net.so
  F8_BENCH_LEAN=1 F8NET_LIB=$LIB timeout 300 python bench.py $BENCH_ARGS --steps ${STEPS:-150} --warmup 20 --per-layer --no-cpu-baseline > /tmp/b_$v.json 2> /tmp/p_$v.txt
  echo "== $v rep $r: $(python -c "import json; d=json.load(open('/tmp/b_$v.json')); print(d['value'], 'sum_kernel_ms', d['whole_net']['sum_kernel_ms'])")"
  grep -E "^ +[0-9]+ .*(${PAT:-stage_chain|basic_chain})" /tmp/p_$v.txt | sed -E 's/^ +[0-9]+ ([^ ]+) +([0-9.]+) us.*/     \2 us  \1/' | cut -c1-90
done
done
