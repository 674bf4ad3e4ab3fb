#!/bin/bash
# On the GPU box: per-launch time of the launches whose name matches $PAT, for several library variants.  tools/lib_ab.sh default nodpp ...
for v in "$@"; do
  LIB=f8net_amd/libf8net_$v.so; [ "$v" = default ] && LIB=f8net_amd/libf8net.so
  F8NET_LIB=$LIB timeout 300 python bench.py $BENCH_ARGS --steps 60 --warmup 10 --per-layer --no-cpu-baseline > /tmp/b_$v.json 2> /tmp/p_$v.txt
  echo "== $v $(python -c "import json; d=json.load(open('/tmp/b_$v.json')); print(d['value'], d['value_unpipelined'])")"
  grep -E "^ +[0-9]+ .*(${PAT:-stem})" /tmp/p_$v.txt | cut -c1-110
done
