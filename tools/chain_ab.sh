#!/bin/bash
# On the GPU box: bench line + chain launch times for several library variants (tools/chain_variants.sh).  tools/chain_ab.sh default a b c
for v in "$@"; do
  LIB=f8net_amd/libf8net_$v.so; [ "$v" = default ] && LIB=f8net_amd/libf8net.so
  F8NET_LIB=$LIB timeout 300 python bench.py $BENCH_ARGS --steps 100 --warmup 10 --per-layer --no-cpu-baseline > /tmp/b_$v.json 2> /tmp/p_$v.txt
  python - <<PY
import json
d = json.load(open("/tmp/b_$v.json"))
print("== $v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
  grep -E "^ +[0-9]+ (stage|basic)_chain" /tmp/p_$v.txt | sed -E 's/^ +[0-9]+ ((stage|basic)_chain_x[0-9a-z_]+):[^ ]+ +([0-9.]+) us.*/     \1 \3 us/'
done
