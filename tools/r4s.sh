#!/bin/bash
# upper bounds (results invalid): no waiting for the neighbours' rows at all
for v in grannp flagnp gran flag; do
  lib=/root/repo/f8net_amd/libf8net_$v.so; [ $v = gran ] && lib=/root/repo/f8net_amd/libf8net.so
  F8NET_LIB=$lib timeout 300 python bench.py --arch resnet50 --steps 200 --warmup 20 --per-layer --no-cpu-baseline > /tmp/b.json 2> /tmp/p.txt
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== $v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
  grep -E "stage_chain" /tmp/p.txt | grep -E "^ +[0-9]+ " | cut -c1-110
done
