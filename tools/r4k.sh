#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q -k "basic_block" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_nets.py -x -q -k "resnet18 or golden" 2>&1 | tail -2
for v in default nw8 default nw8; do
  LIB=f8net_amd/libf8net_$v.so; [ "$v" = default ] && LIB=f8net_amd/libf8net.so
  F8NET_LIB=$LIB timeout 300 python bench.py --arch resnet18 --steps 100 --warmup 10 --per-layer --no-cpu-baseline > /tmp/b.json 2> /tmp/p.txt
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== $v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
  grep -E "^ +[0-9]+ basic_chain" /tmp/p.txt | cut -c1-120
done
