#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q -k "stride2" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_nets.py -x -q -k "resnet50 or golden" 2>&1 | tail -2
for v in default nofq default nofq; do
  LIB=f8net_amd/libf8net_$v.so; [ "$v" = default ] && LIB=f8net_amd/libf8net.so
  F8NET_LIB=$LIB timeout 300 python bench.py --steps 100 --warmup 10 --per-layer --no-cpu-baseline > /tmp/b.json 2> /tmp/p.txt
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== $v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"], "int", d.get("value_int_requant"))
PY
  grep -E "^ +[0-9]+ fused_opener" /tmp/p.txt | cut -c1-120
done
