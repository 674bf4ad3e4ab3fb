#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_nets.py -x -q -k "resnet50" 2>&1 | tail -1
for v in new base new base new base; do
  lib=/root/repo/f8net_amd/libf8net.so; [ $v != new ] && lib=/root/repo/f8net_amd/libf8net_$v.so
  F8NET_LIB=$lib timeout 300 python bench.py --arch resnet50 --steps 200 --warmup 20 --per-layer --no-cpu-baseline > /tmp/b.json 2> /tmp/p.txt
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== $v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
  grep -E "stage_chain_x[34]" /tmp/p.txt | grep -E "^ +[0-9]+ " | cut -c1-110
done
