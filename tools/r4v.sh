#!/bin/bash
# upper bound of a planner-side stream bound: the stream's requantisation in the 3-operation form (exact while v + 2^(n-1) does not wrap)
F8NET_LIB=/root/repo/f8net_amd/libf8net_nowrap.so timeout 600 python -m pytest tests/test_gpu_nets.py -x -q -k "resnet50" 2>&1 | tail -2
for v in nowrap base nowrap base; do
  lib=/root/repo/f8net_amd/libf8net.so; [ $v != base ] && lib=/root/repo/f8net_amd/libf8net_$v.so
  F8NET_LIB=$lib timeout 300 python bench.py --arch resnet50 --steps 200 --warmup 20 --per-layer --no-cpu-baseline > /tmp/b.json 2> /tmp/p.txt
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== $v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
  grep -E "stage_chain" /tmp/p.txt | grep -E "^ +[0-9]+ " | cut -c1-110
done
