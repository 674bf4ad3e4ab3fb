#!/bin/bash
for arch in resnet50 resnet18; do
for v in 0 1 0 1; do
  F8_CHAIN_FILL=$v timeout 300 python bench.py --arch $arch --steps 200 --warmup 20 --no-cpu-baseline > /tmp/b.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== $arch chain_fill=$v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
done; done
