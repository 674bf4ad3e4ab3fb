#!/bin/bash
for arch in mobilenet_v2 resnet18; do
for v in 3 4 3 4; do
  F8_PIPELINE_DEPTH=$v timeout 300 python bench.py --arch $arch --steps 200 --warmup 20 --no-cpu-baseline > /tmp/b.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== $arch depth=$v img/s", d["value"], "unpipelined", d["value_unpipelined"], "host", d.get("value_host_fed"))
PY
done; done
