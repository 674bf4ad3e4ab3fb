#!/bin/bash
for v in 3 4 2 3; do
  F8_PIPELINE_DEPTH=$v timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > /tmp/b.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== depth=$v img/s", d["value"], "unpipelined", d["value_unpipelined"])
PY
done
