#!/bin/bash
for v in 2 1 2 1; do
  F8_SPLIT=$v timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > /tmp/b.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== split=$v img/s", d["value"], "unpipelined", d["value_unpipelined"], "intmodel", d.get("value_intmodel"), "host", d.get("value_host_fed"))
PY
done
