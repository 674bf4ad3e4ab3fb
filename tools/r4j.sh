#!/bin/bash
for v in 0 3 4 6 8 0; do
  F8_STEM_GRID_DIV=$v timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > /tmp/b.json 2> /dev/null
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== stem_grid_div=$v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
done
