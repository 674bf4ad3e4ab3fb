#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q -k "stage_chain" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_nets.py -x -q -k "resnet50 or golden" 2>&1 | tail -2
for v in 1 0 1 0; do
  F8_CHAIN_R2=$v timeout 300 python bench.py --steps 100 --warmup 10 --per-layer --no-cpu-baseline > /tmp/b.json 2> /tmp/p.txt
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== chain_r2=$v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
  grep -E "^ +[0-9]+ (stage_chain)" /tmp/p.txt | cut -c1-120
done
