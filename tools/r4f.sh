#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_nets.py -x -q -k "resnet50 or golden" 2>&1 | tail -3
for ft in 1 0 1; do
  F8_FUSE_TAIL=$ft timeout 300 python bench.py --steps 100 --warmup 10 --per-layer --no-cpu-baseline > /tmp/b.json 2> /tmp/p.txt
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== fuse_tail=$ft img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"], "alg_bytes_per_img", d["whole_net"]["alg_bytes_per_img"], "int", d.get("value_int_requant"))
PY
  grep -E "^ +[0-9]+ (stage_chain|fused_opener)" /tmp/p.txt | cut -c1-130
done
