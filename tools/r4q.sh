#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_pipeline.py tests/test_onnx.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_switches.py -x -q -k "FUSE_POOL or full_resolution" 2>&1 | tail -2
for arch in resnet50 mobilenet_v2; do
for v in 1 0 1 0; do
  F8_FUSE_POOL=$v timeout 300 python bench.py --arch $arch --steps 200 --warmup 20 --per-layer --no-cpu-baseline > /tmp/b.json 2> /tmp/p.txt
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== $arch fuse_pool=$v img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
  grep -E "avgpool|tail.0|stage_3_layer_2.body.4|linear" /tmp/p.txt | grep -E "^ +[0-9]+ " | cut -c1-120
done; done
