#!/bin/bash
# v_pk_mul_f32 in requant_u8x4 vs the commit before: parity, same-box A/B
timeout 300 python -m pytest tests/test_gpu_requant_probe.py -x -q 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_nets.py -x -q 2>&1 | tail -2
for arch in resnet50 mobilenet_v2 resnet18; do
for v in new base new base; do
  lib=/root/repo/f8net_amd/libf8net.so; [ $v != new ] && lib=/root/repo/f8net_amd/libf8net_$v.so
  F8NET_LIB=$lib timeout 300 python bench.py --arch $arch --steps 200 --warmup 20 --no-cpu-baseline > /tmp/b.json 2> /tmp/p.txt
  python - <<PY
import json
d = json.load(open("/tmp/b.json"))
print("== $arch $v img/s", d["value"], "unpipelined", d["value_unpipelined"], "int_requant", d.get("value_int_requant"), "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
done; done
