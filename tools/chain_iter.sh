#!/bin/bash
# Tuning loop for f8_chain.hip on the GPU box (via gpurun): parity tests of the chain launch, per-phase cycle counters of the three
# ResNet-50 instances (needs tools/build_trace.sh), and the bench line with the per-launch table.
#   gpurun --timeout 900 -- 'bash tools/chain_iter.sh <tag>'
TAG=${1:-it}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -2
F8NET_LIB=f8net_amd/libf8net_trace.so F8_TRACE_CHAIN=3 timeout 300 python tools/trace_run.py 2>&1 | grep -i "trace"
timeout 300 python bench.py --steps 200 --warmup 20 --per-layer --no-cpu-baseline > $OUT/bench.json 2> $OUT/perlayer.txt
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("img/s", d["value"], "unpipelined", d["value_unpipelined"], "sum_kernel_ms", d["whole_net"]["sum_kernel_ms"])
PY
grep -E "chain|opener|stem" $OUT/perlayer.txt | head -8
