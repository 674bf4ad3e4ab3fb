#!/bin/bash
for v in trace_new trace_nofin; do
  echo "== $v"
  F8NET_LIB=/root/repo/f8net_amd/libf8net_$v.so F8_TRACE_CHAIN=30 timeout 300 python bench.py --arch resnet50 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep -A10 "trace chain<256" | head -11
done
