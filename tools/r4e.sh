#!/bin/bash
for v in "" _nofin _nopoll; do
  echo "=== trace$v"
  F8NET_LIB=f8net_amd/libf8net_trace$v.so F8_TRACE_CHAIN=3 timeout 300 python tools/trace_run.py 2>&1 | grep -A3 -i "trace chain" | grep -v "wave [1235679]"
done
