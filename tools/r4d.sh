#!/bin/bash
F8NET_LIB=f8net_amd/libf8net_trace.so F8_TRACE_CHAIN=3 timeout 300 python tools/trace_run.py 2>&1 | grep -A12 -i "trace chain"
BENCH_ARGS="" bash tools/chain_ab.sh default "$@" default
