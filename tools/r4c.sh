#!/bin/bash
# round-4 iteration: chain parity (default lib + variant), A/B bench
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -3
for v in "$@"; do
  echo "== parity with $v"; F8NET_LIB=f8net_amd/libf8net_$v.so timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -k "stage_chain_matches" 2>&1 | tail -2
done
BENCH_ARGS="" bash tools/chain_ab.sh default "$@" default
