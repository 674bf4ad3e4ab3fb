#!/bin/bash
# Tuning: trace builds (-DF8_TRACE) of the library that differ in f8_chain.hip's -D flags: tools/chain_trace_variants.sh tag flags...
set -e
TAG=$1; shift
cd /root/repo/f8net_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DF8_TRACE "$@" -c f8_chain.hip -o ../../build/trace/f8_chain_$TAG.oo
OBJS=$(ls ../../build/trace/f8_*.o | grep -v "f8_chain" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../build/trace/f8_chain_$TAG.oo -o ../libf8net_trace_$TAG.so
echo "built libf8net_trace_$TAG.so"
