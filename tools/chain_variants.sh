#!/bin/bash
# Tuning: link variants of libf8net.so that differ only in f8_chain.hip's -D flags.  tools/chain_variants.sh tag "-DF8_CH_S2=4,3" ...
# -> f8net_amd/libf8net_<tag>.so (use with F8NET_LIB).  Needs an up-to-date build/ (f8net_amd/csrc/build.sh).
set -e
TAG=$1; shift
cd /root/repo/f8net_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage "$@" -c f8_chain.hip -o ../../build/f8_chain_$TAG.o 2> ../../build/f8_chain_$TAG.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage "$@" -c f8_bchain.hip -o ../../build/f8_bchain_$TAG.o 2> ../../build/f8_bchain_$TAG.log
grep -E "ScratchSize" ../../build/f8_bchain_$TAG.log | awk '{print $(NF-1)}' | tr '\n' ' '; echo
grep -E "ScratchSize" ../../build/f8_chain_$TAG.log | awk '{print $(NF-1)}' | tr '\n' ' '; echo
OBJS=$(ls ../../build/f8_*.o | grep -v "f8_chain\|f8_bchain" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../build/f8_chain_$TAG.o ../../build/f8_bchain_$TAG.o -o ../libf8net_$TAG.so
echo "built libf8net_$TAG.so"
