#!/bin/bash
# Tuning: a library variant that differs in ONE translation unit's -D flags: tools/variant.sh <tag> <unit, e.g. f8_conv3x3> -DFLAG=... -> f8net_amd/libf8net_<tag>.so
# (needs an up-to-date build/: f8net_amd/csrc/build.sh).  Use with F8NET_LIB.
set -e
TAG=$1; UNIT=$2; shift; shift
cd /root/repo/f8net_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $UNIT.hip -o /tmp/${UNIT}_$TAG.o
OBJS=$(ls ../../build/f8_*.o | grep -v "/$UNIT.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/${UNIT}_$TAG.o -o ../libf8net_$TAG.so
echo "built libf8net_$TAG.so"
