#!/bin/bash
# Builds libf8net.so (gfx950 only) in-tree: f8net_amd/libf8net.so
set -e
cd "$(dirname "$0")"
OUT=../libf8net.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p ../../build
$HIPCC $FLAGS -c f8_kernels.hip -o ../../build/f8_kernels.o &
$HIPCC $FLAGS -c f8_fused.hip -o ../../build/f8_fused.o &
$HIPCC $FLAGS -c f8_conv3x3.hip -o ../../build/f8_conv3x3.o &
$HIPCC $FLAGS -c f8_conv1x1.hip -o ../../build/f8_conv1x1.o &
$HIPCC $FLAGS -x hip -c f8_net.cpp -o ../../build/f8_net.o &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC ../../build/f8_kernels.o ../../build/f8_fused.o ../../build/f8_conv3x3.o ../../build/f8_conv1x1.o ../../build/f8_net.o -o $OUT
echo "built $(readlink -f $OUT)"
# a kernel whose host stub was silently dropped would only fail at dlopen time: catch it here
if nm -D $OUT | grep -q " U _ZN2f8"; then echo "ERROR: undefined f8:: symbols in $OUT"; nm -D $OUT | grep " U _ZN2f8" | head -5; exit 1; fi
