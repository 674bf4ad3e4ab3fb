#!/bin/bash
# Tuning build: the same sources with -DF8_TRACE (per-workgroup phase cycle counters) and any extra -D flags given as arguments,
# linked to f8net_amd/libf8net_trace.so.  Use with F8NET_LIB=f8net_amd/libf8net_trace.so and F8_TRACE_FUSED / F8_TRACE_OPENER /
# F8_TRACE_PATCH / F8_TRACE_LAUNCH = <k> (the k-th launch of that kernel family prints its averages).
set -e
cd "$(dirname "$0")/../f8net_amd/csrc"
OUT=../libf8net_trace.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DF8_TRACE $*"
mkdir -p ../../build/trace
for f in f8_kernels f8_fused f8_conv3x3 f8_stem f8_opener f8_ir f8_p12 f8_wreg f8_wstat f8_s2conv f8_fc f8_chain f8_cchain f8_bchain f8_dwmma f8_pool; do $HIPCC $FLAGS -c $f.hip -o ../../build/trace/$f.o & done
$HIPCC $FLAGS -x hip -c f8_net.cpp -o ../../build/trace/f8_net.o &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC ../../build/trace/*.o -o $OUT
echo "built $(readlink -f $OUT)"
