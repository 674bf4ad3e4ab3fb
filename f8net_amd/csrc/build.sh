#!/bin/bash
# Builds libf8net.so (gfx950 only) in-tree: f8net_amd/libf8net.so
set -e
cd "$(dirname "$0")"
OUT=../libf8net.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage"
mkdir -p ../../build
$HIPCC $FLAGS -c f8_kernels.hip -o ../../build/f8_kernels.o 2> ../../build/f8_kernels.log &
$HIPCC $FLAGS -c f8_fused.hip -o ../../build/f8_fused.o 2> ../../build/f8_fused.log &
$HIPCC $FLAGS -c f8_conv3x3.hip -o ../../build/f8_conv3x3.o 2> ../../build/f8_conv3x3.log &
$HIPCC $FLAGS -c f8_stem.hip -o ../../build/f8_stem.o 2> ../../build/f8_stem.log &
$HIPCC $FLAGS -c f8_opener.hip -o ../../build/f8_opener.o 2> ../../build/f8_opener.log &
$HIPCC $FLAGS -c f8_ir.hip -o ../../build/f8_ir.o 2> ../../build/f8_ir.log &
$HIPCC $FLAGS -c f8_p12.hip -o ../../build/f8_p12.o 2> ../../build/f8_p12.log &
$HIPCC $FLAGS -c f8_wreg.hip -o ../../build/f8_wreg.o 2> ../../build/f8_wreg.log &
$HIPCC $FLAGS -c f8_wstat.hip -o ../../build/f8_wstat.o 2> ../../build/f8_wstat.log &
$HIPCC $FLAGS -c f8_s2conv.hip -o ../../build/f8_s2conv.o 2> ../../build/f8_s2conv.log &
$HIPCC $FLAGS -c f8_fc.hip -o ../../build/f8_fc.o 2> ../../build/f8_fc.log &
$HIPCC $FLAGS -c f8_chain.hip -o ../../build/f8_chain.o 2> ../../build/f8_chain.log &
$HIPCC $FLAGS -c f8_cchain.hip -o ../../build/f8_cchain.o 2> ../../build/f8_cchain.log &
$HIPCC $FLAGS -c f8_bchain.hip -o ../../build/f8_bchain.o 2> ../../build/f8_bchain.log &
$HIPCC $FLAGS -c f8_dwmma.hip -o ../../build/f8_dwmma.o 2> ../../build/f8_dwmma.log &
$HIPCC $FLAGS -c f8_pool.hip -o ../../build/f8_pool.o 2> ../../build/f8_pool.log &
$HIPCC $FLAGS -x hip -c f8_net.cpp -o ../../build/f8_net.o 2> ../../build/f8_net.log &
wait
for f in f8_kernels f8_fused f8_conv3x3 f8_stem f8_opener f8_ir f8_p12 f8_wreg f8_wstat f8_s2conv f8_fc f8_chain f8_cchain f8_bchain f8_dwmma f8_pool f8_net; do grep -E "error|warning:" ../../build/$f.log | grep -v Rpass || true; done
# a failed compile leaves the previous object in place: the link below would silently ship stale code
if grep -lE "(^|[^a-z])error( generated|:)" ../../build/f8_*.log; then echo "ERROR: compile errors (logs above)"; exit 1; fi
$HIPCC --offload-arch=gfx950 -shared -fPIC ../../build/f8_kernels.o ../../build/f8_fused.o ../../build/f8_conv3x3.o ../../build/f8_stem.o ../../build/f8_opener.o ../../build/f8_ir.o ../../build/f8_p12.o ../../build/f8_wreg.o ../../build/f8_wstat.o ../../build/f8_s2conv.o ../../build/f8_fc.o ../../build/f8_chain.o ../../build/f8_cchain.o ../../build/f8_bchain.o ../../build/f8_dwmma.o ../../build/f8_pool.o ../../build/f8_net.o -o $OUT
echo "built $(readlink -f $OUT)"
# device probe of the float requantisation (f8_device.h requant_u8x4): a stand-alone binary, run by tests/test_gpu_requant_probe.py
$HIPCC --offload-arch=gfx950 -O2 ../../tools/ubench/cvt_u8_probe.hip -o ../../tools/ubench/cvt_u8_probe.bin 2> ../../build/cvt_u8_probe.log || { echo "ERROR: cvt_u8_probe"; exit 1; }
# occupancy 1 = a kernel that spilled its accumulators into AGPRs on top of a full VGPR file (round 3: fused_ir_kernel ran like that)
grep -B8 "Occupancy \[waves/SIMD\]: 1" ../../build/f8_*.log 2>/dev/null | grep "Function Name" | sed -E "s/.*Function Name: ([^ ]+).*/NOTE: occupancy 1: \1/" | sort -u | head -20 || true
# a kernel whose host stub was silently dropped would only fail at dlopen time: catch it here
if nm -D $OUT | grep -q " U _ZN2f8"; then echo "ERROR: undefined f8:: symbols in $OUT"; nm -D $OUT | grep " U _ZN2f8" | head -5; exit 1; fi
# private arrays that end up in scratch memory (a dynamically indexed accumulator array costs 2x on the kernel): report them
python3 - <<'PY'
import re, glob
for log in sorted(glob.glob('../../build/f8_*.log')):
    name = None
    for line in open(log, errors='replace'):
        m = re.search(r'Function Name: (\S+)', line)
        if m: name = m.group(1)
        m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
        if m and int(m.group(1)) > 0 and name:
            print(f'WARNING: {name} uses {m.group(1)} bytes/lane of scratch')
PY
