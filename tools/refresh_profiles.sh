#!/bin/bash
# Run on the GPU box (via gpurun): every profile and bench line `profiles/README.md` lists, from the build in the tree.
#   gpurun --timeout 3000 -- 'bash tools/refresh_profiles.sh r02'
# then copy gpurun_out/refresh_<tag>/* into profiles/ (tools/refresh_profiles.sh does not write there itself: gpurun only merges
# gpurun_out/ back).
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/refresh_$TAG
mkdir -p $OUT
cd $REPO
MAIN=1 bash tools/profile.sh $TAG > $OUT/profile_main.log 2>&1
bash tools/profile.sh ${TAG}_r18 --arch resnet18 --bs 128 > $OUT/profile_r18.log 2>&1
bash tools/profile.sh ${TAG}_mbv2 --arch mobilenet_v2 --bs 128 > $OUT/profile_mbv2.log 2>&1
bash tools/profile.sh ${TAG}_r50bs256 --arch resnet50 --bs 256 > $OUT/profile_r50bs256.log 2>&1
for t in $TAG ${TAG}_r18 ${TAG}_mbv2 ${TAG}_r50bs256; do cp gpurun_out/prof_$t/summary/* $OUT/ 2>/dev/null; done
# the stamped counter files must be in place before the bench lines are taken (bench.py reads them)
cp $OUT/pmc_traffic_*.json $OUT/pmc_mfma_*.json $OUT/pmc_limiter_*.json profiles/ 2>/dev/null
timeout 900 python bench.py --steps 200 --warmup 20 --per-layer > $OUT/bench_$TAG.json 2> $OUT/perlayer_$TAG.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_${TAG}_driverflags.json 2> /dev/null
timeout 900 python bench.py --arch resnet18 --bs 128 --steps 200 --warmup 20 --per-layer > $OUT/bench_${TAG}_resnet18_bs128.json 2> $OUT/perlayer_${TAG}_resnet18_bs128.txt
timeout 900 python bench.py --arch mobilenet_v2 --bs 128 --steps 200 --warmup 20 --per-layer > $OUT/bench_${TAG}_mobilenet_v2_bs128.json 2> $OUT/perlayer_${TAG}_mobilenet_v2_bs128.txt
timeout 900 python bench.py --arch resnet50 --bs 256 --steps 200 --warmup 20 --per-layer > $OUT/bench_${TAG}_resnet50_bs256.json 2> $OUT/perlayer_${TAG}_resnet50_bs256.txt
# round 5: what bounds the chain launches — the P3 microbenchmark (vector throughput; 8 / 12 / 16 waves; ablations) and the per-wave phase cycle counters
# of a -DF8_TRACE build (tools/build_trace.sh) for the three ResNet-50 instances; rocprofv3 --att has no decoder library on this image (att_probe)
if [ -x tools/ubench/ubench_mfma_hazard.bin ]; then tools/ubench/ubench_mfma_hazard.bin > $OUT/ubench_mfma_hazard_$TAG.txt 2>&1; fi
if [ -x tools/ubench/ubench_p3.bin ]; then tools/ubench/ubench_p3.bin > $OUT/ubench_p3_$TAG.txt 2>&1; fi
if [ -f f8net_amd/libf8net_trace.so ]; then
  for k in 3 4 5; do F8NET_LIB=f8net_amd/libf8net_trace.so F8_TRACE_CHAIN=$k timeout 200 python tools/trace_run.py 2>&1 | grep -A12 "trace chain"; done > $OUT/chain_trace_$TAG.txt
  F8NET_LIB=f8net_amd/libf8net_trace.so F8_TRACE_CHAIN7=3 timeout 200 python tools/trace_chain7.py 2>&1 | grep "trace cchain" > $OUT/cchain_trace_$TAG.txt      # round 6: the 7x7 cluster chain's phases
fi
(cd /tmp && TMPDIR=/tmp timeout 120 rocprofv3 --att --kernel-trace -d /tmp/att_probe -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -3) > $OUT/att_probe_$TAG.log 2>&1
rm -rf gpurun_out/prof_*
ls -la $OUT
