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
rm -rf gpurun_out/prof_*
ls -la $OUT
