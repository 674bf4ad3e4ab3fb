#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
#   tools/profile.sh <tag> [bench args, e.g. --arch mobilenet_v2 --bs 128]
# Output under gpurun_out/prof_<tag>/ ; summaries are post-processed by tools/summarize_prof.py (which also checks the stamp).
TAG=${1:-r03}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# serialise the in-flight runs: per-kernel durations then match bench.py's live (sequential, HIP-event) measurement
export F8_SPLIT_STREAMS=0
export F8_BENCH_LEAN=1      # the headline loop only: keeps the counter databases small
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline $*"
# the kernel-trace pass runs longer: a 7-pass run ends before the clocks have settled (round 4: its averages sat 7 % above bench.py's live HIP-event figure on the fast boxes)
TRACE_CMD="python $REPO/bench.py --steps 60 --warmup 10 --no-cpu-baseline $*"
echo "$*" > $OUT/bench_args.txt
python - > $OUT/csrc_sha256.txt <<PY
import sys; sys.path.insert(0, '$REPO')
import bench; print(bench.csrc_sha256())
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $TRACE_CMD > $OUT/trace.log 2>&1
# counters: their own runs, --pmc only (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass)
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o pmc -- $CMD > $OUT/pmc_mfma.log 2>&1
# what the waves do when neither roof binds (round 4): SQ has 8 counter slots per pass; the SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* family counts quad-cycles
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1
grep "^{\"metric\"" $OUT/trace.log | tail -1 > $OUT/bench_line.json      # (rocprofv3 prints after the program: the JSON line is not the last one)
# summarise on the box and drop the databases (gpurun merges at most 64 MiB back)
python $REPO/tools/summarize_prof.py $OUT $TAG --out $OUT/summary ${MAIN:+--main} > $OUT/summarize.log 2>&1 || tail -5 $OUT/summarize.log
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma $OUT/pmc_sq $OUT/pmc_sq2
ls $OUT $OUT/summary
