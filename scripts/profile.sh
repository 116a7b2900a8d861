#!/bin/bash
# Profile the headline benchmark with rocprofv3 on the GPU box.
#   scripts/profile.sh <tag>        writes gpurun_out/prof_<tag>/..., then summarise with
#   python scripts/summarise_profile.py <tag>   (copies the judged summaries into profiles/)
# Counters are collected in their own runs (never combined with tracing other than kernel-trace).
set -e
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || true
cd $REPO
find $OUT -name "*.csv" | head -50
du -sh $OUT
