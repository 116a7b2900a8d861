#!/bin/bash
# Profile the headline benchmark with rocprofv3 on the GPU box.
#   scripts/profile.sh <tag>        writes gpurun_out/prof_<tag>/..., then summarise with
#   python scripts/summarise_profile.py <tag>   (copies the judged summaries into profiles/)
# The traced command is the driver's: `bench.py --steps 20 --warmup 5` (headline leg only, no CPU baseline).  Counters are
# collected in their own runs (never combined with tracing other than kernel-trace); the VALU issue costs the summary
# weighs the instruction classes with come from scripts/microbench/issue_cost.hip, run here as well.
set -e
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
PMCBENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra"
hipcc -O3 --offload-arch=gfx950 -o /tmp/issue_cost $REPO/scripts/microbench/issue_cost.hip 2>/dev/null
/tmp/issue_cost 4 > $OUT/issue_cost.jsonl
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMCBENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMCBENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $PMCBENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq2 -- $PMCBENCH > $OUT/pmc_sq2.log 2>&1 || true
# dynamic VALU instruction mix by class
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_mix -- $PMCBENCH > $OUT/pmc_mix.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU --output-format csv -d $OUT/pmc_mix2 -- $PMCBENCH > $OUT/pmc_mix2.log 2>&1 || true
cd $REPO
find $OUT -name "*.csv" | head -50
du -sh $OUT
