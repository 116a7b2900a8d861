#!/bin/bash
# Profile the mixture / gc bubble-dew solvers (configs 4-5, 1e6 rows) with rocprofv3 on the GPU box.
#   scripts/profile_secondary.sh <tag> [mix|gc|all]   writes gpurun_out/prof_sec_<tag>/..., then
#   python scripts/summarise_secondary.py <tag>       (copies the judged summary into profiles/<tag>_secondary_pmc.md)
# Counters in their own passes, never combined with tracing other than kernel-trace.
set -e
TAG=${1:-r03}
WHAT=${2:-all}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_sec_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $REPO/scripts/dev/run_secondary.py $WHAT 3"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_mix -- $CMD > $OUT/pmc_mix.log 2>&1 || true
cd $REPO
du -sh $OUT
