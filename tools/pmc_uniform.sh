#!/bin/bash
# SQ counters of the unclustered backward's launches on uniform points (two passes): bash tools/pmc_uniform.sh <out-dir>
OUT=${1:-gpurun_out/r05pmcu}; ROOT=$(pwd); mkdir -p $OUT; OUT=$(cd $OUT && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -- python $ROOT/tools/prof_hg_uniform.py 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p2 -- python $ROOT/tools/prof_hg_uniform.py 1 > /dev/null 2>&1
for k in hashgrid_bwd_owner hashgrid_bwd_aggregate sort_place gather_dy; do echo "== $k"; python $ROOT/tools/pmc_summary.py $k $OUT/p1 $OUT/p2; done > $OUT/pmc_uniform_summary.txt
cat $OUT/pmc_uniform_summary.txt
