#!/bin/bash
# Round profiles on the GPU box (run from the repo root through gpurun): everything lands in gpurun_out/$TAG/ and is
# copied into profiles/ by hand afterwards.   bash tools/collect_profiles.sh r02
TAG=${1:-r02}
export NESVOR_COMMIT=${2:-unknown}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line + kernel stats of the same command
python $ROOT/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
# (training-step launches only, and the owner pass on the main stream: a kernel that shares the chip with others shows a
#  stretched duration in a trace; the bench line itself times the two launches the same way)
NESVOR_OWNER_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -- python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2>/dev/null
cp $(ls $OUT/kstats/*/*kernel_stats.csv | head -1) $OUT/bench_n1_kernel_stats.csv
# 2. hash-grid kernels: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and SQ counters
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python $ROOT/tools/prof_hashgrid.py P > /dev/null 2>&1
done
python $ROOT/tools/make_traffic_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_traffic.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- python $ROOT/tools/prof_hashgrid.py P > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- python $ROOT/tools/prof_hashgrid.py P > /dev/null 2>&1
for p in 1 2; do cat $OUT/pmc_sq$p/*/*counter_collection.csv > $OUT/pmc_sq_hashgrid_pass$p.csv; done
python $ROOT/tools/pmc_summary.py aggregate $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/pmc_sq_hashgrid_aggregate_summary.txt
python $ROOT/tools/pmc_summary.py owner $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/pmc_sq_hashgrid_owner_summary.txt
# 3. HBM bytes of a whole training step (all kernels of 10 steps, two passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/step_$c -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-extras > /dev/null 2>&1
done
python $ROOT/tools/step_traffic.py $OUT/step_FETCH_SIZE $OUT/step_WRITE_SIZE 32 > $OUT/step_traffic.json
# 4. micro-benchmarks
python $ROOT/tools/bench_hashgrid.py > $OUT/hashgrid_microbench.log 2>&1
python $ROOT/tools/bench_hg_levels.py > $OUT/hashgrid_per_level.log 2>&1
python $ROOT/tools/hg_variants.py r2: fixed32:-DNESVOR_FIXED32=1 noinsert:-DNESVOR_ABLATE=4 nowrite:-DNESVOR_ABLATE=8 noscan:-DNESVOR_ABLATE=2 > $OUT/hashgrid_ab.log 2>&1
rm -rf $OUT/kstats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/step_FETCH_SIZE $OUT/step_WRITE_SIZE
ls -la $OUT
