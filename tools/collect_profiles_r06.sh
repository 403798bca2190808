#!/bin/bash
# Round-6 profiles on the GPU box (run from the repo root through gpurun): everything lands in gpurun_out/r06prof/ and is
# copied into profiles/ afterwards.   bash tools/collect_profiles_r06.sh <commit>
# New against round 4 (verdict item 3d): the kernel trace of the bench runs with the record queues of the hash-grid backward at the
# capacities the untraced run before it settled on (NESVOR_HASHGRID_QUEUE_SAVE / NESVOR_HASHGRID_QUEUE=load:<file>), so no launch
# of the trace takes the overflow fallback that the adaptive sizing goes through in its first iterations (one 10 ms aggregation
# launch among 300-us ones bent round 4's rocprofv3 AVERAGE to 0.60 where medians and events said 0.74) - and the launches
# traced are the PRODUCT's (default streams, fused AdamW, default queue capacities).
# (Tried first: `rocprofv3 --selected-regions` with roctxProfilerResume / Pause from bench.py - the trace came out empty; worst-case
#  queues (NESVOR_HASHGRID_QUEUE=worst) - no settling launches, but the owner pass walks more slices: 113 us against 82.)
TAG=r06
export NESVOR_COMMIT=${1:-unknown}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r06prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the bench line (driver's invocation: 20 steps) and the 200-step default + kernel stats of the same command under rocprofv3
python $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_n1_driver_invocation.json 2> $OUT/bench_n1.err
NESVOR_HASHGRID_QUEUE_SAVE=/tmp/nesvor_queue_scales.json python $ROOT/bench.py > $OUT/bench_n1.json 2>> $OUT/bench_n1.err
NESVOR_HASHGRID_QUEUE=load:/tmp/nesvor_queue_scales.json rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -- python $ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-strict --small-batches "" > $OUT/bench_under_rocprof.json 2>/dev/null
python $ROOT/tools/kernel_stats_settled.py $OUT/kstats > $OUT/bench_n1_kernel_stats.csv
cp $(ls $OUT/kstats/*/*kernel_stats.csv | head -1) $OUT/bench_n1_kernel_stats_rocprof_avg.csv
# 2. kernel timeline of one training step (default streams: owner pass on the side stream) and at 512 pixels (2^17 points)
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-strict --no-kernel-timing --small-batches "" > /dev/null 2>&1
python $ROOT/tools/step_timeline.py $OUT/tl step_prologue > $OUT/step_timeline.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl512 -- python $ROOT/bench.py --batch-size 512 --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-strict --no-kernel-timing --small-batches "" > /dev/null 2>&1
python $ROOT/tools/step_timeline.py $OUT/tl512 step_prologue > $OUT/step_timeline_2p17_points.txt 2>&1
# 3. hash-grid kernels: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and SQ counters
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python $ROOT/tools/prof_hashgrid.py P > /dev/null 2>&1
done
python $ROOT/tools/make_traffic_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/pmc_traffic.json
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq1 -- python $ROOT/tools/prof_hashgrid.py P > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- python $ROOT/tools/prof_hashgrid.py P > /dev/null 2>&1
for p in 1 2; do cat $OUT/pmc_sq$p/*/*counter_collection.csv > $OUT/pmc_sq_hashgrid_pass$p.csv; done
python $ROOT/tools/pmc_summary.py aggregate $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/pmc_sq_hashgrid_aggregate_summary.txt
python $ROOT/tools/pmc_summary.py owner $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/pmc_sq_hashgrid_owner_summary.txt
python $ROOT/tools/pmc_summary.py fwd_cloud $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/pmc_sq_hashgrid_fwd_cloud_summary.txt
# 4. MLP kernels: SQ counters (three passes), HBM traffic
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/mlp_sq1 -- python $ROOT/tools/prof_mlp.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES --output-format csv -d $OUT/mlp_sq2 -- python $ROOT/tools/prof_mlp.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR --output-format csv -d $OUT/mlp_sq3 -- python $ROOT/tools/prof_mlp.py > /dev/null 2>&1
for p in 1 2 3; do cat $OUT/mlp_sq$p/*/*counter_collection.csv > $OUT/pmc_sq_mlp_pass$p.csv; done
{ echo "== mlp_bwd_ws"; python $ROOT/tools/pmc_summary.py mlp_bwd_ws $OUT/mlp_sq1 $OUT/mlp_sq2 $OUT/mlp_sq3; echo "== mlp_fwd_pf"; python $ROOT/tools/pmc_summary.py mlp_fwd_pf $OUT/mlp_sq1 $OUT/mlp_sq2 $OUT/mlp_sq3; } > $OUT/pmc_sq_mlp_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/mlp_$c -- python $ROOT/tools/prof_mlp.py > /dev/null 2>&1
done
{ for k in mlp_bwd_ws mlp_fwd_pf; do echo "== $k (KiB per launch; FETCH_SIZE counts 128-B requests at 64 B on gfx950: double it)"; python $ROOT/tools/pmc_summary.py $k $OUT/mlp_FETCH_SIZE $OUT/mlp_WRITE_SIZE; done; } > $OUT/pmc_traffic_mlp.txt
python - "$OUT/pmc_traffic_mlp.txt" "$NESVOR_COMMIT" > $OUT/pmc_traffic_mlp.json <<'PY'
import json, re, sys
out = {"commit": sys.argv[2] + " (kernels as of the collection run, tools/collect_profiles_r06.sh)",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python tools/prof_mlp.py ; density-network shape "
                 "(32 -> 64 -> 64 -> 16), N = 2^20, two-way fp16 split evaluation, bits-only compact save",
       "unit": "bytes per launch",
       "note": "FETCH_SIZE, WRITE_SIZE are reported in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)"}
kernel = None
for line in open(sys.argv[1]):
    m = re.match(r"== (\S+)", line)
    if m:
        kernel = m.group(1); out[kernel] = {}
        continue
    m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+(\d+)", line)
    if m and kernel:
        out[kernel][m.group(1) + "_KiB"] = int(m.group(2))
for k, v in out.items():
    if isinstance(v, dict) and "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v:
        v["traffic_bytes"] = (2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024
print(json.dumps(out, indent=1))
PY
# 5. HBM bytes of a whole training step (all kernels of 10 steps, two passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/step_$c -- python $ROOT/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-kernel-timing --no-extras --no-strict --small-batches "" > /dev/null 2>&1
done
python $ROOT/tools/step_traffic.py $OUT/step_FETCH_SIZE $OUT/step_WRITE_SIZE 20 > $OUT/step_traffic.json
# 6. micro-benchmarks
python $ROOT/tools/bench_hashgrid.py 2>&1 | grep -v amdgpu.ids > $OUT/hashgrid_microbench.log
python $ROOT/tools/bench_hg_levels.py 2>&1 | grep -v amdgpu.ids > $OUT/hashgrid_per_level.log
# 7. uniform points through the unclustered backward: kernel stats of its seven launches, per-level times, record counts
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/uni -- python $ROOT/tools/prof_hg_uniform.py 1 > $OUT/uniform_backward.log 2>/dev/null
cp $(ls $OUT/uni/*/*kernel_stats.csv | head -1) $OUT/uniform_kernel_stats.csv
python $ROOT/tools/bench_hg_levels.py U 1 2>&1 | grep -v amdgpu.ids > $OUT/uniform_per_level.log
python $ROOT/tools/queue_stats.py U 1 2>&1 | grep -v amdgpu.ids > $OUT/uniform_queue_stats.log
python $ROOT/tools/queue_stats.py 2>&1 | grep -v amdgpu.ids > $OUT/psf_queue_stats.log
# 8. instruction issue rates
hipcc --offload-arch=gfx950 -O3 $ROOT/tools/valu_rate_probe.hip -o /tmp/valu_rate_probe 2>/dev/null && /tmp/valu_rate_probe > $OUT/valu_rate_probe.log 2>&1
rm -rf $OUT/kstats $OUT/tl $OUT/tl512 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/mlp_sq1 $OUT/mlp_sq2 $OUT/mlp_sq3 $OUT/mlp_FETCH_SIZE $OUT/mlp_WRITE_SIZE $OUT/step_FETCH_SIZE $OUT/step_WRITE_SIZE
ls -la $OUT
