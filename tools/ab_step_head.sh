#!/bin/bash
# In-job A/B of the step's head: everything on the main stream (default) vs rounds 4-5's side-stream fork (NESVOR_STEP_HEAD=side),
# at 2^20 points (tools/ab_step.sh) and at 2^17 points (512 pixels).
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05ab
L=nesvor_amd/lib/libnesvor_hip.so
bash tools/ab_step.sh gpurun_out/r05ab main=$L side=$L,NESVOR_STEP_HEAD=side 2>&1 | cut -c1-60
for r in 1 2; do for h in main side; do
  echo -n "512 px, head on $h, round $r: "
  NESVOR_STEP_HEAD=$h python bench.py --batch-size 512 --steps 300 --warmup 20 --repeats 3 --no-cpu-baseline --no-extras --no-strict --no-kernel-timing --small-batches "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d.get('timed_regions_ms_per_step'))"
done; done
