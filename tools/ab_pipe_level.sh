#!/bin/bash
# Round 6: A/B of the pipelined table update (NESVOR_STEP_PIPE_LEVEL: the owner pass in two launches by level range, the next forward
# behind each) in one job, alternating rounds:  bash tools/ab_pipe_level.sh <out-dir> [levels...]
OUT=${1:-gpurun_out/ab_pipe}; shift
LEVELS=${@:-"0 12 10 13 0 12 10 13"}
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p $OUT
for lv in $LEVELS; do
  NESVOR_STEP_PIPE_LEVEL=$lv python bench.py --steps 200 --no-cpu-baseline --no-extras --no-strict > $OUT/pipe_$lv.json 2> $OUT/pipe_$lv.err
  python - "$OUT/pipe_$lv.json" $lv <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["roofline"]["kernels_ms_per_step"]
print(f"pipe level {sys.argv[2]:>2s}: {d['value']:.1f} it/s  {d['ms_per_step']:.4f} ms | fwd {k.get('hashgrid_fwd', 0):.4f} late {k.get('hashgrid_fwd_late', 0):.4f} "
      f"agg {k.get('hashgrid_bwd_aggregate', 0):.4f} owner {k.get('hashgrid_bwd_owner', 0):.4f} union {k.get('hashgrid_owner_fwd_union', 0):.4f}", flush=True)
PY
done
