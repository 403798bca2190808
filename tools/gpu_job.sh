#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
O=gpurun_out/job; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -x > $O/t2.log 2>&1; echo "t2 rc=$?"; tail -3 $O/t2.log
for v in 1 2; do
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-strict --steps 300 2> $O/b$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
done
for sw in NESVOR_ADAMW_IN_OWNER=0 NESVOR_STEP_NATIVE=0; do
  env $sw timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -x -k "not compact_save" > $O/ab_$sw.log 2>&1; echo "$sw rc=$?"; tail -2 $O/ab_$sw.log
done
