#!/bin/bash
# scratch GPU job (edited per experiment)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
O=$ROOT/gpurun_out/job; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "mlp" > $O/t1.log 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -x > $O/t2.log 2>&1; echo "t2 rc=$?"; tail -3 $O/t2.log
for v in 1 1; do
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-strict --steps 300 2> $O/b$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline'].get('kernels_ms_per_step'))"
done
