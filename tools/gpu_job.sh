#!/bin/bash
# scratch GPU job (edited per experiment)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
O=$ROOT/gpurun_out/job; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -q -x > $O/t2.log 2>&1; echo "t2 rc=$?"; tail -3 $O/t2.log
for v in 1 0 1 0; do
NESVOR_DEFER_TABLE_JOIN=$v timeout 600 python bench.py --no-extras --no-cpu-baseline --no-strict --steps 300 2> $O/b$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$v', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
for v in 1; do
rocprofv3 --kernel-trace --output-format csv -d $O/tl$v -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-strict --no-kernel-timing --small-batches "" > /dev/null 2>&1
python $ROOT/tools/step_timeline.py $O/tl$v step_prologue > $O/step_timeline_$v.txt 2>&1
cat $O/step_timeline_$v.txt | cut -c1-150
rm -rf $O/tl$v
done
