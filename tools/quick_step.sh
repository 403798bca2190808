#!/bin/bash
# One short GPU job: the bench line (driver's invocation) + the kernel timeline of one training step.   bash tools/quick_step.sh <tag>
TAG=${1:-q}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-strict --small-batches "" > $OUT/bench.json 2> $OUT/bench.err
python -c "import json;d=json.load(open('$OUT/bench.json'));print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --no-strict --no-kernel-timing --small-batches "" > /dev/null 2>&1
python $ROOT/tools/step_timeline.py $OUT/tl step_prologue > $OUT/step_timeline.txt 2>&1
rm -rf $OUT/tl
cat $OUT/step_timeline.txt
