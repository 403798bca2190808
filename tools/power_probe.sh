#!/bin/bash
# Engine clock / socket power while the training step runs (is the step power-limited on this box?):
#   bash tools/power_probe.sh <out-dir> name=lib.so [name=lib.so ...]
OUT=$1; shift
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  NESVOR_HIP_LIB=$lib python bench.py --steps 25000 --warmup 10 --no-cpu-baseline --no-extras --no-strict --no-kernel-timing --small-batches "" 2>/dev/null > $OUT/pp_$name.json &
  pid=$!
  sleep 16   # import + data synthesis + settle
  for i in $(seq 1 6); do
    echo "== $name sample $i"
    amd-smi metric -g 0 --power --clock 2>/dev/null | grep -E "SOCKET_POWER|GFX_0|CLK:|MIN_CLK|MAX_CLK|CLK_LOCKED|DEEP" | head -12 | tr '\n' ' '; echo
    sleep 2
  done
  wait $pid
  python -c "
import json,sys;d=json.load(open('$OUT/pp_$name.json'));print('$name', round(d['value'],1),'it/s', round(d['ms_per_step'],4),'ms')"
done
