#!/bin/bash
# In-job A/B of the whole training step between builds of the library (boxes differ by a few percent, so only runs of
# one job compare):   bash tools/ab_step.sh <out-dir> name=path/to/lib.so[,ENV=VAL] [name=path ...]   (two alternating rounds)
OUT=$1; shift
mkdir -p $OUT
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; rest=${spec#*=}; lib=${rest%%,*}; extra=""
    if [ "$lib" != "$rest" ]; then extra=${rest#*,}; extra=${extra//,/ }; fi   # name=lib.so,ENV=VAL[,ENV2=VAL2]: extra environment settings
    env NESVOR_HIP_LIB=$lib $extra python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-strict --small-batches "" 2>/dev/null > $OUT/ab_${name}_$round.json
    python - $OUT/ab_${name}_$round.json $name $round <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["roofline"]["kernels_ms_per_step"]
print(f"{sys.argv[2]:10s} round {sys.argv[3]}: {d['value']:.1f} it/s  {d['ms_per_step']:.4f} ms  roofline {d['roofline']['frac']:.4f}  " + " ".join(f"{a}={b:.4f}" for a, b in k.items()))
PY
  done
done
