#!/bin/bash
# The data-parallel step on ONE GPU (RCCL in a group of one rank) under different hardware-queue counts and side-stream
# priorities (round-3 verdict item 6: GPU_MAX_HW_QUEUES=8 took the step from 1.28 to 1.93 ms).
OUT=$1; mkdir -p $OUT
run() {  # label, env...
  label=$1; shift
  env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-strict --no-kernel-timing --small-batches "" 2>/dev/null > $OUT/q_$label.json
  python -c "
import json;d=json.load(open('$OUT/q_$label.json'));print('$label', round(d['value'],1),'it/s', round(d['ms_per_step'],4),'ms')"
}
run single_noddp X=1
run q_default NESVOR_DDP_FORCE=1 X=1
run q4 NESVOR_DDP_FORCE=1 GPU_MAX_HW_QUEUES=4
run q8 NESVOR_DDP_FORCE=1 GPU_MAX_HW_QUEUES=8
run q8_side_low NESVOR_DDP_FORCE=1 GPU_MAX_HW_QUEUES=8 NESVOR_SIDE_STREAM_PRIORITY=0
run q8_side_high NESVOR_DDP_FORCE=1 GPU_MAX_HW_QUEUES=8 NESVOR_SIDE_STREAM_PRIORITY=-1
run q2 NESVOR_DDP_FORCE=1 GPU_MAX_HW_QUEUES=2
run q8_noearly NESVOR_DDP_FORCE=1 GPU_MAX_HW_QUEUES=8 NESVOR_DDP_EARLY_ADAMW=0
run q8_nooverlap NESVOR_DDP_FORCE=1 GPU_MAX_HW_QUEUES=8 NESVOR_DDP_OVERLAP=0
run q4_nooverlap NESVOR_DDP_FORCE=1 GPU_MAX_HW_QUEUES=4 NESVOR_DDP_OVERLAP=0
