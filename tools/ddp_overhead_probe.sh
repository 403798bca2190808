cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r05aj; mkdir -p $OUT
run() { label=$1; shift
  env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-strict --no-kernel-timing --small-batches "" 2>/dev/null > $OUT/q_$label.json
  python -c "
import json;d=json.load(open('$OUT/q_$label.json'));print('$label', round(d['value'],1),'it/s', round(d['ms_per_step'],4),'ms')"
}
for r in 1 2; do
run single X=1
run ddp_forced NESVOR_DDP_FORCE=1
run ddp_forced_nooverlap NESVOR_DDP_FORCE=1 NESVOR_DDP_OVERLAP=0
run ddp_forced_sharded NESVOR_DDP_FORCE=1 NESVOR_DDP_SHARDED=1
done
