#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -x -k "hashgrid or data_parallel or one_call" > gpurun_out/r03_l_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r03_l_tests.log
NESVOR_DDP_FORCE=1 timeout 600 python bench.py --no-cpu-baseline --no-strict --no-extras --steps 200 --small-batches "" > gpurun_out/r03_bench_l_ddp.json 2> gpurun_out/r03_bench_l_ddp.err; echo "bench ddp rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r03_bench_l_ddp.json",):
    d = json.loads([l for l in open(f).read().splitlines() if l.startswith('{"metric"')][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"] if d.get("roofline") else None)
PY
bash tools/collect_profiles_r03.sh c764866 > gpurun_out/r03_collect.log 2>&1; echo "collect rc=$?"
tail -30 gpurun_out/r03_collect.log
