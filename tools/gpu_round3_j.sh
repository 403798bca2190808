#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -x -k "one_call or hashgrid or mlp or direct_step" > gpurun_out/r03_j_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r03_j_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-strict --steps 200 --small-batches "" > gpurun_out/r03_bench_j.json 2> gpurun_out/r03_bench_j.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03_bench_j.json").read().splitlines() if l.startswith('{"metric"')][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernels_ms_per_step"])
print(d["roofline_fwd_bwd_strict"])
PY
tail -3 gpurun_out/r03_bench_j.err
