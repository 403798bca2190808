#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "mlp" > gpurun_out/r03_mlp_tests.log 2>&1; echo "mlp tests rc=$?"
tail -5 gpurun_out/r03_mlp_tests.log
python tools/mlp_variants.py r02:src=tools/scratch/mlp_r02.hip.txt split0:-DNESVOR_SPLIT=0 split2:-DNESVOR_SPLIT=2 > gpurun_out/r03_mlp_variants.log 2>&1
cat gpurun_out/r03_mlp_variants.log
timeout 600 python bench.py --no-cpu-baseline --no-strict --steps 100 --small-batches "" > gpurun_out/r03_bench_h.json 2> gpurun_out/r03_bench_h.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03_bench_h.json").read().splitlines() if l.startswith('{"metric"')][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernels_ms_per_step"])
PY
