#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -x -k "hashgrid or one_call" > gpurun_out/r03_n_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r03_n_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-strict --steps 200 --small-batches "" > gpurun_out/r03_bench_n.json 2> gpurun_out/r03_bench_n.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03_bench_n.json").read().splitlines() if l.startswith('{"metric"')][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernels_ms_per_step"])
s = d["roofline_fwd_bwd_strict"]; print({k: s[k] for k in s if k.endswith("ms") or k.startswith("frac")})
print("uniform", d["roofline_uniform"]["launch_ms"])
PY
python tools/hg_variants.py r3: > gpurun_out/r03_hg_variants.log 2>&1; cut -c1-300 gpurun_out/r03_hg_variants.log
