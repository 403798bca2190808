#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/mlp_variants.py r02:src=tools/scratch/mlp_r02.hip.txt pre0:-DNESVOR_MLP_APREFETCH=0 pre1:-DNESVOR_MLP_APREFETCH=1 pre1s0:-DNESVOR_MLP_APREFETCH=1,-DNESVOR_SPLIT=0 > gpurun_out/r03_mlp_variants.log 2>&1
cat gpurun_out/r03_mlp_variants.log
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_fullsize.py > gpurun_out/r03_gputests.log 2>&1; echo "suite rc=$?"
tail -6 gpurun_out/r03_gputests.log
timeout 600 python bench.py --no-cpu-baseline --steps 200 > gpurun_out/r03_bench_i.json 2> gpurun_out/r03_bench_i.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03_bench_i.json").read().splitlines() if l.startswith('{"metric"')][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernels_ms_per_step"], d["strict_fp32_mfma"])
for r in (d.get("small_batch") or {}).get("runs", []):
    print("  small", r["batch_pixels"], "ms/step", round(r["ms_per_step"], 4), "host issue", round(r["host_issue_ms_per_step"], 4), "timed kernels", round(r["timed_kernels_ms_per_step"], 4))
PY
