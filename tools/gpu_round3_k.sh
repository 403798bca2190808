#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -m gpu -x -k "one_call or hashgrid or direct_step or data_parallel" > gpurun_out/r03_k_tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r03_k_tests.log
NESVOR_DDP_FORCE=1 timeout 600 python bench.py --no-cpu-baseline --no-strict --no-extras --steps 200 --small-batches "2048,1024,512" > gpurun_out/r03_bench_k_ddp.json 2> gpurun_out/r03_bench_k_ddp.err; echo "bench ddp rc=$?"
NESVOR_DDP_FORCE=1 NESVOR_DDP_SHARDED=1 timeout 600 python bench.py --no-cpu-baseline --no-strict --no-extras --steps 200 --small-batches "" > gpurun_out/r03_bench_k_ddp_sharded.json 2> gpurun_out/r03_bench_k_ddp_sharded.err; echo "bench ddp sharded rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r03_bench_k_ddp.json", "gpurun_out/r03_bench_k_ddp_sharded.json"):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith('{"metric"')][-1])
    except Exception as e:
        print(f, "unparsable", e); continue
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"] if d.get("roofline") else None)
    for r in (d.get("small_batch") or {}).get("runs", []):
        print("  small", r["batch_pixels"], "ms/step", round(r["ms_per_step"], 4), "host issue", round(r["host_issue_ms_per_step"], 4), "timed kernels", round(r["timed_kernels_ms_per_step"], 4))
PY
