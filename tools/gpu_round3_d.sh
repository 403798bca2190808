#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o /tmp/splitp 2>/dev/null && /tmp/splitp > gpurun_out/r03_split_probe.log 2>&1
cat gpurun_out/r03_split_probe.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "one_call or direct_step or fused_trainer" > gpurun_out/r03_native_step_tests.log 2>&1; echo "native step tests rc=$?"
tail -15 gpurun_out/r03_native_step_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-strict --steps 100 > gpurun_out/r03_bench_native.json 2> gpurun_out/r03_bench_native.err; echo "bench rc=$?"
NESVOR_DDP_FORCE=1 timeout 600 python bench.py --no-cpu-baseline --no-strict --steps 100 > gpurun_out/r03_bench_native_ddp.json 2> gpurun_out/r03_bench_native_ddp.err; echo "bench ddp rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r03_bench_native.json", "gpurun_out/r03_bench_native_ddp.json"):
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith('{"metric"')][-1])
    except Exception as e:
        print(f, "unparsable", e); continue
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    for r in (d.get("small_batch") or {}).get("runs", []):
        print("  small", r["batch_pixels"], "ms/step", round(r["ms_per_step"], 4), "host issue", round(r["host_issue_ms_per_step"], 4), "timed kernels", round(r["timed_kernels_ms_per_step"], 4))
PY
tail -3 gpurun_out/r03_bench_native.err
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "c5" > gpurun_out/r03_fullsize_c5.log 2>&1; echo "c5 rc=$?"
grep "C5\|passed\|failed" gpurun_out/r03_fullsize_c5.log | tail -5
