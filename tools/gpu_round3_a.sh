#!/bin/bash
# round-3 GPU job A: new full-size parity tests, the whole -m gpu suite, bench (plain and with the RCCL exchange forced on)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python nesvor_amd/csrc/build.py --force > gpurun_out/r03_build.log 2>&1 || { tail -20 gpurun_out/r03_build.log; exit 1; }
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s > gpurun_out/r03_fullsize.log 2>&1; echo "fullsize rc=$?"
tail -5 gpurun_out/r03_fullsize.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize.py > gpurun_out/r03_gputests.log 2>&1; echo "suite rc=$?"
tail -15 gpurun_out/r03_gputests.log
timeout 600 python bench.py > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err; echo "bench rc=$?"
NESVOR_DDP_FORCE=1 timeout 600 python bench.py --no-cpu-baseline --no-strict > gpurun_out/r03_bench_ddp.json 2> gpurun_out/r03_bench_ddp.err; echo "bench ddp rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r03_bench_a.json", "gpurun_out/r03_bench_ddp.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unparsable", e); continue
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["accountings"])
    for r in (d.get("small_batch") or {}).get("runs", []):
        print("  small", r["batch_pixels"], "ms/step", round(r["ms_per_step"], 4), "host issue", round(r["host_issue_ms_per_step"], 4), "timed kernels", round(r["timed_kernels_ms_per_step"], 4))
PY
