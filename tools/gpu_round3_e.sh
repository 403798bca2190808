#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o /tmp/splitp 2>/dev/null && /tmp/splitp > gpurun_out/r03_split_probe.log 2>&1
grep -v "^     x" gpurun_out/r03_split_probe.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "one_call" > gpurun_out/r03_native_step_tests.log 2>&1; echo "native step tests rc=$?"
grep -v "^\s*$" gpurun_out/r03_native_step_tests.log | tail -40
