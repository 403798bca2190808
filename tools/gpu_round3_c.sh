#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o /tmp/splitp 2>/dev/null && /tmp/splitp > gpurun_out/r03_split_probe.log 2>&1
cat gpurun_out/r03_split_probe.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "c5" > gpurun_out/r03_fullsize_c5.log 2>&1; echo "c5 rc=$?"
grep -v "^  File\|^Extension" gpurun_out/r03_fullsize_c5.log | tail -12
