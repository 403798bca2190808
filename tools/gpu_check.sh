#!/bin/bash
# One GPU-box job: rebuild everything from source, the default bench line, smoke(), the GPU test suite.
# Outputs under gpurun_out/check/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/check; mkdir -p $O
python nesvor_amd/csrc/build.py --force > $O/build.log 2>&1; echo "build rc=$?"
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench_n1.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
