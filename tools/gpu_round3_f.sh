#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "mlp" > gpurun_out/r03_mlp_tests.log 2>&1; echo "mlp tests rc=$?"
tail -8 gpurun_out/r03_mlp_tests.log
python tools/mlp_variants.py split0:-DNESVOR_SPLIT=0 split1:-DNESVOR_SPLIT=1 split2:-DNESVOR_SPLIT=2 > gpurun_out/r03_mlp_variants.log 2>&1
cat gpurun_out/r03_mlp_variants.log
