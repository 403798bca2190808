#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/mlp_variants.py r02:src=tools/scratch/mlp_r02.hip.txt split0:-DNESVOR_SPLIT=0 split2:-DNESVOR_SPLIT=2 r02b:src=tools/scratch/mlp_r02.hip.txt > gpurun_out/r03_mlp_variants.log 2>&1
cat gpurun_out/r03_mlp_variants.log
