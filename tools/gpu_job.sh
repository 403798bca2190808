#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
O=$ROOT/gpurun_out/job; mkdir -p $O
timeout 1200 python tools/mlp_variants.py base: mb3_768:-DNESVOR_FWD_MINBLOCKS=3,-DNESVOR_FWD_GRID=768 mb3_512:-DNESVOR_FWD_MINBLOCKS=3 base2: > $O/mlpv.log 2>&1; cat $O/mlpv.log | cut -c1-330
