#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/hg_variants.py r3: nobound:-DNESVOR_HG_NOBOUND=1 oldfixed:-DNESVOR_HG_OLDFIXED=1 both:-DNESVOR_HG_NOBOUND=1,-DNESVOR_HG_OLDFIXED=1 r02:src=tools/scratch/hashgrid_r02.hip.txt > gpurun_out/r03_hg_variants.log 2>&1
cat gpurun_out/r03_hg_variants.log
