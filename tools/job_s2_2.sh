#!/bin/bash
# round 4 session 2, job 2: tr16 planes (swizzled) + sums on side stream + 2048-slot merge table
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2j2; mkdir -p $O
timeout 900 python tools/mlp_variants.py base: tr16:-DNESVOR_MLP_PLANES=2 tr16nl:-DNESVOR_MLP_PLANES=2,-DNESVOR_MLP_HLATE=0 tr16ns:-DNESVOR_MLP_PLANES=2,-DNESVOR_MLP_PLANE_SWZ=0 > $O/mlp_variants.log 2>&1
NESVOR_HIP_LIB=/tmp/nesvor_mlp_variants/libtr16.so timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "mlp or step or train or model" > $O/pytest_tr16.log 2>&1
timeout 900 python tools/hg_variants.py base: s2048:-DNESVOR_HG_SLOTS=2048,-DNESVOR_HG_MINBLOCKS=2 > $O/hg_variants.log 2>&1
L=nesvor_amd/lib/libnesvor_hip.so
timeout 1500 bash tools/ab_step.sh $O nosums=$L,NESVOR_STEP_SUMS_SIDE=0 sums=$L tr16=/tmp/nesvor_mlp_variants/libtr16.so tr16nl=/tmp/nesvor_mlp_variants/libtr16nl.so s2048=/tmp/nesvor_variants/libs2048.so > $O/ab.log 2>&1
cat $O/mlp_variants.log; tail -5 $O/pytest_tr16.log; cat $O/hg_variants.log; cat $O/ab.log
