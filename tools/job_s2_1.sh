#!/bin/bash
# round 4 session 2, job 1: tr16 planes A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2j1; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe > $O/tr16_probe.log 2>&1
timeout 900 python tools/mlp_variants.py base: tr20:-DNESVOR_MLP_PLANES=2 tr16:-DNESVOR_MLP_PLANES=2,-DNESVOR_MLP_PLANE_Q=16 planes1:-DNESVOR_MLP_PLANES=1 > $O/mlp_variants.log 2>&1
NESVOR_HIP_LIB=/tmp/nesvor_mlp_variants/libtr20.so timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "mlp or step or train or model" > $O/pytest_tr20.log 2>&1
timeout 900 bash tools/ab_step.sh $O base=/tmp/nesvor_mlp_variants/libbase.so tr20=/tmp/nesvor_mlp_variants/libtr20.so tr16=/tmp/nesvor_mlp_variants/libtr16.so > $O/ab.log 2>&1
tail -5 $O/tr16_probe.log; cat $O/mlp_variants.log; tail -5 $O/pytest_tr20.log; cat $O/ab.log
