#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2j15; mkdir -p $O
timeout 900 python tools/hg_variants.py direct:-DNESVOR_HG_TRANSPOSE=0 tr16: tr24:-DNESVOR_HG_TRANSPOSE=24 tr8:-DNESVOR_HG_TRANSPOSE=8 > $O/hg_variants.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -x -q -k "hashgrid or grid or step or train" > $O/pytest.log 2>&1
timeout 1500 bash tools/ab_step.sh $O direct=/tmp/nesvor_variants/libdirect.so tr16=/tmp/nesvor_variants/libtr16.so tr24=/tmp/nesvor_variants/libtr24.so > $O/ab.log 2>&1
python tools/bench_hg_levels.py > $O/levels.log 2>&1
cat $O/hg_variants.log; tail -3 $O/pytest.log; cat $O/ab.log; tail -17 $O/levels.log
