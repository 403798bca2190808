#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
O=$ROOT/gpurun_out/job; mkdir -p $O
timeout 2000 python -m pytest tests -q -x -m gpu > $O/t_all.log 2>&1; echo "t_all rc=$?"; tail -6 $O/t_all.log
