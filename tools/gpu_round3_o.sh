#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/collect_profiles_r03.sh 2fb8cf9 > gpurun_out/r03_collect.log 2>&1; echo "collect rc=$?"
tail -3 gpurun_out/r03_collect.log
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r03_gputests_full.log 2>&1; echo "suite rc=$?"
tail -6 gpurun_out/r03_gputests_full.log
