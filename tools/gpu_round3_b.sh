#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_overlap.hip -o /tmp/ovl16 2>/dev/null && /tmp/ovl16 > gpurun_out/r03_mfma_bf16_overlap.log 2>&1
cat gpurun_out/r03_mfma_bf16_overlap.log
AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 600 python tools/scratch/repro_c5.py > gpurun_out/r03_repro_c5.log 2>&1; echo "repro rc=$?"
grep -v "^  File\|^Extension" gpurun_out/r03_repro_c5.log | tail -30
